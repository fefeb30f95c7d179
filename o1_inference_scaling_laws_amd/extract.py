"""response_cache.json -> dense vote tensors.

Resolves every (problem, token_limit, idx) through the SAME cache-key scheme as the reference
(/root/reference/o1.py:85-88 generation key, :119 extraction key) and applies its failure rule:
any sample the reference could not obtain from the cache -- missing generation (the reference then
raises NameError at o1.py:94), extraction cached as None (AssertionError at o1.py:163), missing
extraction (network call) -- becomes a vote for answer 0 with 0 tokens (o1.py:190-192).

Value domain: the engine histograms bins 0..1023.  AIME answers 0..999 map to themselves; any other
value the cache holds (negative, >= 1000, non-integral) is an ordinary candidate for
statistics.multimode, so it is dictionary-encoded per problem into the spare
bins 1000..1023.  A problem with more than 24 distinct such values is re-encoded DENSELY instead:
multimode and the score (o1.py:202-210) depend only on the equality classes of the votes and on
which class the truth is in, so ALL distinct values of that problem (truth included) are mapped
injectively, in first-seen order, onto bins 0..k-1 -- exact while the problem has at most 1024
distinct values (always, for the reference's N <= 128).  Only beyond that DomainOverflow is raised
(there is no CPU fallback).  For a densely re-encoded problem -- and for a spare bin -- the cell's ``min_mode``
is a code, not an answer; ``VoteTensors.decode_bin(p, bin)`` gives the answer back.

O1_MODEL and PROMPT are parameters: they are part of the key (o1.py:86) and are read from the
reference module at install time (o1_dropin.install), never copied into this repository.
"""
from __future__ import annotations

import decimal
import numbers
from dataclasses import dataclass

import numpy as np

SPARE_BASE, SPARE_COUNT = 1000, 24
FAILED_VOTE = (0, 0)  # o1.py:192


NUM_BINS = 1024


class DomainOverflow(ValueError):
    """The encoder at hand is full: > 24 distinct out-of-domain answers for ``ProblemEncoder`` (the
    extractor then switches the problem to ``DenseEncoder``), > 1024 distinct answers for ``DenseEncoder``."""


def generation_key(model: str, prompt: str, problem: str, token_limit: int, idx: int) -> str:
    """o1.py:85-88.  ``prompt`` is the UNFORMATTED template, as in the reference."""
    if idx > 0:
        return f"{model}_{prompt}_{problem}_{token_limit}_{idx}"
    return f"{model}_{prompt}_{problem}_{token_limit}"


def extraction_key(response_content: str) -> str:
    """o1.py:119."""
    return f"extract_answer_{response_content}"


def resolve_vote(cache: dict, model: str, prompt: str, problem: str, token_limit: int, idx: int):
    """generate_single_response (o1.py:148-164) restricted to cache hits -> (answer, tokens)."""
    gk = generation_key(model, prompt, problem, token_limit, idx)
    response = cache.get(gk)
    if response is None:
        return FAILED_VOTE
    try:
        content, tokens = response["content"], response["tokens"]
    except (KeyError, TypeError):
        return FAILED_VOTE
    ek = extraction_key(content)
    if ek not in cache:
        return FAILED_VOTE
    answer = cache[ek]
    if answer is None:      # o1.py:163 assert answer is not None
        return FAILED_VOTE
    return answer, tokens


def _canonical(v):
    """Python equality classes as Counter sees them (equal values hash equal): 3 == 3.0 == Fraction(3) == Decimal(3) == 3+0j,
    True == 1.  Every number with an integral value becomes that int; anything else is its own class."""
    if isinstance(v, (int, np.integer, np.bool_)):          # (bool is an int; np.bool_ is neither, but np.True_ == 1 for Counter)
        return int(v)
    if isinstance(v, (numbers.Number, decimal.Decimal)):
        try:
            if isinstance(v, complex):
                if v.imag != 0:
                    return v
                v = v.real
            if v == v and v == int(v):       # (NaN != NaN; int(inf) raises OverflowError)
                return int(v)
        except (TypeError, ValueError, ArithmeticError):     # (OverflowError: int(inf); decimal.InvalidOperation: Decimal('sNaN') == ...)
            pass
    return v


class ProblemEncoder:
    """Per-problem dictionary: in-domain ints -> themselves, anything else -> spare bins."""

    def __init__(self):
        self._codes = {}

    def encode(self, v) -> int:
        v = _canonical(v)
        if isinstance(v, int) and 0 <= v < SPARE_BASE:
            return v
        code = self._codes.get(v)
        if code is None:
            if len(self._codes) >= SPARE_COUNT:
                raise DomainOverflow(f"more than {SPARE_COUNT} distinct out-of-domain answers in one problem")
            code = SPARE_BASE + len(self._codes)
            self._codes[v] = code
        return code


class DenseEncoder:
    """Per-problem dictionary over ALL values: the i-th distinct value seen -> bin i (injective, so the
    equality classes -- all that statistics.multimode and ``truth in modes`` look at -- are preserved)."""

    def __init__(self):
        self._codes = {}

    def encode(self, v) -> int:
        v = _canonical(v)
        code = self._codes.get(v)
        if code is None:
            if len(self._codes) >= NUM_BINS:
                raise DomainOverflow(f"more than {NUM_BINS} distinct answers in one problem")
            code = len(self._codes)
            self._codes[v] = code
        return code

    def table(self):
        return list(self._codes)


@dataclass
class VoteTensors:
    answers: np.ndarray   # int32 [P, B, Nmax]
    tokens: np.ndarray    # int32 [P, B, Nmax]
    n_valid: np.ndarray   # int32 [B]
    truth: np.ndarray     # int32 [P]
    code_tables: dict = None   # p -> [value of bin 0, value of bin 1, ...] for densely re-encoded problems
    spare_tables: dict = None  # p -> {spare bin 1000..1023: the out-of-domain value it stands for}

    def decode_bin(self, p: int, bin_: int):
        """The answer value behind bin ``bin_`` of problem ``p`` -- e.g. ``cells["min_mode"][p, b]``: the bin itself for an
        in-domain answer, the original (canonical) value for a spare bin, the table entry for a densely re-encoded problem;
        ``None`` for -1 (no votes)."""
        bin_ = int(bin_)
        if bin_ < 0:
            return None
        dense = (self.code_tables or {}).get(p)
        if dense is not None:
            return dense[bin_]
        return (self.spare_tables or {}).get(p, {}).get(bin_, bin_)


def _in_domain_codes(vals):
    """int32 codes of one sample pool when EVERY answer is an int (or bool: True == 1) in 0..999 -- the only case the
    reference's extractor produces for AIME (o1.py:140 int(extracted_answer)) -- else None (the caller then encodes
    value by value).  One numpy conversion instead of a dictionary probe per vote."""
    try:
        arr = np.asarray(vals)
    except (ValueError, OverflowError, TypeError):
        return None
    if arr.ndim != 1 or arr.dtype.kind not in "iub":
        return None
    if arr.size and (int(arr.min()) < 0 or int(arr.max()) >= SPARE_BASE):
        return None
    return arr.astype(np.int32, copy=False)


def _token_codes(toks):
    """int32 token counts of one sample pool (o1.py:102 stores completion_tokens as a JSON number)."""
    try:
        arr = np.asarray(toks)
        if arr.ndim == 1 and arr.dtype.kind in "iub" and (not arr.size or (-2 ** 31 <= int(arr.min()) and int(arr.max()) < 2 ** 31)):
            return arr.astype(np.int32, copy=False)
    except (ValueError, OverflowError, TypeError):
        pass
    out = np.empty((len(toks),), dtype=np.int32)
    for i, tok in enumerate(toks):
        tok = int(tok)
        if not -2 ** 31 <= tok < 2 ** 31:
            raise ValueError(f"token count {tok} does not fit int32")
        out[i] = tok
    return out


_MISSING = object()


def build_vote_tensors(dataset, cache: dict, budgets, model: str, prompt: str, alloc=None) -> VoteTensors:
    """budgets: [(key_token_limit, N)] -- the (actual_token_limit, N) pairs of o1.py:274-277 / :302.

    Budget b of problem p votes over samples idx = 0..N_b-1 of key_token_limit_b (prefix of one
    sample pool when several budgets share a key_token_limit).

    ``alloc(shape, dtype) -> ndarray`` supplies the answers / tokens tensors (``Engine.pinned_empty``: page-locked memory
    the HOST-mode call DMAs in place); default: ordinary numpy memory.

    Cost per sample: what the reference's key scheme forces -- one generation key built and hashed (o1.py:85-88: the
    ~0.6 KB prompt + the problem text are part of EVERY key), one extraction key built and hashed (o1.py:119: the whole
    completion text) -- and nothing else: the key prefix is formatted once per (problem, token limit), the loop body is
    three dictionary probes on local names, answers and token counts are collected per sample pool and converted by numpy
    in one piece, and every budget of a pool is a slice copy of it.  (resolve_vote() above is the same rule, one sample
    at a time.)
    """
    P, B = len(dataset), len(budgets)
    nmax = max([n for _, n in budgets], default=0)
    shape = (P, B, max(nmax, 1))
    if alloc is None:
        answers = np.zeros(shape, dtype=np.int32)
        tokens = np.zeros(shape, dtype=np.int32)
    else:
        answers, tokens = alloc(shape, np.int32), alloc(shape, np.int32)
        answers[...] = 0
        tokens[...] = 0
    n_valid = np.array([n for _, n in budgets], dtype=np.int32).reshape(B)
    truth = np.zeros((P,), dtype=np.int32)
    code_tables, spare_tables = {}, {}
    pools = {}                                                  # key_limit -> samples needed (first-seen order of the budgets)
    for key_limit, n in budgets:
        pools[key_limit] = max(pools.get(key_limit, 0), n)
    suffix = [""] + [f"_{i}" for i in range(1, nmax)]           # o1.py:85-88: idx 0 has no suffix
    get = cache.get
    missing = _MISSING
    for p, example in enumerate(dataset):
        problem = example["problem"]
        raw = {}                                                # key_limit -> (answers, tokens) of its sample pool
        for key_limit, n in pools.items():
            prefix = f"{model}_{prompt}_{problem}_{key_limit}"
            vals, toks = [0] * n, [0] * n                       # FAILED_VOTE = (0, 0) unless the sample resolves (o1.py:190-192)
            for idx in range(n):
                response = get(prefix + suffix[idx])
                if response is None:
                    continue
                try:
                    content, tok = response["content"], response["tokens"]
                except (KeyError, TypeError):
                    continue
                ans = get(f"extract_answer_{content}", missing)
                if ans is missing or ans is None:               # o1.py:163 assert answer is not None
                    continue
                vals[idx], toks[idx] = ans, tok
            raw[key_limit] = (vals, _token_codes(toks))
        true_answer = int(example["answer"])                    # o1.py:206
        codes = {}
        if 0 <= true_answer < SPARE_BASE:
            truth_code = true_answer
            for key_limit, (vals, _toks) in raw.items():
                arr = _in_domain_codes(vals)
                if arr is None:
                    codes = None
                    break
                codes[key_limit] = arr
        else:
            codes = None
        if codes is None:
            # some answer is not an in-domain int: encode value by value, in the first-seen order of the budgets' samples
            order, seen = [], set()
            for key_limit, n in budgets:
                for idx in range(n):
                    if (key_limit, idx) not in seen:
                        seen.add((key_limit, idx))
                        order.append((key_limit, idx))
            try:
                enc = ProblemEncoder()
                truth_code = enc.encode(true_answer)
                flat = {k: enc.encode(raw[k[0]][0][k[1]]) for k in order}
                if enc._codes:
                    spare_tables[p] = {code: value for value, code in enc._codes.items()}
            except DomainOverflow:                              # > 24 distinct out-of-domain values: dense re-encoding
                enc = DenseEncoder()
                truth_code = enc.encode(true_answer)
                flat = {k: enc.encode(raw[k[0]][0][k[1]]) for k in order}
                code_tables[p] = enc.table()
            codes = {key_limit: np.fromiter((flat[(key_limit, i)] for i in range(n)), dtype=np.int32, count=n)
                     for key_limit, n in pools.items()}
        truth[p] = truth_code
        for b, (key_limit, n) in enumerate(budgets):
            if n:
                answers[p, b, :n] = codes[key_limit][:n]
                tokens[p, b, :n] = raw[key_limit][1][:n]
    return VoteTensors(answers, tokens, n_valid, truth, code_tables, spare_tables)

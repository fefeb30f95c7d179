"""TEST INFRASTRUCTURE -- run the UNMODIFIED reference (/root/reference/o1.py) under stub modules.

Where /root/reference exists (the build container) the reference is imported from there.  On the GPU
box it does not exist: there the harness imports the sourceless bytecode ``oracle/make_ref.py`` compiled
from it into the git-ignored ``oracle/_ref/`` (the same code objects CPython would execute from the
sources).  Used by tests/golden/make_golden.py to generate the committed fixtures, by
``tests/test_reference_live.py`` (CPU: oracle adapter as the engine; ``-m gpu``: the HIP engine) and by
``bench.py``'s ``cpu_baseline.reference_loop``.  Recipe: SURVEY.md 8c.

No reference source is copied into the repository: O1_MODEL / PROMPT are read out of o1.py at run time
(ast) -- or out of oracle/_ref/manifest.json, where the build recorded them -- because they are part of
the cache-key scheme (o1.py:85-88).
"""
from __future__ import annotations

import ast
import contextlib
import importlib
import io
import json
import os
import sys
import tempfile
import types

REFERENCE_SOURCES = "/root/reference"


def _sources_present() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_SOURCES, "o1.py"))


def reference_root() -> str | None:
    """Directory ``import o1`` is resolved from: the read-only sources, else the compiled copy in oracle/_ref."""
    if _sources_present():
        return REFERENCE_SOURCES
    from oracle import make_ref
    return make_ref.REF_OUT if make_ref.available() else None


def reference_kind() -> str | None:
    """"sources" | "bytecode" | None -- for the result files that quote a reference timing."""
    root = reference_root()
    return None if root is None else ("sources" if root == REFERENCE_SOURCES else "bytecode")


def reference_available() -> bool:
    return reference_root() is not None


def reference_constants() -> dict:
    """O1_MODEL, PROMPT, RESPONSE_CACHE_FILENAME as assigned at o1.py:17,20,21-30."""
    if not _sources_present():
        from oracle import make_ref
        return dict(make_ref.manifest()["constants"])
    with open(os.path.join(REFERENCE_SOURCES, "o1.py")) as f:
        tree = ast.parse(f.read())
    out = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            name = node.targets[0].id
            if name in ("O1_MODEL", "PROMPT", "RESPONSE_CACHE_FILENAME"):
                out[name] = ast.literal_eval(node.value)
    assert set(out) == {"O1_MODEL", "PROMPT", "RESPONSE_CACHE_FILENAME"}, out.keys()
    return out


def generation_key(model: str, prompt: str, problem: str, token_limit: int, idx: int) -> str:
    """o1.py:85-88 (note: PROMPT is the UNFORMATTED template)."""
    if idx > 0:
        return f"{model}_{prompt}_{problem}_{token_limit}_{idx}"
    return f"{model}_{prompt}_{problem}_{token_limit}"


def extraction_key(content: str) -> str:
    """o1.py:119."""
    return f"extract_answer_{content}"


class _Dataset(list):
    """Stands in for the HF dataset object at o1.py:40-47 (only .filter and len/iter are used)."""

    def filter(self, fn):
        return _Dataset([ex for ex in self if fn(ex)])


def make_dataset(answers, prefix="Problem"):
    """30 problems (o1.py:46 asserts 30) with string answers like the HF column (o1.py:206)."""
    return _Dataset(
        {"problem": f"{prefix} {i}: compute f({i}).", "answer": str(a), "url": f"https://aops/2024_AIME_{i}"}
        for i, a in enumerate(answers)
    )


def build_cache(consts, dataset, samples):
    """samples: iterable of (problem_idx, token_limit, idx, answer, tokens).

    answer: Python int (extraction hit), None (extraction cached as None -> AssertionError at
    o1.py:163 -> vote (0, 0) at o1.py:190-192), or the string "MISSING" (no generation entry at all
    -> cache miss -> NameError at o1.py:94 -> vote (0, 0)).
    The completion text is made unique per sample so extraction keys never collide.
    """
    cache = {}
    for (p, token_limit, idx, answer, tokens) in samples:
        if answer == "MISSING":
            continue
        problem = dataset[p]["problem"]
        content = f"[completion p={p} T={token_limit} i={idx}] final answer: {answer}"
        cache[generation_key(consts["O1_MODEL"], consts["PROMPT"], problem, token_limit, idx)] = {
            "content": content, "tokens": int(tokens)}
        cache[extraction_key(content)] = answer
    return cache


def _install_stubs(dataset):
    openai = types.ModuleType("openai")

    class _Completions:
        def create(self, *a, **k):
            raise RuntimeError("stub OpenAI client: network call attempted (cache miss)")

    class OpenAI:  # noqa: N801 - mirrors the imported name at o1.py:6
        def __init__(self, *a, **k):
            self.chat = types.SimpleNamespace(completions=_Completions())

    openai.OpenAI = OpenAI
    ipython = types.ModuleType("IPython")
    ipython.embed = lambda *a, **k: None
    ipython.get_ipython = lambda: None
    ipython.version_info = (9, 0, 0)  # matplotlib probes sys.modules['IPython'].version_info
    datasets = types.ModuleType("datasets")
    datasets.load_dataset = lambda name: {"train": dataset}
    saved = {k: sys.modules.get(k) for k in ("openai", "IPython", "datasets")}
    sys.modules["openai"], sys.modules["IPython"], sys.modules["datasets"] = openai, ipython, datasets
    return saved


@contextlib.contextmanager
def imported_reference(dataset, import_cache):
    """Import the unmodified o1.py (which RUNS the whole pipeline at import, o1.py:312-315) in a
    scratch cwd holding helpers/response_cache.json = import_cache.  Yields (module, workdir);
    workdir/helpers/results_log_*.json are the files the import-time run wrote."""
    root = reference_root()
    assert root is not None, "neither /root/reference nor oracle/_ref (python -m oracle.make_ref) is present"
    import logging
    old_cwd = os.getcwd()
    old_env = {k: os.environ.get(k) for k in ("OPENAI_API_KEY", "MPLBACKEND")}
    saved_mods = _install_stubs(dataset)
    workdir = tempfile.mkdtemp(prefix="scv_ref_")
    os.makedirs(os.path.join(workdir, "helpers"))
    os.makedirs(os.path.join(workdir, "graphs"))
    with open(os.path.join(workdir, "helpers", "response_cache.json"), "w") as f:
        json.dump(import_cache, f)
    os.environ["OPENAI_API_KEY"] = "stub"
    os.environ["MPLBACKEND"] = "Agg"
    sys.path.insert(0, root)
    for m in [k for k in sys.modules if k == "o1" or k == "helpers" or k.startswith("helpers.")]:
        del sys.modules[m]
    os.chdir(workdir)
    root_level = logging.getLogger().level
    try:
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
            mod = importlib.import_module("o1")
        logging.getLogger().setLevel(logging.CRITICAL + 1)  # o1.py:191 logs a traceback per failed vote
        yield mod, workdir
    finally:
        logging.getLogger().setLevel(root_level)
        os.chdir(old_cwd)
        sys.path.remove(root)
        for m in [k for k in sys.modules if k == "o1" or k == "helpers" or k.startswith("helpers.")]:
            del sys.modules[m]
        for k, v in saved_mods.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        for k, v in old_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

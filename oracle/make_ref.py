"""TEST INFRASTRUCTURE -- recipe that materialises ``oracle/_ref/``: the UNMODIFIED reference, compiled.

The reference (/root/reference/o1.py + helpers/plot_helpers.py) is pure Python, so "compiling it from the sources
where they lie" means byte-compiling: ``py_compile`` turns each file into a sourceless ``.pyc`` under
``oracle/_ref/`` (git-ignored like every built artefact, NOT gpurun-ignored, so it travels to the GPU box next to
``libscvote.so``).  No reference source text enters the repository or its history; what lands in ``_ref`` is the
binary CPython 3.10 executes anyway.  ``manifest.json`` records the sha256 of every source file the bytecode was
built from, the interpreter's bytecode magic, and the three module constants the cache-key scheme needs
(o1.py:17,20,21-30: O1_MODEL, RESPONSE_CACHE_FILENAME, PROMPT), read with ``ast`` exactly like
``ref_harness.reference_constants``.

Run by ``__graft_entry__.build()`` when /root/reference is present (the build container); the GPU box only uses the
prebuilt files.  ``oracle/ref_harness.py`` imports the reference from /root/reference when it exists and from here
otherwise, so the same harness drives

  * the ``-m gpu`` live-reference tests (the unmodified drivers + plot/log writers with the HIP engine installed), and
  * ``bench.py``'s ``cpu_baseline.reference_loop`` (the unmodified ``o1.run_experiments`` timed on the GPU box's host).
"""
from __future__ import annotations

import ast
import hashlib
import importlib.util
import json
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference"
REF_OUT = os.path.join(HERE, "_ref")
FILES = ("o1.py", "helpers/plot_helpers.py")        # everything SURVEY 8a/8b cites; examine_helper.py is out of scope


def _sha256(path: str) -> str:
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def _constants(src_path: str) -> dict:
    with open(src_path) as f:
        tree = ast.parse(f.read())
    out = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            if node.targets[0].id in ("O1_MODEL", "PROMPT", "RESPONSE_CACHE_FILENAME"):
                out[node.targets[0].id] = ast.literal_eval(node.value)
    assert set(out) == {"O1_MODEL", "PROMPT", "RESPONSE_CACHE_FILENAME"}, sorted(out)
    return out


def available() -> bool:
    """True when oracle/_ref holds bytecode this interpreter can import."""
    man = os.path.join(REF_OUT, "manifest.json")
    if not os.path.isfile(man):
        return False
    try:
        with open(man) as f:
            m = json.load(f)
    except (OSError, ValueError):
        return False
    return m.get("magic") == importlib.util.MAGIC_NUMBER.hex() and all(
        os.path.isfile(os.path.join(REF_OUT, f + "c")) for f in FILES)


def manifest() -> dict:
    with open(os.path.join(REF_OUT, "manifest.json")) as f:
        return json.load(f)


def build(force: bool = False) -> str | None:
    """Byte-compile the reference into oracle/_ref/.  Returns the directory, or None when /root/reference is absent
    (then whatever a previous build left is used as is)."""
    if not os.path.isfile(os.path.join(REF_SRC, "o1.py")):
        return REF_OUT if available() else None
    want = {f: _sha256(os.path.join(REF_SRC, f)) for f in FILES}
    if not force and available() and manifest().get("sha256") == want:
        return REF_OUT
    shutil.rmtree(REF_OUT, ignore_errors=True)
    os.makedirs(os.path.join(REF_OUT, "helpers"))
    for f in FILES:
        # dfile: the path tracebacks show -- the original location, so errors still cite /root/reference/o1.py:LINE
        py_compile.compile(os.path.join(REF_SRC, f), cfile=os.path.join(REF_OUT, f + "c"),
                           dfile=os.path.join(REF_SRC, f), doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
    with open(os.path.join(REF_OUT, "manifest.json"), "w") as fh:
        json.dump({"source_root": REF_SRC, "sha256": want, "magic": importlib.util.MAGIC_NUMBER.hex(),
                   "python": sys.version.split()[0], "constants": _constants(os.path.join(REF_SRC, "o1.py")),
                   "what": "sourceless bytecode of the unmodified reference (py_compile); test infrastructure only"},
                  fh, indent=1)
    return REF_OUT


if __name__ == "__main__":
    print(build(force=True))

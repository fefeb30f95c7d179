"""The C-ABI library loads and exports every symbol include/scvote.h declares.  No compute calls
(no GPU here); on a GPU-less box the product must refuse loudly, never fall back."""
import ctypes
import os
import re

import numpy as np
import pytest

from o1_inference_scaling_laws_amd import _build, _lib
from o1_inference_scaling_laws_amd.engine import CELL_DTYPE

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(REPO, "include", "scvote.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(scv_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ("scv_create", "scv_destroy", "scv_aggregate_i32", "scv_bootstrap", "scv_synth_fill_i32",
                 "scv_last_error", "scv_device_count", "scv_last_kernel_ns"):
        assert must in names


def test_library_is_built_in_tree_and_exports_every_declared_symbol():
    if not os.path.exists(_build.LIB_PATH):            # fresh checkout: the .so is git-ignored; hipcc cross-compiles gfx950 here
        _build.build()
    assert os.path.exists(_build.LIB_PATH), "run __graft_entry__.build() first"
    assert os.path.commonpath([REPO, _build.LIB_PATH]) == REPO
    L = ctypes.CDLL(_build.LIB_PATH)
    for name in declared_functions():
        assert hasattr(L, name), f"{name} declared in include/scvote.h but not exported"


def test_binding_declares_every_symbol():
    L = _lib.load()
    for name in declared_functions():
        fn = getattr(L, name)
        assert fn.argtypes is not None, name


def test_cell_struct_layout_matches_header():
    src = open(os.path.join(REPO, "include", "scvote.h")).read()
    body = re.search(r"typedef struct scv_cell \{(.*?)\} scv_cell;", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(u?int\d+_t)\s+(\w+)(\[\d+\])?;", body)
    assert [f[1] for f in fields] == list(CELL_DTYPE.names)
    assert CELL_DTYPE.itemsize == 16 and CELL_DTYPE.fields["min_mode"][1] == 10 and CELL_DTYPE.fields["hit"][1] == 12


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from o1_inference_scaling_laws_amd.engine import Engine
    with pytest.raises(_lib.ScvError) as ei:
        Engine()
    assert ei.value.code == _lib.ERR_NO_DEVICE
    assert _lib.load().scv_device_count() == 0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "o1_inference_scaling_laws_amd")
    for root, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "libscv_oracle" not in text and "scv_oracle.c" not in text, f
    _ = np


def test_every_option_and_stat_key_is_documented_in_the_header():
    """scv_set_option / scv_get_stat accept string keys: each key the library compares against must be named in the
    header's documentation of that entry point (the keys are part of the boundary)."""
    src = open(os.path.join(REPO, "o1_inference_scaling_laws_amd", "csrc", "scvote.hip")).read()
    hdr = open(os.path.join(REPO, "include", "scvote.h")).read()
    for fn in ("scv_set_option", "scv_get_stat"):
        body = src[src.index(f"int {fn}("):]
        body = body[: body.index("\n}\n")]
        keys = re.findall(r'!strcmp\(key, "([a-z0-9_]+)"\)', body)
        # round 4: at most 15 option keys are part of the boundary (+ the deprecated alias "auto_geometry"); stat keys (read-only counters) up to 20
        assert 5 <= len(keys) <= (16 if fn == "scv_set_option" else 20), (fn, len(keys))
        missing = [k for k in keys if f'"{k}"' not in hdr]
        assert not missing, f"{fn}: keys not documented in include/scvote.h: {missing}"


def test_no_kernel_spills_to_scratch():
    """Compiler metadata of every gfx950 kernel (one device-only compile, ~1 min, no GPU): nothing on the hot paths may
    carry scratch -- a launch bound set for more waves than a variant's registers allow shows up here, not as a silent
    slowdown on the GPU box (the token variants of the one-vector register shapes once did: 48-84 B)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(REPO, "tools", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = mod.collect()
    assert 100 <= len(rows) <= 154, len(rows)                        # round 4: pruned from 231 to <= 130; round 6: + scv_reg_cells<8, 3 | 4>, scv_sort_prefix2<true>, scv_one_vote, scv_sort_cells<24 | 40 | 56>, - scv_merge_partials
    spilled = {r[0]: r[4] for r in rows if r[4]}                     # (round 5: no exception left -- 17..32 votes with tokens on one lane per cell went to scv_reg_cells)
    assert not spilled, spilled
    # SGPR values kept in VGPR lanes (round 6): the prefix sort kernels had 157 .. 602 of them, read back inside the step loop -- 64 hoisted "i < nmax" masks
    # of a rare branch, 16 + 16 slot conditions, one select mask per scanned vote; 20 .. 110 are left, none of those in a loop per vote
    lanes = {r[0]: r[7] for r in rows if r[0].startswith("scv_sort_prefix")}
    assert len(lanes) == 6 and max(lanes.values()) <= 110 and lanes["scv_sort_prefix2<false>"] <= 64 and lanes["scv_sort_prefix<64, false>"] <= 48, lanes
    head = [r for r in rows if r[0] == "scv_hist_argmax<4, 1024, 4, false, false>"]
    assert head and head[0][1] <= 128 and head[0][4] == 0            # the headline kernel: 16 waves per CU need <= 128 VGPRs


def test_copy_pipelines_keep_their_waits():
    """The one-lane-per-cell kernels copy a step's rows HBM -> LDS by LDS-DMA and keep the NEXT copy in flight while the current rows are sorted.
    hipcc does not count those copies (inline asm), so every `s_waitcnt vmcnt` behind the first copy is one the source put there on purpose; a
    load the compiler can see inside the step loop would make it add a vmcnt(0) of its own and drain the copy in flight (how the truth gather
    became a 4-byte LDS-DMA in round 4, and why round 5's prefix kernels work out the budget classes before their first copy).  Device-only
    compile, no GPU: the waits behind the first `global_load_lds` of each kernel are counted in its ISA."""
    import re
    import subprocess
    import tempfile
    csrc = os.path.join(REPO, "o1_inference_scaling_laws_amd", "csrc")
    expect = {                                                       # kernel (mangled-name fragment) -> vmcnt waits from its step loop on: (least, most)
        "scv_sort_prefixILi64ELb0EE": (1, 1), "scv_sort_prefixILi32ELb0EE": (1, 1),      # the top of a step
        "scv_sort_prefixILi64ELb1EE": (2, 2), "scv_sort_prefixILi32ELb1EE": (2, 2),      # + the step's tokens, behind the sort
        "scv_sort_prefix2ILb0EE": (2, 2),                                                    # half A, half B
        "scv_sort_prefix2ILb1EE": (4, 4),                                                    # + the token steps behind the sort: their half A, half B
        "scv_sort_cellsILi64ELb0ELb0EE": (3, 5), "scv_sort_cellsILi16ELb0ELb0EE": (3, 5),  # vmcnt(0) / (1) / (2) by the stores left in flight
    }
    with tempfile.TemporaryDirectory() as d:
        isa = ""
        for unit in ("scvote_sort_prefix.hip", "scvote_sort.hip"):
            out = os.path.join(d, unit + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", out, os.path.join(csrc, unit)],
                           check=True, stderr=subprocess.DEVNULL)
            isa += open(out).read()
    for frag, (least, most) in expect.items():
        m = re.search(r"^(_ZN3scv\d+" + re.escape(frag) + r"\w*):[^\n]*\n(.*?)s_endpgm", isa, re.S | re.M)
        assert m, frag
        body = m.group(2)
        assert "scratch_" not in body, frag
        behind = body[body.index("SCV_STEP_LOOP"):] if "SCV_STEP_LOOP" in body else body[body.index("global_load_lds_dword"):]
        waits = re.findall(r"vmcnt\(\d+\)", behind)
        assert least <= len(waits) <= most, (frag, waits)
        if "sort_prefix" in frag:                                    # (scv_sort_cells keeps a load for budget lists beyond its n_valid cache, in a branch of its own)
            assert not re.search(r"\b(global|flat|buffer)_load_dword", behind.replace("global_load_lds_dword", "")), frag      # no load the compiler counts


def test_communicator_refuses_without_a_device():
    """scv_comm_create on a box without a GPU: SCV_ERR_NO_DEVICE, no communicator, no crash (and no RCCL needed to say so)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _lib.load()
    comm = ctypes.c_void_p()
    for flags in (_lib.COMM_PEER, _lib.COMM_RCCL):
        assert L.scv_comm_create(ctypes.byref(comm), None, 0, 0, flags) == _lib.ERR_NO_DEVICE and not comm.value
    assert b"no HIP device" in L.scv_last_error()


def test_every_extern_c_entry_runs_inside_the_exception_guard():
    """SURVEY 8b / INTEGRATION.md: no C++ exception crosses the ABI.  Every int-returning extern "C" function of the two host
    translation units has a body that is one `return guarded([&]() -> int { ... });` (csrc/scvote.hip, csrc/scvote_comm.hip),
    except the one-line accessors that cannot throw; the GPU suite injects the faults (test_no_cpp_exception_crosses_the_abi)."""
    trivial = {"scv_comm_size"}
    for unit in ("scvote.hip", "scvote_comm.hip"):
        src = open(os.path.join(REPO, "o1_inference_scaling_laws_amd", "csrc", unit)).read()
        blocks = re.findall(r'extern "C" \{(.*?)\n\}  // extern "C"', src, flags=re.S)
        assert blocks, unit
        seen = 0
        for blk in blocks:
            for m in re.finditer(r"^int (scv_\w+)\([^)]*\) \{\n(.*?)\n\}$", blk, flags=re.S | re.M):
                name, body = m.group(1), m.group(2)
                if name in trivial:
                    continue
                seen += 1
                assert body.lstrip().startswith("return guarded([&]() -> int {") and body.rstrip().endswith("});"), (unit, name)
        assert seen >= (19 if unit == "scvote.hip" else 7), (unit, seen)


def test_no_null_stream_memset_or_memcpy_on_the_launch_path():
    """hipMemset / hipMemcpy (the forms without a stream) run on the NULL stream and, for device memory, return before the device is done; a
    context's own stream is non-blocking and does not wait for the null stream.  Round 6: the split-N scratch was cleared that way right before a
    launch and lost sums (profiles/r06_split_scratch_race.log).  The product sources may use the stream-less forms only in scv_create, where a
    hipDeviceSynchronize follows before the context is handed out."""
    csrc = os.path.join(REPO, "o1_inference_scaling_laws_amd", "csrc")
    found = []
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h")):
            continue
        for i, line in enumerate(open(os.path.join(csrc, f)).read().split("\n"), 1):
            code = line.split("//")[0]
            if re.search(r"\bhipMemset\(|\bhipMemcpy\(|\bhipMemsetD\d+\(|\bhipMemcpy(DtoH|HtoD|DtoD)\(", code):
                found.append((f, i, code.strip()))
    assert [x[0] for x in found] == ["scvote.hip", "scvote.hip"], found          # d_err and d_tickets at scv_create
    src = open(os.path.join(csrc, "scvote.hip")).read().split("\n")
    last = max(i for _, i, _ in found)
    assert any("hipDeviceSynchronize()" in l for l in src[last - 1: last + 3]), "the memsets of scv_create must be followed by a device synchronisation"
    assert all("ctx->d_err" in c or "ctx->d_tickets" in c for _, _, c in found), found

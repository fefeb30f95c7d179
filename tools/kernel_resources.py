#!/usr/bin/env python3
"""Register / LDS / scratch usage of every gfx950 kernel of the library (all translation units of csrc/, compiled in
parallel), from the compiler's own metadata (hipcc --cuda-device-only -S; no GPU needed).  Usage: kernel_resources.py [out.md]"""
import os
import re
import subprocess
import sys
import tempfile

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(R, "o1_inference_scaling_laws_amd", "csrc")


def collect():
    """[(kernel, vgpr, agpr, sgpr, scratch_bytes, occupancy, static_lds_bytes, sgpr_spills)], sorted by name.  sgpr_spills = scalar values the register
    allocator keeps in VGPR lanes (v_writelane / v_readlane + hazard s_nops wherever they are used: no memory traffic, but VALU issue slots -- round 6
    found 578 of them in scv_sort_prefix2, most read back inside its step loop: DESIGN.md 3.8)."""
    from concurrent.futures import ThreadPoolExecutor
    units = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    with tempfile.TemporaryDirectory() as d:
        def one(unit):
            asm = os.path.join(d, unit + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", asm,
                            os.path.join(CSRC, unit)], check=True, stderr=subprocess.DEVNULL)
            return open(asm).read()
        with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 4)) as pool:
            s = "\n".join(pool.map(one, units))
    mangled_names = re.findall(r"\.amdhsa_kernel (\S+)", s)
    demangled = subprocess.run(["c++filt"], input="\n".join(mangled_names), capture_output=True, text=True).stdout.splitlines()
    spills = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n){0,14}?\s+\.sgpr_spill_count:\s+(\d+)", s)}
    rows = []
    for mangled, name in zip(mangled_names, demangled):
        i = s.find("; -- End function", s.find("\n" + mangled + ":"))
        info = s[i:i + 8000]
        g = lambda k: int(re.search(k + r": (\d+)", info).group(1))   # noqa: E731
        short = re.sub(r"\(scv::AggArgs\)|\(.*\)$", "", name).replace("void ", "").replace("scv::", "")
        rows.append((short, g("NumVgprs"), g("NumAgprs"), g("NumSgprs"), g("ScratchSize"), g("Occupancy"), g("LDSByteSize"), spills.get(mangled, 0)))
    rows.sort()
    return rows


def main():
    rows = collect()
    out = ["# Kernel resources (gfx950, hipcc -O3; compiler metadata, `tools/kernel_resources.py`)", "",
           f"{len(rows)} kernels; `occupancy` = waves per SIMD the register count allows (LDS may allow fewer: dynamic LDS is sized by the host).", "",
           "| kernel | VGPR | AGPR | SGPR | scratch B | occupancy | static LDS B | SGPRs kept in VGPR lanes |", "|---|---|---|---|---|---|---|---|"]
    out += [f"| `{n}` | {v} | {a} | {sg} | {sc} | {oc} | {l} | {sp} |" for n, v, a, sg, sc, oc, l, sp in rows]
    spill = [r for r in rows if r[4]]
    out += ["", f"Kernels with scratch: {len(spill)}" + (": " + ", ".join(f"`{r[0]}` ({r[4]} B)" for r in spill) if spill else "") + "."]
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()

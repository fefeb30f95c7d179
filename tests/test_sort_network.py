"""CPU proof of the counting core of the sorted-cells kernel (csrc/scvote_sort.hip.h): the compile-time compare-exchange
network (csrc/scvote_sortnet.h, plain C++) sorts -- 0-1 principle, exhaustive up to 24 wires and, bit-sliced, for all 2^32 inputs of the 32-wire network -- and a scalar emulation of the device code's packed form
(lockstep halves, one cross merge, run-length scan with the carry between the halves, sentinels) equals a brute-force
statistics.multimode for every shape.  tests/sortnet_check.cpp is compiled with g++ and run here; no GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_sorting_network_and_packed_scan_on_cpu(tmp_path):
    exe = tmp_path / "sortnet_check"
    subprocess.check_call(["g++", "-O3", "-std=c++17", "-o", str(exe), os.path.join(HERE, "sortnet_check.cpp")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "network: sorts (exchanges on 4 / 8 / 16 / 24 / 32 wires: 5 19 63 132 191" in out.stdout
    assert "valley merge: sorts every 0-1 valley (exchanges on 24 wires: 52)" in out.stdout
    assert "packed sort + scan: equals statistics.multimode" in out.stdout
    # round 5: the 32-wire network on ALL 2^32 0-1 inputs (bit-sliced), the block property scv_sort_prefix rests on (after merge phase p every
    # aligned block of 2 p wires is sorted), the two-file merge of scv_sort_prefix2 on every pair of sorted 0-1 halves, and scalar restatements
    # of the prefix kernels' scans (keyed, key-free, 128 votes with sentinels) against a brute-force statistics.multimode
    assert "32 wires, all 2^32 inputs, every aligned block of 2 p wires after phase p: sorted" in out.stdout
    assert "two-file merge: sorts every pair of sorted 0-1 halves" in out.stdout
    assert "prefix block scans + 128-vote scan: equal statistics.multimode" in out.stdout

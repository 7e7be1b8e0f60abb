"""The row kernels keep the store discipline of csrc/store_fence.h: in the generated gfx950 ISA no counted
`s_waitcnt vmcnt(N > 0)` releases consumers while a vector-memory store and a load can both be in flight (the schedule
bank_get and dfa_points had when they faulted in lanes 48-63 beside a busy second queue: DESIGN.md section 4).
hipcc -S cross-compiles without a GPU."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_store_scan  # noqa: E402

# every kernel file but gemm.hip, whose only flagged store is the zero fill of a dead tile (rows past m_live), a
# workgroup-uniform branch that returns right after it; the scan reads the ISA as straight-line text and cannot see that.
ROW_KERNEL_FILES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(ROOT, "simpb_amd", "csrc", "*.hip"))
                          if os.path.basename(p) != "gemm.hip")


@pytest.mark.parametrize("name", ROW_KERNEL_FILES)
def test_no_store_among_outstanding_loads(name, tmp_path):
    src = os.path.join(ROOT, "simpb_amd", "csrc", name + ".hip")
    asm = str(tmp_path / (name + ".s"))
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", asm, src],
                   check=True, stderr=subprocess.DEVNULL)
    hits = isa_store_scan.scan(asm)
    assert not hits, {k: v[0] for k, v in hits.items()}


def test_scanner_sees_the_pattern(tmp_path):
    """The scanner flags the schedule bank_get had in round 1 (store, then counted waits on older loads)."""
    asm = tmp_path / "k.s"
    asm.write_text("kern:\n\tglobal_load_dwordx4 v[2:5], v[0:1], off\n\tglobal_load_dwordx4 v[6:9], v[0:1], off offset:16\n"
                   "\ts_waitcnt vmcnt(1)\n\tglobal_store_dwordx3 v[10:11], v[2:4], off\n\ts_waitcnt vmcnt(1)\n"
                   "\tv_mov_b32_e32 v12, v6\n\ts_endpgm\n")
    assert "kern" in isa_store_scan.scan(str(asm))
    asm.write_text("kern:\n\tglobal_load_dwordx4 v[2:5], v[0:1], off\n\ts_waitcnt vmcnt(0)\n"
                   "\tglobal_store_dwordx3 v[10:11], v[2:4], off\n\ts_endpgm\n")
    assert not isa_store_scan.scan(str(asm))
    # dfa_points' round-1 loop: this camera's stores in flight while the next camera's rows arrive behind counted waits
    asm.write_text("kern:\n.LBB0_1:\n\tglobal_load_dwordx4 v[2:5], v[0:1], off\n\tglobal_load_dwordx4 v[6:9], v[0:1], off offset:16\n"
                   "\ts_waitcnt vmcnt(1)\n\tv_mov_b32_e32 v12, v2\n\ts_waitcnt vmcnt(0)\n\tglobal_store_dword v[10:11], v12, off\n"
                   "\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n")
    assert not isa_store_scan.scan(str(asm))  # (straight-line reading: the loop-carried store needs the real kernel's unroll)

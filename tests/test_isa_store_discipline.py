"""Three rules checked on the generated gfx950 ISA (hipcc -S cross-compiles without a GPU; skipped where there is no hipcc):

1. NO double-K matrix instruction anywhere in the product library. While a v_mfma_f32_32x32x16_f16 / 16x16x32_f16 (or
   their bf16 siblings, gfx950 only) executes anywhere on the chip, OTHER kernels' vector arithmetic goes wrong in lanes
   48-63 of a wave (profiles/r02_mfma_x16_interference/): the backbone stream runs beside the decoder
   stream by design, so one stray builtin (or -DSIMPB_MFMA_F16_K16=1) would corrupt detections silently. Checked on the
   assembly of every csrc/*.hip AND on the disassembly of the library the tests load.
2. The row kernels keep the store discipline of csrc/store_fence.h: no counted `s_waitcnt vmcnt(N > 0)` releases consumers
   while a vector-memory store and a load can both be in flight (the schedule bank_get and dfa_points had when they were
   the first victims). Hardening, not the fix (rule 1 is). Files that include store_fence.h are scanned with loop
   back-edges followed (a store at the end of one trip is in flight during the next trip's counted waits); the others
   (samplers, convolutions, backward kernels: grid-stride loops that prefetch across trips on purpose) get the
   straight-line reading only, as a regression guard; gemm.hip is exempt (its flagged store is the zero fill of a dead
   tile behind a workgroup-uniform branch that returns right after).
3. The sampler / row kernels are built without packed-FP32 instructions (test_row_kernels_carry_no_packed_fp32)."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_store_scan  # noqa: E402

CSRC = os.path.join(ROOT, "simpb_amd", "csrc")
ALL_FILES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(CSRC, "*.hip")))
ROW_KERNEL_FILES = [n for n in ALL_FILES if n != "gemm"]
needs_hipcc = pytest.mark.skipif(isa_store_scan.hipcc() is None, reason="no hipcc on this host ($HIPCC, /opt/rocm/bin, PATH)")


def _disciplined(name):
    return "store_fence.h" in open(os.path.join(CSRC, name + ".hip")).read()


@pytest.fixture(scope="session")
def asm(tmp_path_factory):
    """{stem: assembly listing} of every csrc/*.hip, compiled once (in parallel) for all ISA tests."""
    return isa_store_scan.compile_all(str(tmp_path_factory.mktemp("isa")))


@needs_hipcc
@pytest.mark.parametrize("name", ROW_KERNEL_FILES)
def test_no_store_among_outstanding_loads(name, asm):
    hits = isa_store_scan.scan(asm[name], loops=_disciplined(name))
    assert not hits, {k: v[0] for k, v in hits.items()}


@needs_hipcc
@pytest.mark.parametrize("name", ALL_FILES)
def test_no_double_k_mfma_in_source_isa(name, asm):
    assert "v_mfma" in open(asm["gemm"]).read()  # the listing does contain matrix instructions: the scan is not vacuous
    assert not isa_store_scan.banned_mfma(asm[name])


@needs_hipcc
def test_row_kernels_carry_no_packed_fp32(asm):
    """Rule 3 (victim side of rule 1, profiles/r02_mfma_x16_interference/README.md "Victim side"): the sampler / row kernels --
    every kernel ever recorded as a victim of a double-K matrix instruction running elsewhere on the chip -- contain no
    packed-FP32 instruction; with them daf_fwd_rows returned wrong channels in most launches beside such a kernel, without
    them in none of 1 000 (simpb_amd/build.py: NO_PACKED_FP32, per file)."""
    from simpb_amd import build
    assert set(build.ROW_KERNEL_FILES) <= set(ALL_FILES)
    for name in build.ROW_KERNEL_FILES:
        text = open(asm[name]).read()
        hits = sorted({m.group(0) for m in isa_store_scan.PACKED_FP32.finditer(text)})
        assert not hits, (name, hits)
    # (the scan does see the instruction where it is allowed: the matrix kernels' epilogues)
    assert any(isa_store_scan.PACKED_FP32.search(open(asm[n]).read()) for n in ALL_FILES if n not in build.ROW_KERNEL_FILES)


def test_no_double_k_mfma_in_built_library(tmp_path):
    """The library the GPU tests and the bench load, disassembled: what actually runs, whatever flags built it."""
    objdump = next((p for p in ("/opt/rocm/lib/llvm/bin/llvm-objdump", shutil.which("llvm-objdump")) if p and os.path.exists(p)), None)
    if objdump is None:
        pytest.skip("no llvm-objdump on this host")
    from simpb_amd import build
    if not os.path.exists(build.LIB):
        pytest.skip("library not built (python -m simpb_amd.build)")
    lib = shutil.copy(build.LIB, tmp_path / "lib.so")   # the bundle extractor writes next to its input
    subprocess.run([objdump, "--offloading", str(lib)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp_path)
    objs = sorted(glob.glob(str(tmp_path / "lib.so.*gfx950*")))
    assert objs, "no gfx950 code object in the library"
    text = "".join(subprocess.run([objdump, "-d", o], check=True, capture_output=True, text=True).stdout for o in objs)
    assert text.count("v_mfma_f32_32x32x8_f16") > 100 and text.count("v_mfma_f32_32x32x2_f32") > 100  # the scan sees matrix code
    bad = sorted({m.group(0) for m in isa_store_scan.BANNED_MFMA.finditer(text)})
    assert not bad, bad


def test_banned_mfma_pattern():
    """The pattern covers the instructions the interference table names and spares the ones measured clean."""
    for bad in ("v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x32_bf16",
                "v_mfma_f32_16x16x128_f8f6f4", "v_mfma_f32_32x32x64_f8f6f4", "v_mfma_i32_32x32x32_i8", "v_mfma_i32_16x16x64_i8",
                "v_mfma_f32_16x16x32_fp8_fp8", "v_smfmac_f32_16x16x64_f16", "v_mfma_f32_32x32x16_bf8_bf8"):
        assert isa_store_scan.BANNED_MFMA.search("\t" + bad + " a[0:15], v[0:3], v[4:7], a[0:15]"), bad
    for ok in ("v_mfma_f32_32x32x8_f16", "v_mfma_f32_16x16x16_f16", "v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x4_f32",
               "v_mfma_f32_4x4x1_16b_f32", "v_mfma_f32_32x32x8f16", "v_mfma_f32_4x4x4_16b_f16"):
        assert not isa_store_scan.BANNED_MFMA.search("\t" + ok + " a[0:15], v[0:1], v[2:3], a[0:15]"), ok


def test_scanner_sees_the_pattern(tmp_path):
    """The scanner flags the schedule bank_get had in round 1 (store, then counted waits on older loads) and the
    loop-carried form dfa_points had (this trip's store in flight while the next trip's loads arrive behind counted waits)."""
    asm = tmp_path / "k.s"
    asm.write_text("kern:\n\tglobal_load_dwordx4 v[2:5], v[0:1], off\n\tglobal_load_dwordx4 v[6:9], v[0:1], off offset:16\n"
                   "\ts_waitcnt vmcnt(1)\n\tglobal_store_dwordx3 v[10:11], v[2:4], off\n\ts_waitcnt vmcnt(1)\n"
                   "\tv_mov_b32_e32 v12, v6\n\ts_endpgm\n")
    assert "kern" in isa_store_scan.scan(str(asm))
    asm.write_text("kern:\n\tglobal_load_dwordx4 v[2:5], v[0:1], off\n\ts_waitcnt vmcnt(0)\n"
                   "\tglobal_store_dwordx3 v[10:11], v[2:4], off\n\ts_endpgm\n")
    assert not isa_store_scan.scan(str(asm))
    loop = ("kern:\n.LBB0_1:\n\tglobal_load_dwordx4 v[2:5], v[0:1], off\n\tglobal_load_dwordx4 v[6:9], v[0:1], off offset:16\n"
            "\ts_waitcnt vmcnt(1)\n\tv_mov_b32_e32 v12, v2\n\ts_waitcnt vmcnt(0)\n\tglobal_store_dword v[10:11], v12, off\n"
            "%s\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n")
    asm.write_text(loop % "")
    assert not isa_store_scan.scan(str(asm), loops=False)   # the straight-line reading misses it ...
    assert "kern" in isa_store_scan.scan(str(asm))           # ... the back-edge carries the store into the next trip
    asm.write_text(loop % "\ts_waitcnt vmcnt(0)\n")          # store retired before the branch (store_fence.h stores_retired)
    assert not isa_store_scan.scan(str(asm))

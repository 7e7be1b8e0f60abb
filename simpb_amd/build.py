"""Build the C-ABI HIP library in-tree: hipcc --offload-arch=gfx950 -> simpb_amd/csrc/libsimpb_hip.so.
hipcc cross-compiles without a GPU, so this also runs in the build container. One hipcc process per source file
(they are independent translation units), then one link."""
import glob
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(CSRC, "libsimpb_hip.so")
STAMP = LIB + ".srchash"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# The sampler / row kernels (plain vector arithmetic on gathered rows and small tables: every kernel that was ever recorded as
# a VICTIM of the gfx950 double-K matrix instructions running elsewhere on the chip) are built without
# packed-FP32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 come from the SLP and loop vectorisers). Round 3, one
# run of tools/daf_stress.py: daf_fwd_rows beside a conv1x1 built on v_mfma_f32_32x32x16_f16 returned wrong channels in
# 892-1 758 of 1 000-2 000 launches with packed FP32 and in 0 of 1 000 without (profiles/r02_mfma_x16_interference/README.md,
# "Victim side"). The product issues no such matrix instruction any more; this keeps its vector kernels right beside a
# foreign one as well. Costs nothing measurable in these latency-bound kernels (the whole library without SLP: +0.6 % per frame).
NO_PACKED_FP32 = ["-fno-slp-vectorize", "-fno-vectorize"]
ROW_KERNEL_FILES = ("deform_agg", "deform_agg_fused", "msda", "msda_lin", "alloc", "bank", "rowops", "dfa_prep", "decode", "format")


def flags_for(src):
    stem = os.path.basename(src)[:-4]
    return FLAGS + (NO_PACKED_FP32 if stem in ROW_KERNEL_FILES else [])


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(os.path.dirname(HERE), "include", "simpb_hip.h")]


def source_hash():
    h = hashlib.sha256()
    for path in sources() + headers():
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(repr((FLAGS, NO_PACKED_FP32, ROW_KERNEL_FILES)).encode())   # a flag change is a rebuild as well
    return h.hexdigest()


def needs_build():
    """Content hash, not mtimes: the snapshot sent to the GPU box does not keep mtimes, and a
    rebuild there would only repeat the one done here."""
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    return open(STAMP).read().strip() != source_hash()


def build_extension(force=False, verbose=False, jobs=None, extra_flags=(), out=None):
    """Compile csrc/*.hip into the in-tree library. `extra_flags` + `out` build a VARIANT next to it instead (its own
    object directory, no stamp): used by tools/daf_stress.py to rebuild the kernels with the single-instruction FP16
    matrix step (-DSIMPB_MFMA_F16_K16=1, csrc/mfma_f16.h) for the interference measurement."""
    variant = out is not None
    if not variant and not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj_dir = OBJ if not variant else os.path.join(os.path.dirname(os.path.abspath(out)), "_obj")
    os.makedirs(obj_dir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc] + flags_for(src) + list(extra_flags) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as pool:
        objects = list(pool.map(compile_one, sources()))
    target = LIB if not variant else os.path.abspath(out)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-o", target] + objects
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, check=True)
    if not variant:
        with open(STAMP, "w") as f:
            f.write(source_hash())
    return target


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))

"""Build the C-ABI HIP library in-tree: hipcc --offload-arch=gfx950 -> simpb_amd/csrc/libsimpb_hip.so.
hipcc cross-compiles without a GPU, so this also runs in the build container."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libsimpb_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


STAMP = LIB + ".srchash"


def source_hash():
    import hashlib
    h = hashlib.sha256()
    for path in sources() + [os.path.join(os.path.dirname(HERE), "include", "simpb_hip.h")]:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    """Content hash, not mtimes: the snapshot sent to the GPU box does not keep mtimes, and a
    rebuild there would only repeat the one done here."""
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    return open(STAMP).read().strip() != source_hash()


def build_extension(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as f:
        f.write(source_hash())
    return LIB


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))

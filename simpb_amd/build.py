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


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(os.path.dirname(HERE), "include", "simpb_hip.h")]
    return any(os.path.getmtime(s) > t for s in deps)


def build_extension(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))

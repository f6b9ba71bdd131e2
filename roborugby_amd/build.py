"""Builds the HIP shared library in-tree (roborugby_amd/libroborugby_amd.so) for gfx950."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "rr_kernels.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "rr_sim.hpp"), os.path.join(HERE, "csrc", "rr_extras.hpp"),
        os.path.join(os.path.dirname(HERE), "include", "roborugby_amd.h")]
LIB = os.path.join(HERE, "libroborugby_amd.so")
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def find_hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X library cannot be built here")


def is_stale():
    return not os.path.exists(LIB) or any(os.path.getmtime(LIB) < os.path.getmtime(d) for d in DEPS)


def build_hip_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> libroborugby_amd.so (cross-compiles without a GPU)."""
    if not force and not is_stale():
        return LIB
    cmd = [find_hipcc()] + HIPCC_FLAGS + ["-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_hip_library(force=True, verbose=True))

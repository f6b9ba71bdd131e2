"""Builds the HIP shared library in-tree (roborugby_amd/libroborugby_amd.so) for gfx950.

Staleness is decided by CONTENT, not mtime: the build records the sha256 of every source it compiled (and of the compiler
flags) next to the library; a library whose recorded hash differs from the sources' is rebuilt before anything loads it.
(A snapshot copied to another box keeps contents, not necessarily timestamps.)"""
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "rr_kernels.hip")
SRC_KSTEP = os.path.join(HERE, "csrc", "rr_kstep_inst.hip")  # explicit instantiations of the step kernel, one share per -DRR_PART
SRC_DQN = os.path.join(HERE, "csrc", "rr_dqn.hip")            # fused DQN update (config 5): its own translation unit
KSTEP_PARTS = 7                                                # == RR_KSTEP_PARTS in csrc/rr_kstep.hpp
DEPS = [SRC, SRC_KSTEP, SRC_DQN, os.path.join(HERE, "csrc", "rr_kstep.hpp"), os.path.join(HERE, "csrc", "rr_sim.hpp"),
        os.path.join(HERE, "csrc", "rr_extras.hpp"), os.path.join(os.path.dirname(HERE), "include", "roborugby_amd.h")]
LIB = os.path.join(HERE, "libroborugby_amd.so")
# the exact-trig parity build: the same sources with -DRR_EXACT_TRIG=1 (sin / cos of the robot kinematics in double-double, ~correctly
# rounded: agrees with the reference's glibc in 99.8 % of the evaluations instead of 97 %; csrc/rr_sim.hpp) -- opt-in, slower
LIB_EXACT = os.path.join(HERE, "libroborugby_amd_exact.so")
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def find_hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X library cannot be built here")


def sources_present():
    return all(os.path.exists(d) for d in DEPS)


def source_hash():
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for d in DEPS:
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read())
    return h.hexdigest()


def is_stale(exact=False):
    lib = LIB_EXACT if exact else LIB
    if not os.path.exists(lib) or not os.path.exists(lib + ".srchash"):
        return True
    with open(lib + ".srchash") as f:
        return f.read().strip() != source_hash()


def build_hip_library(force=False, verbose=False, jobs=None, exact=False):
    """hipcc --offload-arch=gfx950 ... -> libroborugby_amd.so (cross-compiles without a GPU).

    The step kernel's instantiations (14 configurations x up to five variants: most of the compile time) are compiled as
    KSTEP_PARTS objects next to the main translation unit, in parallel, and linked into the one shared library; no relocatable
    device code (every kernel is self-contained, csrc/rr_kstep.hpp).  RR_BUILD_JOBS / `jobs` bounds the parallelism."""
    lib_path = LIB_EXACT if exact else LIB
    if not force and not is_stale(exact):
        return lib_path
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    digest = source_hash()
    hipcc = find_hipcc()
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + (["-DRR_EXACT_TRIG=1"] if exact else [])
    jobs = jobs or int(os.environ.get("RR_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 1)
    with tempfile.TemporaryDirectory(prefix="rr_build_") as tmpd:
        units = [(SRC, ["-DRR_SPLIT_BUILD"], os.path.join(tmpd, "rr_kernels.o"))]
        units += [(SRC_KSTEP, [f"-DRR_PART={k}"], os.path.join(tmpd, f"rr_kstep_{k}.o")) for k in range(KSTEP_PARTS)]
        units += [(SRC_DQN, [], os.path.join(tmpd, "rr_dqn.o"))]

        def compile_one(u):
            src, defs, obj = u
            cmd = [hipcc] + cflags + defs + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            return obj

        with ThreadPoolExecutor(max_workers=jobs) as ex:
            objs = list(ex.map(compile_one, units))
        tmp = lib_path + f".tmp{os.getpid()}"
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", tmp] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    os.replace(tmp, lib_path)  # atomic: a concurrent loader sees the old or the new library, never half of one
    with open(lib_path + ".srchash", "w") as f:
        f.write(digest + "\n")
    return lib_path


# ---- one-shape libraries: entity counts outside the built list (the reference's counts are free integers, RR_Constants.py:30-34)
SHAPES_DIR = os.path.join(HERE, "shapes")
BUILT_SHAPES = {(1, 0, 1, 0), (2, 2, 4, 4), (1, 1, 1, 1)}  # inside libroborugby_amd.so (csrc/rr_kstep.hpp: RR_FOR_EACH_CFG)


def shape_lanes(nrh, nrg, nbp, nbn):
    """lanes per arena of a one-shape library: one lane per entity, a power of two >= 2; raises for counts the phases' masks cannot hold"""
    nr, nb = nrh + nrg, nbp + nbn
    if min(nrh, nrg, nbp, nbn) < 0 or nrh < 1 or nb < 1 or nbp < 1:
        raise ValueError("at least one happy robot and one positive ball (the observation and the rewards are built around them)")
    if nr > 8 or nb > 11 or nr * nb > 32:
        raise ValueError(f"{nr} robots x {nb} balls: the contact masks hold at most 8 robots, 11 balls and 32 ball-robot pairs per arena")
    vw = 2
    while vw < max(nr, nb):
        vw *= 2
    return vw


def shape_lib_path(nrh, nrg, nbp, nbn, exact=False):
    return os.path.join(SHAPES_DIR, f"libroborugby_amd_{nrh}x{nrg}_{nbp}x{nbn}{'_exact' if exact else ''}.so")


def shape_is_stale(nrh, nrg, nbp, nbn, exact=False):
    lib = shape_lib_path(nrh, nrg, nbp, nbn, exact)
    if not os.path.exists(lib) or not os.path.exists(lib + ".srchash"):
        return True
    with open(lib + ".srchash") as f:
        return f.read().strip() != source_hash()


def build_shape_library(nrh, nrg, nbp, nbn, force=False, verbose=False, exact=False):
    """hipcc ... -DRR_CUSTOM_SHAPE -> roborugby_amd/shapes/libroborugby_amd_<nrh>x<nrg>_<nbp>x<nbn>.so: the same sources and ABI for ONE
    shape (fp64 and fp32-state precisions, every k_step variant), one translation unit, a minute or two of hipcc for G-sized shapes."""
    vw = shape_lanes(nrh, nrg, nbp, nbn)
    lib_path = shape_lib_path(nrh, nrg, nbp, nbn, exact)
    if not force and not shape_is_stale(nrh, nrg, nbp, nbn, exact):
        return lib_path
    os.makedirs(SHAPES_DIR, exist_ok=True)
    digest = source_hash()
    tmp = lib_path + f".tmp{os.getpid()}"
    cmd = [find_hipcc()] + HIPCC_FLAGS + (["-DRR_EXACT_TRIG=1"] if exact else []) + [
        "-DRR_CUSTOM_SHAPE", "-DRR_CFG_SUBSET=9", f"-DRR_NRH={nrh}", f"-DRR_NRG={nrg}", f"-DRR_NBP={nbp}", f"-DRR_NBN={nbn}", f"-DRR_CVW={vw}",
        "-o", tmp, SRC, SRC_DQN]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, lib_path)
    with open(lib_path + ".srchash", "w") as f:
        f.write(digest + "\n")
    return lib_path


if __name__ == "__main__":  # python -m roborugby_amd.build [--force]: rebuilds what is stale (everything with --force)
    import sys
    _force = "--force" in sys.argv[1:]
    _shape = [a for a in sys.argv[1:] if a.startswith("--shape=")]  # --shape=NRH,NRG,NBP,NBN: a one-shape library instead
    if _shape:
        print(build_shape_library(*[int(x) for x in _shape[0].split("=")[1].split(",")], force=_force, verbose=True))
    else:
        print(build_hip_library(force=_force, verbose=True))
        print(build_hip_library(force=_force, verbose=True, exact=True))

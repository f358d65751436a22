"""Builds vegs_amd/_lib/libvegsrast.so (C ABI: include/vegs_rast.h) with hipcc for gfx950.

hipcc cross-compiles without a GPU.  Flags that are part of the numerics contract:
  -ffp-contract=off      every fused multiply-add in the kernels is an explicit fmaf()
  -munsafe-fp-atomics    fp32 atomicAdd lowers to the hardware global/LDS atomic, not a CAS loop
Run:  python -m vegs_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_lib")
OBJ_DIR = os.path.join(OUT_DIR, "obj")
LIB = os.path.join(OUT_DIR, "libvegsrast.so")
SOURCES = ["api.hip", "preprocess.hip", "binning.hip", "render_fwd.hip", "render_bwd.hip", "preprocess_bwd.hip", "knn.hip", "losses.hip", "optim.hip", "instances.hip", "densify.hip", "xgmi.hip"]
HEADERS = ["vr_device.h", "vr_host.h", "vr_segment.h", os.path.join("..", "..", "include", "vegs_rast.h"),
           os.path.join("..", "..", "include", "vegs_rast_debug.h"),
           os.path.join("..", "..", "include", "vegs_loss.h"), os.path.join("..", "..", "include", "vegs_optim.h"),
           os.path.join("..", "..", "include", "vegs_instances.h"), os.path.join("..", "..", "include", "vegs_xgmi.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-munsafe-fp-atomics", "-fPIC",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# Reproducer builds (profiles/experiments/README.md): the same sources with a -D switch, into their own library beside the
# product one -- loaded only when VEGS_LIB names it (vegs_amd/_capi.py).  Never built by build() without --variant.
VARIANTS = {"early": ["-DVR_EARLY_SH"], "nopack": ["-DVR_BWD_PACK_TAILS=0", "-DVR_BWD_PREFETCH=0"],
            "nopref": ["-DVR_BWD_PREFETCH=0"]}


def build_variant(name, force=False, verbose=False):
    global OBJ_DIR, LIB
    keep = (OBJ_DIR, LIB)
    OBJ_DIR = os.path.join(OUT_DIR, "obj_" + name)
    LIB = os.path.join(OUT_DIR, "libvegsrast_%s.so" % name)
    FLAGS.extend(VARIANTS[name])
    try:
        return build(force, verbose, harness=False)
    finally:
        del FLAGS[-len(VARIANTS[name]):]
        OBJ_DIR, LIB = keep


def build(force=False, verbose=False, harness=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        if force or _newer(obj, [sp] + hdrs):
            jobs.append([_hipcc(), *FLAGS, "-c", sp, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for warn in ex.map(run, jobs):
                if verbose and warn.strip():
                    print(warn)
    objs = [os.path.join(OBJ_DIR, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _newer(LIB, objs):
        run([_hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", LIB])
    if harness:
        build_c_harness(force or bool(jobs))
    return LIB


HARNESS_SRC = os.path.join(os.path.dirname(HERE), "tools", "c_harness", "vr_harness.c")
HARNESS = os.path.join(OUT_DIR, "vr_harness")


def build_c_harness(force=False):
    """tools/c_harness/vr_harness.c: a plain C11 program (gcc, no hipcc, no Python) that drives the C ABI.
    Built next to the library so that it travels to the GPU box; tests/test_gpu_c_harness.py runs it."""
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if not force and not _newer(HARNESS, [HARNESS_SRC, LIB, os.path.join(CSRC, "..", "..", "include", "vegs_rast.h")]):
        return HARNESS
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-I" + os.path.join(os.path.dirname(HERE), "include"),
           "-I" + os.path.join(rocm, "include"), HARNESS_SRC, "-o", HARNESS, "-L" + OUT_DIR, "-lvegsrast",
           "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("gcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return HARNESS


if __name__ == "__main__":
    if "--variant" in sys.argv:
        print(build_variant(sys.argv[sys.argv.index("--variant") + 1], force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))

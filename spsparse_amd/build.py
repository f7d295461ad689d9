"""Builds libspsparse_amd.so (HIP, gfx950 only) in-tree with hipcc.

    python -m spsparse_amd.build [--force]

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the
repository snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# developer experiments: SPSAMD_VARIANT=name builds lib/name/libspsparse_amd.so with SPSAMD_CXXFLAGS (capi.py loads it
# when SPSAMD_LIB points there); the shipped library is the one without a variant
VARIANT = os.environ.get("SPSAMD_VARIANT", "")
LIBDIR = os.path.join(HERE, "lib", VARIANT) if VARIANT else os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libspsparse_amd.so")
SOURCES = ["prims.hip", "consolidate.hip", "spgemm.hip", "symbolic_heavy.hip", "k_light.hip", "k_hash.hip", "k_dense.hip", "k_tiles.hip",
           "workload.hip", "capi.hip", "dist.hip"]
HEADERS = ["internal.h", "devutil.h", "spgemm_dev.h", "spgemm_host.h", "spgemm_hash.h", "workload_common.h", os.path.join("..", "..", "include", "spsparse_amd.h")]
# -ffp-contract=off: products and sums are rounded separately like the
# reference's x86-64 build (`sum += a*b`, multiply_sparse.hpp:228,235).
# -munsafe-fp-atomics: f64 atomic adds compile to ds_add_f64 / global_atomic_add_f64.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-Wall", "-Wno-unused-result"]
FLAGS += os.environ.get("SPSAMD_CXXFLAGS", "").split()      # developer experiments (-DDENSE_U=4 ...)


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc()] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

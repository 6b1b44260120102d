"""Build the HIP extension in-tree: helib_amd/lib/libhelib_amd.so (gfx950 only).

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the
gpurun snapshot, so the GPU box loads exactly what was built here.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "libhelib_amd.so")
# (source, extra flags, object): ntt_kernels.hip is compiled once per ring size, in parallel (see the top of that file)
UNITS = [("ntt_kernels.hip", ["-DHX_NTT_ONLY=13"], "ntt_kernels_13.o"), ("ntt_kernels.hip", ["-DHX_NTT_ONLY=14"], "ntt_kernels_14.o"),
         ("ntt_kernels.hip", ["-DHX_NTT_ONLY=15"], "ntt_kernels_15.o"), ("ntt_dispatch.hip", [], "ntt_dispatch.o"),
         ("conv_kernels.hip", [], "conv_kernels.o"), ("pfa_kernels.hip", [], "pfa_kernels.o"), ("rns_mfma_kernels.hip", [], "rns_mfma_kernels.o"),
         ("engine.hip", [], "engine.o")]
SOURCES = sorted({u[0] for u in UNITS})
HEADERS = ["ntt_core.h", "dev_common.h", "rns_kernels.h", "rns_types.h", "hostmath.h", "conv_core.h", "bluestein.h", "norm_kernels.h",
           "prg_kernels.h", "arena.h", "prof.h", "conv_dev.h", "ntt_kernel_util.h", "norm_r16.h", "work_map.h",
           os.path.join("..", "..", "include", "helib_amd.h")]
# headers only some units include (a change there does not rebuild the row kernels: minutes)
UNIT_HEADERS = {"pfa_kernels.hip": ["pfa_core.h", "pfa_dev.h"], "rns_mfma_kernels.hip": ["mfma_ext.h", "rns_mfma_dev.h"],
                "engine.hip": ["pfa_core.h", "pfa_dev.h", "switches.h", "mfma_ext.h", "rns_mfma_dev.h"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
         "-Wno-pass-failed"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X engine cannot be built")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, extra_flags=(), verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    deps = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    cc = hipcc()
    objs, jobs = [], []
    for s, unit_flags, o in UNITS:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, o)
        objs.append(obj)
        if force or _stale(obj, deps + [src] + [os.path.join(CSRC, h) for h in UNIT_HEADERS.get(s, [])]):
            jobs.append([cc, *FLAGS, *unit_flags, *extra_flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(SO, objs):
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, *objs])
    build_host(force=force, verbose=verbose)
    return SO


HOST_SO = os.path.join(LIBDIR, "libhelib_amd_host.so")


def build_host(force=False, verbose=False, link_dir=None, link_lib="helib_amd", out=None):
    """The C++17 host (include/helib_amd_ctxt.hpp, helib_amd_keys.hpp) as a shared library behind
    include/helib_amd_host.h: plain g++, linked against libhelib_amd.so (tests link it against their CPU
    stand-in for the C ABI instead)."""
    inc = os.path.join(HERE, "..", "include")
    src = os.path.join(CSRC, "host_session.cpp")
    out = out or HOST_SO
    deps = [src] + [os.path.join(inc, h) for h in os.listdir(inc)]
    if not (force or _stale(out, deps)):
        return out
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-I" + inc, src, "-L" + (link_dir or LIBDIR),
           "-l" + link_lib, "-Wl,-rpath," + (link_dir or "$ORIGIN"), "-o", out]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

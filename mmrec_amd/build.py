"""Build libmmrec_hip.so for gfx950 with hipcc (in-tree, so the .so travels to the GPU box).

    python -m mmrec_amd.build [--force]

hipcc cross-compiles without a GPU.  No torch headers are involved: the library is a plain C ABI
(include/mmrec_hip.h)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "libmmrec_hip.so")
SOURCES = ["api.hip", "spmm.hip", "spmm_narrow.hip", "bpr.hip", "infonce.hip", "gemm.hip", "topk.hip", "topk_filter.hip", "graph.hip", "evalsample.hip", "adam.hip", "layer_ew.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wno-unused-result"]
EXTRA_FLAGS = {}    # per-file extras: {source name: [flags]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm: /opt/rocm/bin/hipcc)")


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    deps = [os.path.join(CSRC, h) for h in ("common.h", "mfma_stream.h", "topk_sort.h", "topk_filter.h", "spmm_narrow.h")] + [
        os.path.join(PKG, "..", "include", "mmrec_hip.h")]
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _newer([src] + deps, obj):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr))
        return src

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for done in ex.map(compile_one, jobs):
            if verbose:
                print("[mmrec_amd.build] compiled", os.path.basename(done), flush=True)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _newer(objs, LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stderr))
        if verbose:
            print("[mmrec_amd.build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

"""Build libeae.so (the HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m eae_amd.build        # or: python hybrid-autoencoder-.../build.py

hipcc cross-compiles without a GPU.  Objects are cached by source mtime under csrc/_obj/.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libeae.so")
SOURCES = ["eae_api.hip", "eae_conv_launch.hip", "eae_edge_launch.hip", "eae_wgrad_launch.hip", "eae_fc_launch.hip",
           "eae_misc.hip", "eae_head.hip", "eae_mlp.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"] + os.environ.get("EAE_EXTRA_FLAGS", "").split()


def _newest_header():
    t = 0.0
    for d in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(d):
            if f.endswith((".h", ".cuh")):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def _compile(src, hdr_t, verbose):
    s = os.path.join(CSRC, src)
    o = os.path.join(OBJ, src.replace(".hip", ".o"))
    if os.path.exists(o) and os.path.getmtime(o) >= max(os.path.getmtime(s), hdr_t):
        return o, False
    cmd = ["hipcc", *FLAGS, "-c", s, "-o", o]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return o, True


def build(verbose=True, force=False):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    hdr_t = _newest_header()
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        res = list(ex.map(lambda s: _compile(s, hdr_t, verbose), SOURCES))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.exists(LIB):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

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
            if f.endswith(".h"):
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


def device_code_objects(lib=LIB):
    """The gfx950 code objects embedded in the library (clang offload bundles of the .hip_fatbin section), as bytes."""
    import re
    import struct
    d = open(lib, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", d):
        p = m.start()
        n = struct.unpack_from("<Q", d, p + 24)[0]
        o = p + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", d, o)
            o += 24
            triple = d[o:o + tl].decode(errors="replace")
            o += tl
            if "gfx950" in triple and size:
                out.append(d[p + off:p + off + size])
    return out


def scan_packed_fp32(lib=LIB, objdump="/opt/rocm/lib/llvm/bin/llvm-objdump"):
    """(number of v_pk_*_f32 instructions, those carrying an `op_sel:` modifier) in the library's device code.
    The second number must be 0: on gfx950 / ROCm 7.2 that form returned wrong low-lane results in lanes 48-63 whenever an
    MFMA kernel shared the SIMD (tools/race_mixed.py is the A/B reproducer)."""
    import re
    import tempfile
    total = opsel = 0
    for co in device_code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".o") as f:
            f.write(co)
            f.flush()
            r = subprocess.run([objdump, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True)
        for line in r.stdout.splitlines():
            if re.search(r"\bv_pk_\w+_f32\b", line):
                total += 1
                opsel += "op_sel:" in line
    return total, opsel


def scan_wide_stores_sgpr_soffset(lib=LIB, objdump="/opt/rocm/lib/llvm/bin/llvm-objdump"):
    """(number of buffer_store_dwordx3/x4 instructions, those whose soffset operand is an SGPR) in the library's device code.
    The second number must be 0: for that form hipcc (ROCm 7.2) emits no wait state between the store and a following VALU write
    of its data registers, and on gfx950 the store then sometimes ships the overwritten dword (round 4: the dy tensor a backward-data
    kernel stored differed from run to run in the third dword of pieces of channel chunks 2 and 3, whose byte offsets 128 / 192 are
    no inline constants; tools/debug_det.py).  Put such offsets into the vector offset or the 12-bit immediate instead."""
    import re
    import tempfile
    total = sgpr = 0
    for co in device_code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".o") as f:
            f.write(co)
            f.flush()
            r = subprocess.run([objdump, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True)
        for line in r.stdout.splitlines():
            m = re.search(r"\bbuffer_store_dwordx[34]\s+(.*)", line)
            if m:
                total += 1
                ops = [o.strip() for o in m.group(1).split("//")[0].split(",")]
                # vdata, vaddr (or `off`), srsrc, soffset [modifiers]
                soff = ops[3].split()[0] if len(ops) > 3 else "0"
                sgpr += bool(re.match(r"^(s\d+|m0|ttmp\d+)$", soff))
    return total, sgpr


def scan_packed_fp32_rccl(objdump="/opt/rocm/lib/llvm/bin/llvm-objdump", bundler="/opt/rocm/lib/llvm/bin/clang-offload-bundler",
                          objcopy="/opt/rocm/lib/llvm/bin/llvm-objcopy"):
    """The same census for the gfx950 code of the librccl.so that torch loads (its collectives run beside the MFMA kernels when
    EAE_DP_OVERLAP=1).  The library keeps a COMPRESSED offload bundle: .hip_fatbin is dumped, unbundled for gfx950 and
    disassembled in a stream (277 MB of code: about 3 minutes, 1 GB of scratch files under $TMPDIR).
    Returns (v_pk_*_f32 total, with `op_sel:`, with `op_sel_hi:`).  torch 2.10.0+rocm7.0: (325, 0, 168)."""
    import re
    import tempfile
    import torch
    lib = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "rccl.fatbin"), os.path.join(d, "rccl_gfx950.co")
        subprocess.run([objcopy, "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(d, "dummy.so")], check=True)
        subprocess.run([bundler, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True)
        total = opsel = opsel_hi = 0
        with subprocess.Popen([objdump, "-d", "--mcpu=gfx950", co], stdout=subprocess.PIPE, text=True) as pr:
            for line in pr.stdout:
                if re.search(r"\bv_pk_\w+_f32\b", line):
                    total += 1
                    opsel += "op_sel:" in line
                    opsel_hi += "op_sel_hi:" in line
    return total, opsel, opsel_hi


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

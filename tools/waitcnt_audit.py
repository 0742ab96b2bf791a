#!/usr/bin/env python3
"""Static audit of the vector-memory pipelining hipcc actually emitted (no GPU needed).

    python tools/waitcnt_audit.py [file.hip ...]        # default: every translation unit of csrc/

For every kernel: the `s_waitcnt vmcnt(N)` values found INSIDE loops next to the number of vector-memory loads the loop body
issues.  A software-pipelined loop that is meant to keep a tile in flight must wait with N >= (loads of one tile); a loop whose
waits all reach vmcnt(0) drains its prefetch every iteration (round 3's weight-gradient producers did: hipcc merges the
s_waitcnt state of every path into a loop header pessimistically, and a prefetch issued under `if (t + k < n)` made "nothing
newer in flight" one of those paths).  Also lists scratch traffic inside loops (a spill reload is a vector-memory operation and
waits in issue order behind the prefetch).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hybrid-autoencoder-mlp-pipeline-for-satellite-image-classification_amd")
CSRC = os.path.join(PKG, "csrc")


def asm_of(src):
    out = os.path.join(tempfile.gettempdir(), "eae_audit_" + os.path.basename(src) + ".s")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))):
        flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S"]
        flags += os.environ.get("EAE_EXTRA_FLAGS", "").split()
        subprocess.run(["hipcc", *flags, "-o", out, src], check=True, stderr=subprocess.DEVNULL)
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.split("\n")


def audit(path):
    rows = []
    cur, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is None:
            continue
        body.append(line)
        if line.startswith(".Lfunc_end"):
            rows.append((cur, body))
            cur = None
    out = []
    for name, body in rows:
        if not any(".amdhsa_kernel" in l or "s_endpgm" in l for l in body):
            continue
        # loop regions: basic blocks whose label comment says "in Loop" / "Loop Header"
        in_loop = False
        loops = {}          # header id -> dict
        key = None
        for l in body:
            lm = re.match(r"^\.LBB\d+_\d+:\s*;(.*)$", l)
            if lm:
                c = lm.group(1)
                hm = re.search(r"Header=(BB\d+_\d+)", c)
                if "Loop Header" in c:
                    key = re.match(r"^\.L(BB\d+_\d+)", l).group(1)
                    in_loop = True
                elif hm:
                    key = hm.group(1)
                    in_loop = True
                else:
                    in_loop = False
                if in_loop:
                    loops.setdefault(key, dict(waits=[], loads=0, scratch=0, mfma=0, barriers=0))
                continue
            if re.match(r"^\.LBB\d+_\d+:", l):
                in_loop = False
                continue
            if not in_loop:
                continue
            d = loops[key]
            w = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
            if w:
                d["waits"].append(int(w.group(1)))
            if re.search(r"\b(buffer_load|global_load|flat_load)", l):
                d["loads"] += 1
            if "scratch_" in l:
                d["scratch"] += 1
            if "v_mfma" in l:
                d["mfma"] += 1
            if "s_barrier" in l:
                d["barriers"] += 1
        regs = {}
        for l in body:
            for k in ("NumVgprs", "ScratchSize", "Occupancy"):
                m = re.match(r"^; %s: (\d+)" % k, l)
                if m:
                    regs[k] = int(m.group(1))
        out.append((name, regs, loops))
    return out


def main():
    srcs = sys.argv[1:] or [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    bad = 0
    for s in srcs:
        res = audit(asm_of(s))
        names = demangle([n for n, _, _ in res])
        print("== " + os.path.basename(s))
        for (n, regs, loops), dn in zip(res, names):
            dn = re.sub(r"\(.*$", "", dn)
            for k, d in loops.items():
                if d["loads"] == 0:
                    continue
                drains = d["waits"].count(0)
                flag = "DRAINS" if drains and d["loads"] >= 4 else ""
                if d["scratch"]:
                    flag += " SCRATCH-IN-LOOP"
                if flag:
                    bad += 1
                print("  %-78s loop %-9s loads %3d mfma %3d barriers %d vmcnt %s  [vgpr %s scratch %s] %s" % (
                    dn[:78], k, d["loads"], d["mfma"], d["barriers"], d["waits"], regs.get("NumVgprs"), regs.get("ScratchSize"), flag))
    print("%d loop(s) flagged" % bad)


if __name__ == "__main__":
    main()

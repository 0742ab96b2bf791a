#!/usr/bin/env python3
"""A/B of library variants on ONE box (boxes differ by up to 7 %): interleaved bench.py runs per variant, then (--prof) one
rocprofv3 --kernel-trace --stats pass per variant with the per-kernel average durations side by side.

    tools/ab.py [--prof] [--rounds 3] [--steps 300] [--batch 512] name=path/to/libeae_x.so[;ENV=VAL...] ...

`path` may be "-" for the product library.  Diagnostic tool (GPU box)."""
import csv
import glob
import json
import os
import re
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
prof = "--prof" in args
args = [a for a in args if a != "--prof"]


def opt(name, default):
    if name in args:
        i = args.index(name)
        v = args[i + 1]
        del args[i:i + 2]
        return v
    return default


rounds, steps, batch = int(opt("--rounds", 3)), opt("--steps", "300"), opt("--batch", "512")
variants = []
for a in args:
    name, rest = a.split("=", 1)
    parts = rest.split(";")
    env = dict(p.split("=", 1) for p in parts[1:])
    if parts[0] != "-":
        env["EAE_LIB_PATH"] = os.path.abspath(parts[0])
    variants.append((name, env))
cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "20", "--batch", batch, "--no-cpu-baseline", "--no-configs", "--no-roofline"]
res = {n: [] for n, _ in variants}
for r in range(rounds):
    for n, env in variants:
        out = subprocess.run(cmd, env={**os.environ, **env}, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(n, "FAILED", out.stderr[-400:])
            continue
        res[n].append(json.loads(line[-1])["ms_per_step"])
for n, v in res.items():
    if v:
        print(f"{n:24s} ms/step median {statistics.median(v):.4f}  min {min(v):.4f}  all {v}", flush=True)
if prof:
    tabs = {}
    for n, env in variants:
        d = f"/tmp/ab_prof_{n}"
        subprocess.run(["rm", "-rf", d])
        e = {**os.environ, **env, "TMPDIR": "/tmp"}
        subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "t", "--", sys.executable,
                        os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "10", "--batch", batch, "--no-cpu-baseline", "--no-configs", "--no-roofline"],
                       env=e, cwd="/tmp", capture_output=True, text=True)
        f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
        if not f:
            print(n, "no stats"); continue
        tabs[n] = {r["Name"]: (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(f[0]))}
        out = os.path.join(ROOT, "gpurun_out", f"ab_stats_{n}.csv")
        subprocess.run(["cp", f[0], out])
        tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
        if tr:
            subprocess.run(["cp", tr[0], os.path.join(ROOT, "gpurun_out", f"ab_trace_{n}.csv")])
            tl = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "timeline.py"), tr[0]], capture_output=True, text=True).stdout
            open(os.path.join(ROOT, "gpurun_out", f"ab_timeline_{n}.txt"), "w").write(tl)
    names = sorted(set().union(*[set(t) for t in tabs.values()]), key=lambda k: -max(t.get(k, (0, 0))[0] * t.get(k, (0, 0))[1] for t in tabs.values()))
    print(f"{'kernel':70s} " + " ".join(f"{n:>12s}" for n in tabs))
    for k in names[:45]:
        short = re.sub(r"\(.*", "", k).replace("void ", "")[:70]
        print(f"{short:70s} " + " ".join(f"{tabs[n].get(k, (0, 0))[0]:12.1f}" for n in tabs))

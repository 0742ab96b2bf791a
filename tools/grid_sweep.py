#!/usr/bin/env python3
"""Sweep of the B=64 grid leg (bench.grid_b64_leg): K concurrent configurations x hardware queues x graph replay.  Each point runs in a
fresh process (GPU_MAX_HW_QUEUES is read when HIP initialises).  Diagnostic tool (GPU box).

    python tools/grid_sweep.py [K,K,...] [Q,Q,...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ks = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4,8").split(",")]
qs = [v for v in (sys.argv[2] if len(sys.argv) > 2 else "4,8").split(",")]
code = "import json, sys; sys.path.insert(0, %r); import bench; print('RES', json.dumps(bench.grid_b64_leg(k=int(sys.argv[1]))))" % ROOT
for q in qs:
    for graph in ("0", "1"):
        for k in ks:
            env = {**os.environ, "GPU_MAX_HW_QUEUES": q, "EAE_GRID_GRAPH": graph}
            out = subprocess.run([sys.executable, "-c", code, str(k)], env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("RES ")]
            if not line:
                print(f"queues {q} graph {graph} K {k}: FAILED {out.stderr[-300:]}", flush=True)
                continue
            r = json.loads(line[-1][4:])
            print(f"queues {q} graph {graph} K {k:2d}: {r['concurrent']['images_per_s']:10.0f} img/s  (k1 {r['k1']['images_per_s']:.0f}, "
                  f"{r['concurrent']['ms_per_step_per_config']} ms per step per config)", flush=True)

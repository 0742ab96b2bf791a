#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   -> ms/step and roofline-kernel time of bench.py for each value of the env var
var=$1; shift
for v in "$@"; do
  out=$(env "$var=$v" timeout -k 10 200 python bench.py --no-cpu-baseline 2>&1 | tail -1)
  echo "$var=$v $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["avg_launch_us"])')"
done

for w in 64 128 192 256 "256,128" ; do
  for wl in c5bf16; do
    r=$(EAE_WGRAD_WGS=$w python bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$wl WGS=$w ms=$r"
  done
done

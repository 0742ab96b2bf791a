for cfg in "EAE_IG_MT=1" "EAE_IG_MT=0" "EAE_IG_MT=1 EAE_IG_WGS_PER_CU=1" "EAE_IG_MT=1 EAE_IG_WGS_PER_CU=4"; do
  for wl in c5bf16; do
    r=$(env $cfg python bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$wl $cfg ms=$r"
  done
done

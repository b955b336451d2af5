for cfg in "X=1" "GNNB_NO_TOP=1" "GNNB_NO_TOP=1 GNNB_NO_DENSE_LDS=1"; do
  for b in 2 16 64 128; do
    r=$(env $cfg python bench.py --batch $b --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])")
    echo "$cfg B=$b -> $r ms"
  done
done

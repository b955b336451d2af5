for b in 1 2 16 64; do
  python bench.py --batch $b --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('B=$b', d['ms_per_step'], 'ms', round(d['value']/1e6,2), 'M scores/s', 'instrumented', d['instrumented_ms_per_step'])"
done
python bench.py --batch 2 --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/b2.json 2>/dev/null; python tools/kern_table.py gpurun_out/b2.json

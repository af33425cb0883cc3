# A/B: the write pass walks the ticks in reverse (the default; LSN_WRITE_FORWARD=1 = first to last) -- does the depth the count pass read last come from cache?
run() { timeout -k 10 200 python bench.py --core-only --steps 400 --warmup 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), 'ms/step  kernel', round(d['roofline']['kernel_avg_ms'],4), 'ms  frac', round(d['roofline']['frac'],4), 'step_frac', round(d['roofline']['step_frac'],4))"; }
for v in 0 1 1 0; do LSN_WRITE_FORWARD=$((1-v)) run "reverse=$v"; done
timeout -k 10 300 python -m pytest tests/test_fusion_gpu.py -m gpu -x -q -k "device_resident or full_size" 2>&1 | tail -2

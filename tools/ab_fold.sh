run() { timeout -k 10 200 python bench.py --core-only --steps 400 --warmup 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), 'ms/step  kernel', round(d['roofline']['kernel_avg_ms'],4), 'ms  step_frac', round(d['roofline']['step_frac'],4))"; }
for v in 0 1 1 0; do LSN_FOLD_SCAN=$v run "fold_scan=$v"; done
LSN_FOLD_SCAN=1 timeout -k 10 300 python -m pytest tests/test_fusion_gpu.py tests/test_thresholds_gpu.py -m gpu -x -q 2>&1 | tail -2
ICP_REPS=1 LSN_ICP_DEBUG=1 timeout -k 10 100 python3 tools/icp_driver.py 2>&1 | grep "lsn icp" | head -12
ICP_SENSORS=8 ICP_REPS=1 LSN_ICP_DEBUG=1 timeout -k 10 100 python3 tools/icp_driver.py 2>&1 | grep "lsn icp" | head -12

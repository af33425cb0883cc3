set -e
mkdir -p gpurun_out/r04
run() { timeout -k 10 200 python bench.py --core-only --steps 400 --warmup 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), 'ms/step  kernel', round(d['roofline']['kernel_avg_ms'],4), 'ms  frac', round(d['roofline']['frac'],4), 'step_frac', round(d['roofline']['step_frac'],4))"; }
for v in 3 4 31 3 4 31; do
  make -C livescan3d_amd/csrc -j12 EXTRA=-DLSN_STAGE_PAD_SHIFT=$v build/fusion.o build/exchange.o > /dev/null 2>&1 || true
  touch livescan3d_amd/csrc/fusion_shared.hpp
  make -C livescan3d_amd/csrc -j12 EXTRA=-DLSN_STAGE_PAD_SHIFT=$v > /dev/null 2>&1
  run "pad_shift=$v"
done

# Dev helper (GPU box): the look-back window of the single-pass kernel, A/B/B/A: LSN_LOOK_SLOTS = 1 (round 4: 64 predecessors per round trip), 4, 8, 16
set -e
O=gpurun_out/r05f; mkdir -p $O
for v in 1 8 4 16 8 1; do
  touch livescan3d_amd/csrc/fusion.hip
  make -C livescan3d_amd/csrc -j12 EXTRA=-DLSN_LOOK_SLOTS=$v > /dev/null 2>&1
  echo "== LSN_LOOK_SLOTS=$v"; timeout -k 10 200 python3 tools/one_tick_driver.py 2>/dev/null
done
touch livescan3d_amd/csrc/fusion.hip; make -C livescan3d_amd/csrc -j12 > /dev/null 2>&1

# Dev helper (GPU box): waves per SIMD of the triangle count pass, A/B/A/B on the chained scene tick
O=gpurun_out/r05h; mkdir -p $O
for v in 5 7 8 5 7 8; do
  touch livescan3d_amd/csrc/mesh.hip
  make -C livescan3d_amd/csrc -j12 EXTRA=-DLSN_TRI_MIN_WAVES=$v > /dev/null 2>&1
  echo "== LSN_TRI_MIN_WAVES=$v"; bash tools/prof.sh r05h/tick_$v 3 python3 tools/tick_driver.py scene 64 6 2>&1 | grep -E "tri_kernel<0|radial_band"; tail -12 gpurun_out/r05h/tick_$v.log | grep "ticks:" | tail -2
done
touch livescan3d_amd/csrc/mesh.hip; make -C livescan3d_amd/csrc -j12 > /dev/null 2>&1

# Dev helper (GPU box): radial_band_kernel's workgroup size x rows per band, A/B on the chained scene tick (64 ticks = 512 frames per launch)
O=gpurun_out/r06; mkdir -p $O
for v in ${AB_VARIANTS:-"512 12" "256 6" "256 8" "256 12" "512 8" "1024 12" "512 12"}; do set -- $v
  touch livescan3d_amd/csrc/radial.hip
  make -C livescan3d_amd/csrc -j12 EXTRA="-DLSN_BAND_THREADS=$1" > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== threads=$1 rows=$2: $(LSN_RADIAL_BAND_ROWS=$2 bash tools/prof.sh r06/band_$1_$2 12 python3 tools/tick_driver.py scene 64 6 2>&1 | grep -E "radial_band" | awk '{print $3, $4, $5}') $(grep 'ticks:' $O/band_$1_$2.log | tail -3 | awk '{printf "%s ", $4}')"
done
touch livescan3d_amd/csrc/radial.hip; make -C livescan3d_amd/csrc -j12 > /dev/null 2>&1

# Dev helper (GPU box): radial_band_kernel knobs on the chained scene tick: AB_VARIANTS="threads,rows,fly ..."
O=gpurun_out/r06; mkdir -p $O
for v in ${AB_VARIANTS:-256,6,6 256,6,4 256,6,8 256,6,3 256,6,12 256,6,6}; do
  t=${v%%,*}; rest=${v#*,}; r=${rest%%,*}; f=${rest#*,}
  touch livescan3d_amd/csrc/radial.hip
  make -C livescan3d_amd/csrc -j12 EXTRA="-DLSN_BAND_THREADS=$t -DLSN_BAND_FLY=$f $AB_EXTRA" > /dev/null 2>&1 || { echo build failed; exit 1; }
  LSN_RADIAL_BAND_ROWS=$r bash tools/prof.sh r06/band_${t}_${r}_$f 12 python3 tools/tick_driver.py scene 64 6 > $O/band_${t}_${r}_$f.txt 2>&1
  echo "threads=$t rows=$r fly=$f: band $(grep radial_band $O/band_${t}_${r}_$f.txt | grep -oE 'avg_us= *[0-9.]+') tick $(grep ticks: $O/band_${t}_${r}_$f.log | tail -2 | awk '{printf "%s ", $4}')"
done
touch livescan3d_amd/csrc/radial.hip; make -C livescan3d_amd/csrc -j12 > /dev/null 2>&1

# Dev helper (GPU box): compile-time knobs of the radial stage on the chained scene tick; AB_VARIANTS="name:flags ..."
O=gpurun_out/r06; mkdir -p $O
for v in ${AB_VARIANTS:-base: spec:-DLSN_BAND_SPEC=true fix128:-DLSN_FIX_THREADS=128 fix512:-DLSN_FIX_THREADS=512 base:}; do
  n=${v%%:*}; f=${v#*:}
  touch livescan3d_amd/csrc/radial.hip
  make -C livescan3d_amd/csrc -j12 EXTRA="$f" > /dev/null 2>&1 || { echo "$n: build failed"; continue; }
  bash tools/prof.sh r06/knob_$n 12 python3 tools/tick_driver.py scene 64 6 > $O/knob_$n.txt 2>&1
  echo "$n [$f]: band $(grep radial_band $O/knob_$n.txt | grep -oE 'avg_us= *[0-9.]+') fix $(grep 'close_fix_kernel' $O/knob_$n.txt | grep -oE 'avg_us= *[0-9.]+') tick $(grep ticks: $O/knob_$n.log | tail -3 | awk '{printf "%s ", $4}')"
done
touch livescan3d_amd/csrc/radial.hip; make -C livescan3d_amd/csrc -j12 > /dev/null 2>&1

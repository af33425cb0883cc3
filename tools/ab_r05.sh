# Dev helper (GPU box): this tree's ICP against the round-5 library (livescan3d_amd/lib/libNativeUtils_r05.so, built from git HEAD~), same box
for sens in 2 8; do
  for v in r05 0 1 2 r05 1; do
    if [ $v = r05 ]; then lib=$PWD/livescan3d_amd/lib/libNativeUtils_r05.so; near=0; else lib=; near=$v; fi
    echo "sensors=$sens lib=${v}: $(ICP_SENSORS=$sens ICP_REPS=5 LSN_NATIVE_LIB=$lib LSN_ICP_NEAR=$near timeout -k 10 120 python tools/icp_driver.py 2>&1 | grep -E 'ms/iter|settled' | awk '{printf "%s ", ($1=="n1")?substr($6,1,7):$4}')"
  done
done

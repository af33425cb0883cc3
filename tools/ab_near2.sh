# Dev helper (GPU box): the cull kernel's lanes per query and loads in flight, per-kernel times on configs[1] / configs[2]
O=gpurun_out/r06; mkdir -p $O
for v in ${AB_VARIANTS:-"4 8" "8 8" "8 4" "4 16"}; do set -- $v
  touch livescan3d_amd/csrc/icp.hip
  make -C livescan3d_amd/csrc -j12 EXTRA="-DLSN_CULL_TEAM=$1 -DLSN_NEAR_BATCH=$2" > /dev/null 2>&1 || { echo build failed; exit 1; }
  for sens in 2 8; do for near in 2 1 0; do
    echo "== team=$1 batch=$2 sensors=$sens near=$near"
    ICP_SENSORS=$sens LSN_ICP_NEAR=$near bash tools/prof.sh r06/ab_t$1_b$2_s${sens}_n$near 8 python3 tools/icp_driver.py 2>&1 | grep -E "nn_cull_kernel<true|nn_scan|nn_blocks|nn_finish" | cut -c1-50,100-
    grep -E "ms/iter|settled" $O/ab_t$1_b$2_s${sens}_n$near.log | tail -3 | awk '{printf "%s ", ($1=="n1")?$6:$4} END {print ""}'
  done; done
done
touch livescan3d_amd/csrc/icp.hip; make -C livescan3d_amd/csrc -j12 > /dev/null 2>&1

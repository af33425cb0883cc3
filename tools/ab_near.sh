# A/B of the ICP near path (LSN_ICP_NEAR=0/1) and its candidate cap on configs[1] (2 sensors) and configs[2] (8 sensors)
for sens in 2 8; do
  for v in "1 128" "0 128" "1 32" "1 64" "1 256" "1 128" "0 128"; do set -- $v
    echo "sensors=$sens near=$1 cap=$2: $(ICP_SENSORS=$sens ICP_REPS=4 LSN_ICP_NEAR=$1 LSN_ICP_NEAR_PTS=$2 timeout -k 10 120 python tools/icp_driver.py 2>&1 | grep -E 'ms/iter|settled' | awk '{printf "%s ", ($1=="n1")?$6:$4}')"
  done
done

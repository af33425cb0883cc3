# Dev helper (GPU box): ICP ms/iteration (configs[1] / configs[2]) under environment variants; AB_VARIANTS="VAR=a VAR=b ..."
for v in ${AB_VARIANTS}; do
  for sens in 2 8; do
    echo "$v sensors=$sens: $(env $v ICP_SENSORS=$sens ICP_REPS=5 timeout -k 10 120 python tools/icp_driver.py 2>&1 | grep -E 'ms/iter' | awk '{printf "%s ", substr($6,1,7)}')"
  done
done

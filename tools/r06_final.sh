#!/bin/bash
# Round-6 evidence run (GPU box), part $1 = a | b.  a: PMC traffic of the fusion kernels (so that the bench lines carry `traffic`), the PMC pass over
# the chained scene tick with the stall counters (full_tick.kernels), the two bench commands.  b: rocprofv3 kernel stats of the timed region, of the
# chained tick, of the host merge call, of ICP at configs[1] and configs[2], of the 1024x1024 shapes; the sharded host flow's rehearsal.
# Everything lands under gpurun_out/r06f/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06f; mkdir -p $O
cd $R
if [ "$1" = a ]; then
  bash tools/pmc.sh r06 0 > $O/pmc_fusion.txt 2>&1
  cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null; cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
  echo "== pmc fusion done"; tail -2 $O/pmc_fusion.txt | cut -c1-300
  bash tools/pmc_tick.sh r06f scene "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH SQ_INSTS_SMEM" > $O/pmc_tick.txt 2>&1
  cp gpurun_out/pmc_tick_r06f/summary.json profiles/pmc_tick_scene.json; cp gpurun_out/pmc_tick_r06f/summary.json $O/pmc_tick_scene.json
  echo "== pmc tick done"; grep -c . $O/pmc_tick.txt
  timeout -k 10 500 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "== bench full rc=$?"; python3 tools/line_summary.py $O/bench_full.json
  ( time timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err ) 2> $O/bench_driver_command.time; echo "== bench driver cmd rc=$?"; grep real $O/bench_driver_command.time
  python3 tools/line_summary.py $O/bench_driver_command.json
  rm -rf gpurun_out/pmc_r06 gpurun_out/pmc_tick_r06f
else
  prof() { tag=$1; shift; cd /tmp && export TMPDIR=/tmp && cd $R; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- "$@" > $O/$tag.out 2> $O/$tag.err; python3 tools/prof_stats.py $O/$tag 9 | tee $O/${tag}_stats.txt; cp $(ls -t $O/$tag/*/*kernel_stats.csv | head -1) $O/rocprofv3_kernel_stats_$tag.csv; rm -rf $O/$tag; }
  prof core python3 bench.py --core-only; cp $O/core.out $O/bench_core_under_rocprof.json
  prof tick_scene python3 tools/tick_driver.py scene 64 6; grep ticks: $O/tick_scene.out | tail -3
  prof host_merge_scene python3 tools/host_trace.py scene 40
  ICP_SENSORS=2 prof icp_configs1 python3 tools/icp_driver.py; grep -E "ms/iter|settled" $O/icp_configs1.out | tail -3
  ICP_SENSORS=8 prof icp_configs2 python3 tools/icp_driver.py; grep -E "ms/iter|settled" $O/icp_configs2.out | tail -3
  prof cfg4_shape python3 tools/count_driver.py 8 16 1024 1024; tail -1 $O/cfg4_shape.out
  bash tools/ab_r05.sh > $O/ab_icp_vs_round5.txt 2>&1; cat $O/ab_icp_vs_round5.txt
  for v in r05 new; do if [ $v = r05 ]; then lib=$PWD/livescan3d_amd/lib/libNativeUtils_r05.so; else lib=; fi; echo "$v: $(LSN_NATIVE_LIB=$lib timeout -k 10 200 python tools/refine_driver.py 2>&1 | tail -1)"; done | tee $O/refine_vs_round5.txt
  LSN_HOST_DEVICES=0,0 timeout -k 10 120 python3 tools/host_path.py > $O/host_path_two_parts_one_gpu.txt 2>&1; grep -v "^{" $O/host_path_two_parts_one_gpu.txt | tail -7
fi

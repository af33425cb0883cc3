#!/bin/bash
# Round-5 evidence run (GPU box): PMC traffic of the fusion kernels first (so that the bench lines carry `traffic`), the PMC pass over the chained
# scene tick (full_tick.kernels), the two bench commands, rocprofv3 kernel stats of the timed region, of the chained tick, of the host merge call
# and of ICP.  Everything lands under gpurun_out/r05f/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O
cd $R
bash tools/pmc.sh r05 0 > $O/pmc_fusion.txt 2>&1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null; cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
echo "== pmc fusion done"; tail -2 $O/pmc_fusion.txt | cut -c1-300
bash tools/pmc_tick.sh r05 scene > $O/pmc_tick.txt 2>&1
cp gpurun_out/pmc_tick_r05/summary.json profiles/pmc_tick_scene.json; cp gpurun_out/pmc_tick_r05/summary.json $O/pmc_tick_scene.json
echo "== pmc tick done"
timeout -k 10 500 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "== bench full rc=$?"
( time timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err ) 2> $O/bench_driver_command.time; echo "== bench driver cmd"; grep real $O/bench_driver_command.time
cd /tmp && export TMPDIR=/tmp && cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/core -- python3 bench.py --core-only > $O/bench_core_under_rocprof.json 2> $O/core.err; echo "== core profile rc=$?"
python3 tools/prof_stats.py $O/core 6 | tee $O/core_stats.txt
cp $(ls $O/core/*/*kernel_stats.csv | tail -1) $O/rocprofv3_kernel_stats_core.csv
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tick -- python3 tools/tick_driver.py scene 64 6 > /dev/null 2> $O/tick.err; python3 tools/prof_stats.py $O/tick 8 | tee $O/tick_stats.txt
cp $(ls $O/tick/*/*kernel_stats.csv | tail -1) $O/rocprofv3_kernel_stats_tick_scene.csv
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/host -- python3 tools/host_trace.py scene 40 > /dev/null 2> $O/host.err; python3 tools/prof_stats.py $O/host 8 | tee $O/host_stats.txt
cp $(ls $O/host/*/*kernel_stats.csv | tail -1) $O/rocprofv3_kernel_stats_host_merge_scene.csv
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/icp -- python3 tools/icp_driver.py > /dev/null 2> $O/icp.err; python3 tools/prof_stats.py $O/icp 8 | tee $O/icp_stats.txt
cp $(ls $O/icp/*/*kernel_stats.csv | tail -1) $O/rocprofv3_kernel_stats_icp_configs1.csv
LSN_HOST_DEVICES=0,0 timeout -k 10 120 python3 tools/host_path.py > $O/host_path_two_parts_one_gpu.txt 2>&1; grep -v "^{" $O/host_path_two_parts_one_gpu.txt | tail -7
rm -rf $O/core $O/host $O/icp $O/tick gpurun_out/pmc_r05 gpurun_out/pmc_tick_r05

#!/bin/bash
# Dev helper (GPU box): PMC passes for the kernels of the multi-GPU step (pack_kernel / recon_kernel), driven through
# tools/shard_driver.py (the two kernels alone, numpy-generated frames); usage: tools/pmc_shard.sh <tag>
# (Round 2 noted that a --pmc pass "crashes the process when torch.distributed is initialised": the library then dlopen()ed the system's
# librccl.so.1 next to the librccl.so torch had already mapped -- two RCCL instances in one profiled process.  Since rccl() prefers the
# mapped instance, `LSN_BENCH_FORCE_DIST=1 rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --core-only ...` completes: round-3 log kept as
# profiles/r03_pmc_pass_with_torch_distributed.log.)
# Counters are collected in separate runs (TCC slot limits; FETCH_SIZE and WRITE_SIZE cannot share a pass).
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_shard_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/$name -- python3 $GRAFT_REPO_ROOT/tools/shard_driver.py > $out/$name.log 2>&1 || echo "pass $name failed: $(tail -1 $out/$name.log)"
done
python3 - <<PY
import csv, glob, collections, json
summary = collections.defaultdict(dict)
for d in sorted(glob.glob("$out/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(s in k for s in ("pack_kernel", "recon_kernel", "count_thr", "tick_base")): continue
            name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                summary[k][c] = round(sum(v) / len(v), 1)
                summary[k]["dispatches"] = len(v)
json.dump(summary, open("$out/summary.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(summary.items()):
    print(k, v)
PY

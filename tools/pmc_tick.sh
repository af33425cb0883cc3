#!/bin/bash
# Dev helper (GPU box): PMC passes over the kernels of the chained tick (tools/tick_driver.py, 16 ticks = 128 frames per launch);
# usage: tools/pmc_tick.sh <tag> [scene|noise] [extra counter set ...].  Separate runs per counter set (TCC slot limits).
tag=$1; kind=${2:-scene}; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_tick_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sets=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "$@")
for set in "${sets[@]}"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/$name -- python3 tools/tick_driver.py $kind 16 2 > $out/$name.log 2>&1 || echo "pass $name failed: $(tail -1 $out/$name.log)"
done
python3 - <<PY
import csv, glob, collections, json
summary = collections.defaultdict(dict)
for d in sorted(glob.glob("$out/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(s in k for s in ("tri_kernel", "radial_", "close_", "fuse_kernel", "count_thr")): continue
            name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                summary[k][c] = round(sum(v) / len(v), 1)
                summary[k]["dispatches"] = len(v)
import hashlib
hsh = hashlib.sha256()
for f in ("mesh.hip", "radial.hip", "fusion.hip", "fusion_shared.hpp"):
    hsh.update(open("livescan3d_amd/csrc/" + f, "rb").read())
summary["_sources_sha256"] = hsh.hexdigest()     # bench.py reports these counters only while the kernels are the ones they were read from
summary["_workload"] = "tools/tick_driver.py $kind 16: 16 ticks x 8 x 512x424 $kind frames per launch (128 frames, 27.8 M pixels), radial -> vertices -> triangles"
json.dump(summary, open("$out/summary.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(summary.items()):
    print(k, v)
PY

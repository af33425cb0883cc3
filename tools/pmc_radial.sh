#!/bin/bash
# Dev helper (GPU box): PMC passes for the radial-correction kernels (tools/radial_driver.py, scene frames, 16 ticks = 128 frames per launch);
# usage: tools/pmc_radial.sh <tag> [extra counter set ...].  Separate runs per counter set (TCC slot limits).
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_radial_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sets=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "$@")
for set in "${sets[@]}"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/$name -- python3 tools/radial_driver.py scene 16 2 > $out/$name.log 2>&1 || echo "pass $name failed: $(tail -1 $out/$name.log)"
done
python3 - <<PY
import csv, glob, collections, json
summary = collections.defaultdict(dict)
for d in sorted(glob.glob("$out/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(s in k for s in ("radial_", "close_")): continue
            name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                summary[k][c] = round(sum(v) / len(v), 1)
                summary[k]["dispatches"] = len(v)
summary["_workload"] = "tools/radial_driver.py scene 16: 16 ticks x 8 x 512x424 scene frames per launch (128 frames, 27.8 M pixels: 139 MB in, 139 MB out), in place then out of place"
json.dump(summary, open("$out/summary.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(summary.items()):
    print(k, v)
PY

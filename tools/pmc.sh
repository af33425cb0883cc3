#!/bin/bash
# Dev helper (GPU box): PMC passes for the fusion kernels; usage: tools/pmc.sh <tag> <mode>
# Counters are collected in separate runs (TCC slot limits; FETCH_SIZE and WRITE_SIZE cannot share a pass).
tag=$1; mode=$2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/$name -- python3 $GRAFT_REPO_ROOT/tools/pmc_driver.py $mode > $out/$name.log 2>&1 || echo "pass $name failed: $(tail -1 $out/$name.log)"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$out/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "fuse_kernel" not in k and "scan_kernel" not in k and "count_thr" not in k: continue
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
PY
grep -h "algorithmic_bytes" $out/*.log | head -1
python3 - <<PY
import csv, glob, json, os
def avg(counter, kernel):
    vals = []
    for f in glob.glob("$out/" + counter + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter: vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None
kernel = "fuse_kernel<1" if "$mode" == "0" else "run_kernel"
fetch, write = avg("FETCH_SIZE", kernel), avg("WRITE_SIZE", kernel)
if fetch is not None and write is not None:
    path = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_traffic.json")
    d = json.load(open(path)) if os.path.exists(path) else {}
    T = int(os.environ.get("PMC_TICKS", "64"))
    import hashlib
    hsh = hashlib.sha256()
    for f in ("fusion.hip", "fusion_shared.hpp"):
        hsh.update(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "livescan3d_amd", "csrc", f), "rb").read())
    d[f"mode$mode-8x512x424-ticks{T}"] = {"kernel": kernel, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "kernel_sources_sha256": hsh.hexdigest(),
        "hbm_bytes_per_launch": int((2 * fetch + write) * 1024), "note": "FETCH_SIZE doubled (gfx950 reports half of a wide coalesced read); WRITE_SIZE as read"}
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)
    print("traffic", d)
PY

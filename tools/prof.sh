#!/bin/bash
# Dev helper (GPU box): kernel trace + stats of one driver.  usage: tools/prof.sh <tag> <rows> python3 tools/<driver>.py args...
# (the program itself follows the tag: rocprofv3 must start python directly)
tag=$1; rows=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- "$@" > $out.log 2>&1 || { echo "profile failed"; tail -5 $out.log; exit 1; }
cd $GRAFT_REPO_ROOT && python3 tools/prof_stats.py $out $rows

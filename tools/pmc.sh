#!/bin/bash
# usage: tools/pmc.sh <outdir> <kernel-substring> "<counters>" -- <program args...>   (bounded by timeout)
out=$1; shift; kern=$1; shift; ctrs=$1; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 150 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/$out -- "$@" > gpurun_out/$out.log 2>&1
echo "rc=$?"
python - <<PY
import csv, glob, collections
fs = glob.glob("gpurun_out/$out/*/*counter_collection.csv")
if fs:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "$kern" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("$out", k, "%.5g" % (sum(v[1:]) / max(1, len(v) - 1)), "n=%d" % len(v))
PY

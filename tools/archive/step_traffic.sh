#!/bin/bash
# HBM traffic of one train step from the TCC counters (GPU box, repo root): two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel
# trace only) over tools/train_loop.py, summed per kernel name and divided by the number of steps.  -> gpurun_out/<tag>_step_traffic.txt
tag=${1:-r03}; steps=6
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/st_$c -- python3 tools/train_loop.py $steps > gpurun_out/st_$c.log 2>&1 || exit 1
done
python3 - <<PY > gpurun_out/${tag}_step_traffic.txt
import csv, glob, collections, re
steps = $steps + 2          # train_loop.py runs two warm-up steps
tot = {}
per = collections.defaultdict(lambda: [0.0, 0.0, 0])
for i, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    f = glob.glob("gpurun_out/st_%s/*/*counter_collection.csv" % c)[0]
    s = 0.0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        v = float(r["Counter_Value"]) * 1024 * (2 if c == "FETCH_SIZE" else 1)      # KB; gfx950: 16-byte-per-lane reads tallied at half
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
        per[k][i] += v
        per[k][2] += (i == 0)
        s += v
    tot[c] = s
print("HBM traffic per train step (hg2 + DSNT + JS, batch 32) from the TCC counters: FETCH_SIZE x 2 (gfx950 correction, applied to every")
print("kernel: an upper bound for kernels whose reads are narrower than 16 bytes per lane) and WRITE_SIZE, rocprofv3 --pmc passes of")
print("tools/train_loop.py %d (+2 warm-up) divided by %d steps; the one-off set-up launches of the first step are included (< 1 %%)." % ($steps, steps))
print("read %.2f GB + written %.2f GB = %.2f GB per step   (algorithmic: SURVEY 8(d) 21.5 GB = 4 B x 3 x conv in+out elements; launch lists, bench.py step_bounds: see the bench line)" % (tot["FETCH_SIZE"] / steps / 1e9, tot["WRITE_SIZE"] / steps / 1e9, (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / steps / 1e9))
print("%-62s %8s %8s %8s" % ("kernel", "calls", "read MB", "write MB"))
for k, v in sorted(per.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:28]:
    print("%-62s %8.1f %8.1f %8.1f" % (k, v[2] / steps, v[0] / steps / 1e6, v[1] / steps / 1e6))
PY
rm -rf gpurun_out/st_FETCH_SIZE gpurun_out/st_WRITE_SIZE
cat gpurun_out/${tag}_step_traffic.txt

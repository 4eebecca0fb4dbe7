"""Sum the TCC FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh per kernel name: HBM traffic of one train step.
usage: step_traffic.py <timed steps> <label>   (reads gpurun_out/st_FETCH_SIZE, gpurun_out/st_WRITE_SIZE)"""
import csv, glob, collections, re, sys
steps = int(sys.argv[1]) + 2          # train_loop.py runs two warm-up steps
label = sys.argv[2] if len(sys.argv) > 2 else ''
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
print("HBM traffic per train step (%s) from the TCC counters: FETCH_SIZE x 2 (gfx950 correction, applied to every" % label)
print("kernel: an upper bound for kernels whose reads are narrower than 16 bytes per lane) and WRITE_SIZE, rocprofv3 --pmc passes of")
print("tools/train_loop.py %d (+2 warm-up) divided by %d steps; the one-off set-up launches of the first step are included (< 1 %%)." % (steps - 2, steps))
print("read %.2f GB + written %.2f GB = %.2f GB per step" % (tot["FETCH_SIZE"] / steps / 1e9, tot["WRITE_SIZE"] / steps / 1e9, (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / steps / 1e9))
print("%-62s %8s %8s %8s" % ("kernel", "calls", "read MB", "write MB"))
for k, v in sorted(per.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:32]:
    print("%-62s %8.1f %8.1f %8.1f" % (k, v[2] / steps, v[0] / steps / 1e6, v[1] / steps / 1e6))

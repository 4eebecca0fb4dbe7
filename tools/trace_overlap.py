"""Where does the wall time of a train step go?  Reads a rocprofv3 *_kernel_trace.csv of tools/train_loop.py
and reports, for the last `steps` steps: wall, busy (union of kernel intervals), idle gaps, time with >= 2
kernels in flight, and per kernel name the EXCLUSIVE time (it alone on the GPU) and the idle time that
follows it.   python tools/trace_overlap.py <kernel_trace.csv> <steps> [marker-kernel-substring]"""
import csv, sys, collections
path, steps = sys.argv[1], int(sys.argv[2])
marker = sys.argv[3] if len(sys.argv) > 3 else 'rmsprop'
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', ''),
                 r.get('Queue_Id', ''), int(r.get('Grid_Size_X', 0) or 0) * int(r.get('Grid_Size_Y', 1) or 1), int(r.get('Workgroup_Size_X', 1) or 1)))
rows.sort()
ends = [i for i, r in enumerate(rows) if marker in r[2]]
assert len(ends) > steps, 'marker kernel %r seen %d times' % (marker, len(ends))
lo, hi = ends[-steps - 1] + 1, ends[-1] + 1
rows = rows[lo:hi]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for i, r in enumerate(rows):
    ev.append((r[0], 1, i)); ev.append((r[1], -1, i))
ev.sort()
active, last = set(), t0
busy = multi = idle = thin = 0
thin_by = collections.Counter()
excl = collections.Counter(); gap_after = collections.Counter(); tot = collections.Counter(); cnt = collections.Counter()
last_ended = None
for t, d, i in ev:
    dt = t - last
    if len(active) == 0:
        idle += dt
        if last_ended is not None: gap_after[rows[last_ended][2]] += dt
    else:
        busy += dt
        if len(active) >= 2: multi += dt
        if sum(rows[j][4] // max(1, rows[j][5]) for j in active) < 256:      # fewer workgroups in flight than CUs
            thin += dt
            for j in active: thin_by[rows[j][2]] += dt
        else: excl[rows[next(iter(active))][2]] += dt
    if d == 1: active.add(i)
    else: active.discard(i); last_ended = i
    last = t
for r in rows: tot[r[2]] += r[1] - r[0]; cnt[r[2]] += 1
ms = lambda x: x / steps / 1e6
print('per step: wall %.3f ms | busy %.3f | idle %.3f | >=2 kernels in flight %.3f | sum of kernel durations %.3f | launches %d' % (
    ms(t1 - t0), ms(busy), ms(idle), ms(multi), ms(sum(tot.values())), len(rows) // steps))
print('under-filled (all kernels in flight together < 256 workgroups): %.3f ms/step; by kernel: %s' % (
    ms(thin), ', '.join('%s %.2f' % (k[:28], ms(v)) for k, v in thin_by.most_common(8))))
small = sum(r[1] - r[0] for r in rows if r[4] // max(1, r[5]) < 256)
print('kernels with < 256 workgroups: %d launches/step, %.3f ms/step' % (sum(1 for r in rows if r[4] // max(1, r[5]) < 256) // steps, ms(small)))
print('%-58s %6s %8s %8s %8s' % ('kernel', 'n/step', 'total', 'alone', 'gap-after'))
for k, v in sorted(tot.items(), key=lambda kv: -excl[kv[0]] - gap_after[kv[0]])[:28]:
    print('%-58s %6.1f %8.3f %8.3f %8.3f' % (k[:58], cnt[k] / steps, ms(v), ms(excl[k]), ms(gap_after[k])))

print()
print('by (kernel, workgroups): n/step, avg us, total ms/step, alone ms/step')
g = collections.defaultdict(lambda: [0, 0]); 
for r in rows:
    k = (r[2][:50], r[4] // max(1, r[5])); g[k][0] += 1; g[k][1] += r[1] - r[0]
for k, v in sorted(g.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%-52s wgs %6d  n %5.1f  avg %7.1f us  total %7.3f ms' % (k[0], k[1], v[0] / steps, v[1] / v[0] / 1e3, ms(v[1])))

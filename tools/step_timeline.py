"""One train step as a flat timeline: python tools/step_timeline.py <kernel_trace.csv> [marker]  ->  one line per kernel of
the LAST complete step: start (us from the step's first kernel), duration (us), idle gap before it on the whole GPU,
queue, workgroups, kernels in flight when it started, name."""
import csv, sys
path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else 'rmsprop'
rows = []
for r in csv.DictReader(open(path)):
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    wg = max(1, int(r.get('Workgroup_Size_X', 1) or 1))
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name, r.get('Queue_Id', ''),
                 int(r.get('Grid_Size_X', 0) or 0) * int(r.get('Grid_Size_Y', 1) or 1) // wg))
rows.sort()
ends = [i for i, r in enumerate(rows) if marker in r[2]]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
t0 = step[0][0]
busy_until = t0
for i, (s, e, name, q, wgs) in enumerate(step):
    inflight = sum(1 for (s2, e2, _, _, _) in step[max(0, i - 12):i] if e2 > s)
    gap = max(0, s - busy_until)
    busy_until = max(busy_until, e)
    print('%9.1f %7.1f gap %6.1f q%-3s wg %6d fl %d  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, q, wgs, inflight, name[:70]))
print('step wall %.1f us' % ((step[-1][1] - t0) / 1e3))

"""Summarise a rocprofv3 kernel_stats.csv per training step: python tools/prof_summary.py <csv> <steps>"""
import csv, sys
path, steps = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(path)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time per step: %.2f ms' % (tot / steps / 1e6))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    name = r['Name'].split('(')[0].replace('void ', '')[:60]
    print('%-62s calls/step %6.1f  ms/step %7.3f  avg_us %8.1f  %5.1f%%' % (
        name, float(r['Calls']) / steps, float(r['TotalDurationNs']) / steps / 1e6, float(r['AverageNs']) / 1e3, float(r['Percentage'])))

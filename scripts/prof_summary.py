"""summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel totals, per step"""
import csv, glob, sys
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time %.2f ms (%.2f ms/step over %g steps)' % (tot / 1e6, tot / 1e6 / steps, steps))
for r in rows[:top]:
    print('%-62s calls %5s  %7.3f ms/step  avg %8.1f us  %5.1f%%' % (
        r['Name'][:62], r['Calls'], float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3,
        100 * float(r['TotalDurationNs']) / tot))

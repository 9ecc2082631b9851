"""kernel_stats.csv of a rocprofv3 --kernel-trace --stats run -> markdown table (profiles/)."""
import csv, glob, sys
src = glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv')[0]
title = sys.argv[2]
rows = list(csv.DictReader(open(src)))
print('# ' + title + '\n')
print('| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|')
for r in rows[:24]:
    print('| `%s` | %s | %.2f | %.2f | %.2f | %s |' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3,
                                               float(r['MaxNs']) / 1e3, r['Percentage'][:6]))

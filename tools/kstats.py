"""Top kernels of a rocprofv3 --kernel-trace --stats run (development helper): python tools/kstats.py <dir> [steps]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print('total ms per step', round(tot / 1e6 / steps, 3))
for r in rows[:16]:
    print(r['Name'][:72], round(int(r['Calls']) / steps, 1), round(float(r['AverageNs']) / 1e3, 1), round(int(r['TotalDurationNs']) / 1e6 / steps, 3))

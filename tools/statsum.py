#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 kernel_stats.csv: statsum.py <csv> [steps] [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(r['Name'][:110].ljust(110), r['Calls'].rjust(6), ("%.1f" % (float(r['AverageNs']) / 1e3)).rjust(9), r['Percentage'])
print("total kernel ms/step", sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps)

#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel (name, grid) the mean of each counter."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        key = (r['Kernel_Name'][:60], r.get('Grid_Size', ''))
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, ctrs in acc.items():
    print(key)
    for c, v in sorted(ctrs.items()):
        print('    %-24s %14.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))

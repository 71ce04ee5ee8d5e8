#!/usr/bin/env python3
"""Median GPU-side duration per (kernel, grid) from a rocprofv3 --kernel-trace CSV."""
import csv
import sys
from collections import defaultdict

d = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    g = r.get('Grid_Size') or (r.get('Grid_Size_X', '') + 'x' + r.get('Grid_Size_Y', '') + 'x' + r.get('Grid_Size_Z', ''))
    d[(r['Kernel_Name'][:80], g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for k, v in d.items():
    if flt and flt not in k[0]:
        continue
    v = sorted(v)
    print('%8.1f us (min %7.1f, n=%3d) grid %-14s %s' % (v[len(v) // 2], v[0], len(v), k[1], k[0]))

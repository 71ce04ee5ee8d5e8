#!/usr/bin/env python3
"""Per-kernel mean of the counters in rocprofv3 --pmc counter_collection CSVs (several passes may be given).
    python tools/pmc_summary.py <filter|filter> dir1/b_counter_collection.csv [dir2/...]"""
import collections
import csv
import sys


def main():
    keys = sys.argv[1].split('|')
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sys.argv[2:]:
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if not any(k in n for k in keys):
                continue
            per[(n[:60], r['Grid_Size'], r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
        for (n, g, _), cs in per.items():
            for c, v in cs.items():
                out[(n, g)][c].append(v)
    for k, cs in sorted(out.items()):
        print(k[0], 'grid', k[1])
        for c, v in sorted(cs.items()):
            print('    %-26s %14.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))


if __name__ == '__main__':
    main()

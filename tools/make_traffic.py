#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes) ->
profiles/<round>_pmc_traffic.json: per (kernel symbol, grid threads) the mean KB fetched / written per
launch and the corrected HBM byte count (gfx950: FETCH_SIZE reports half the bytes of wide coalesced
reads -> doubled; WRITE_SIZE is exact)."""
import csv
import glob
import json
import sys
from collections import defaultdict


def collect(pattern, counter):
    acc = defaultdict(list)
    for path in glob.glob(pattern):
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] != counter:
                continue
            name = r['Kernel_Name'].replace('(anonymous namespace)::', '')
            name = name[:name.index('(')] if '(' in name else name
            acc[(name.replace('void ', '').strip(), int(r['Grid_Size']))].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main(fetch_glob, write_glob, out):
    f, w = collect(fetch_glob, 'FETCH_SIZE'), collect(write_glob, 'WRITE_SIZE')
    res = {}
    for k in sorted(set(f) | set(w)):
        if 'at::' in k[0] or '__amd' in k[0] or 'rocclr' in k[0]:
            continue
        fk, wk = f.get(k, 0.0), w.get(k, 0.0)
        res['%s|%d' % k] = {'fetch_kb': fk, 'write_kb': wk, 'hbm_bytes': int(2 * fk * 1024 + wk * 1024)}
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
    print('wrote', out, len(res), 'kernels')


if __name__ == '__main__':
    main(*sys.argv[1:4])

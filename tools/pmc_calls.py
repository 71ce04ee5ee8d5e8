#!/usr/bin/env python3
"""rocprofv3 --pmc passes over tools/kbench.py -> counters PER C-ABI CALL (an entry point may launch several kernels):
the counter trace is cut at the `af_marker_kernel` dispatches kbench puts in front of every op, segment k belongs to
entry k of kbench's manifest, and the counters of all library kernels in the segment are summed and divided by the calls.

    traffic:  python tools/pmc_calls.py traffic <manifest.json> <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [commit]
              (FETCH_SIZE / WRITE_SIZE from SEPARATE runs, as MI355X_MICROARCH.md prescribes; gfx950: FETCH_SIZE reports half the
               bytes of wide coalesced reads -> doubled; WRITE_SIZE is exact)
    valu:     python tools/pmc_calls.py valu <manifest.json> <sq counter_collection.csv> <out.json> [commit]
              (valu_busy = SQ_ACTIVE_INST_VALU / (8 * SQ_BUSY_CYCLES): the share of the chip's SIMD time spent issuing VALU)
"""
import csv
import json
import sys
from collections import defaultdict


def segments(path):
    """-> list of segments; a segment = {counter: total, '_kernels': {name: dispatches}}; cut at af_marker_kernel."""
    csv.field_size_limit(1 << 30)
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    segs, cur, seen = [], None, set()
    for r in rows:
        name = r['Kernel_Name']
        if 'af_marker_kernel' in name:
            if int(r['Dispatch_Id']) not in seen:  # one row per counter per dispatch
                seen.add(int(r['Dispatch_Id']))
                cur = {'_kernels': defaultdict(int), '_disp': set()}
                segs.append(cur)
            continue
        if cur is None or 'at::' in name or '__amd' in name or 'rocclr' in name or 'Cijk' in name:
            continue
        cur[r['Counter_Name']] = cur.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        if int(r['Dispatch_Id']) not in cur['_disp']:
            cur['_disp'].add(int(r['Dispatch_Id']))
            short = name.replace('(anonymous namespace)::', '').replace('void ', '')
            cur['_kernels'][short[:short.index('(')] if '(' in short else short] += 1
    return segs


def key(m):
    return '%s|%s' % (m['name'], ','.join(str(v) for v in m['shape']))


def main():
    kind, manifest = sys.argv[1], json.load(open(sys.argv[2]))
    if kind == 'traffic':
        fs, ws, out = segments(sys.argv[3]), segments(sys.argv[4]), sys.argv[5]
        commit = sys.argv[6] if len(sys.argv) > 6 else None
        assert len(fs) == len(manifest) == len(ws), (len(fs), len(ws), len(manifest))
        res = {'_meta': {'commit': commit, 'what': 'HBM bytes per C-ABI call: 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes), rocprofv3 --pmc, '
                                                   'separate passes over tools/kbench.py (warm caches: the same buffers every iteration)'}}
        for m, f, w in zip(manifest, fs, ws):
            if m['name'] is None:
                continue
            fk, wk = f.get('FETCH_SIZE', 0.0) / m['calls'], w.get('WRITE_SIZE', 0.0) / m['calls']
            res[key(m)] = {'fetch_kb': fk, 'write_kb': wk, 'hbm_bytes': int(2 * fk * 1024 + wk * 1024),
                           'kernels_per_call': {k: v / m['calls'] for k, v in f['_kernels'].items()}}
    else:
        ss, out = segments(sys.argv[3]), sys.argv[4]
        commit = sys.argv[5] if len(sys.argv) > 5 else None
        assert len(ss) == len(manifest), (len(ss), len(manifest))
        res = {'_meta': {'commit': commit, 'what': 'SQ counters per C-ABI call (rocprofv3 --pmc over tools/kbench.py); valu_busy = '
                                                   'SQ_ACTIVE_INST_VALU / (8 * SQ_BUSY_CYCLES)'}}
        for m, s in zip(manifest, ss):
            if m['name'] is None:
                continue
            c = {k: v / m['calls'] for k, v in s.items() if not k.startswith('_')}
            bc = c.get('SQ_BUSY_CYCLES', 0.0)
            c['valu_busy'] = c.get('SQ_ACTIVE_INST_VALU', 0.0) / (8.0 * bc) if bc else None
            c['valu_lane_instructions'] = 64.0 * c.get('SQ_INSTS_VALU', 0.0)  # upper bound (all lanes active)
            res[key(m)] = c
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
    print('wrote', out, len(res) - 1, 'calls')


if __name__ == '__main__':
    main()

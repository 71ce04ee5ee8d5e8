#!/usr/bin/env python3
"""gpurun_out/parity_margins.json (written by a GPU pytest session, tests/conftest.py) -> per test the worst
err / tol over its assert_close() calls.   python tools/margins_summary.py gpurun_out/parity_margins.json profiles/r03_parity_margins.json [commit]"""
import json
import sys
from collections import defaultdict

rows = json.load(open(sys.argv[1]))
per = defaultdict(list)
for r in rows:
    per[r['test']].append(r)
out = {'_meta': {'commit': sys.argv[3] if len(sys.argv) > 3 else None,
                 'what': 'per test: the assert_close() call with the largest err/tol; err_over_tol = 1 is the tolerance itself',
                 'note': 'tests left with more than 20x headroom compare at an absolute tolerance that is already at or below '
                         'the SURVEY 8(a) figure (1e-6 .. 5e-6 on O(1) values: the measured error there is 1-8 ulp) or are '
                         'bit-identical (max_err 0); dividing those further would only test the rounding of one box'}}
for t, rs in sorted(per.items()):
    w = max(rs, key=lambda r: r['err_over_tol'])
    out[t] = {'worst': w['what'], 'err_over_tol': w['err_over_tol'], 'max_err': w['max_err'], 'atol': w['atol'], 'rtol': w['rtol'],
              'calls': len(rs), 'headroom_x': (1.0 / w['err_over_tol']) if w['err_over_tol'] > 0 else None}
json.dump(out, open(sys.argv[2], 'w'), indent=1)
loose = [(t, v['headroom_x']) for t, v in out.items() if t != '_meta' and v['headroom_x'] and v['headroom_x'] > 20]
print('%d tests, %d with more than 20x headroom' % (len(out) - 1, len(loose)))
for t, h in sorted(loose, key=lambda x: -x[1])[:60]:
    print('  %8.0fx  %s' % (h, t))

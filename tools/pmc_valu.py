#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_* pass over tools/kbench.py -> profiles/<round>_pmc_valu.json: per (kernel symbol, grid
threads) the mean counters per launch and the share of the chip's SIMD time spent issuing VALU
instructions (valu_busy = SQ_ACTIVE_INST_VALU / (8 * SQ_BUSY_CYCLES), see below) -- what bench.py uses
to label a kernel VALU-bound rather than HBM-bound.

    rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU \
        SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_valu -o pmc -- python3 tools/kbench.py --iters 2
    python tools/pmc_valu.py gpurun_out/pmc_valu/pmc_counter_collection.csv profiles/r02_pmc_valu.json
"""
import csv
import json
import sys
from collections import defaultdict


def main(path, out):
    csv.field_size_limit(1 << 30)
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '')
        if 'at::' in name or '__amd' in name or 'rocclr' in name:
            continue
        name = (name[:name.index('(')] if '(' in name else name).replace('void ', '').strip()
        acc['%s|%d' % (name, int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
    res = {}
    for k, c in sorted(acc.items()):
        m = {n: sum(v) / len(v) for n, v in c.items()}
        wc = m.get('SQ_WAVE_CYCLES', 0.0)
        m['wait_share_of_wave_cycles'] = m.get('SQ_WAIT_ANY', 0.0) / wc if wc else None
        # SQ_BUSY_CYCLES is summed over the chip's 32 shader engines and counts cycles; SQ_ACTIVE_INST_VALU is summed
        # over waves and counts quad-cycles; 1024 SIMDs => VALU-busy share of the chip = ACTIVE_INST_VALU / (8 * BUSY)
        bc = m.get('SQ_BUSY_CYCLES', 0.0)
        m['valu_busy'] = m.get('SQ_ACTIVE_INST_VALU', 0.0) / (8.0 * bc) if bc else None
        res[k] = m
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
    print('wrote', out, len(res), 'kernels')
    for k, m in res.items():
        print('%-70s valu-busy %.2f  wave-wait %.2f' % (k[:70], m['valu_busy'] or 0, m['wait_share_of_wave_cycles'] or 0))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])

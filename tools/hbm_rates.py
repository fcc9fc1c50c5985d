#!/usr/bin/env python3
"""profiles/<round>_hbm_rates.md (usage: tools/hbm_rates.py r02): measured HBM GB/s per kernel = PMC bytes per launch
(profiles/<round>_pmc_hbm_traffic*.json, tools/pmc_traffic.sh) / average launch duration of the rocprofv3
trace (profiles/<round>_bench_nusc_*_kernel_stats.csv, tools/prof_bench.sh).  Run after refreshing both."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else 'r02'     # round prefix of the profile files


def table(stats_csv, traffic_json):
    stats = {}
    for r in csv.DictReader(open(os.path.join(ROOT, stats_csv))):
        n = re.sub(r'\(anonymous namespace\)::|void ', '', r['Name'])
        stats[re.sub(r'\(.*', '', n)] = float(r['AverageNs']) / 1e3
    rows = []
    for k, v in json.load(open(os.path.join(ROOT, traffic_json))).items():
        if k in stats and not k.startswith(('at::', '__amd')) and 'pack' not in k:
            us = stats[k]
            rows.append((k, us, v['fetch_bytes_corrected'] / 1e6, v['write_bytes'] / 1e6,
                         v['traffic_bytes'] / 1e6 / us * 1e3))
    rows.sort(key=lambda r: -r[1])
    out = ['| kernel | avg launch µs (rocprofv3 trace) | HBM fetch MB / launch | HBM write MB / launch | '
           'measured HBM GB/s | of 8 TB/s |', '|---|---|---|---|---|---|']
    out += [f'| `{k}` | {us:.1f} | {f:.1f} | {w:.1f} | {g:.0f} | {g / 8000:.1%} |' for k, us, f, w, g in rows]
    return '\n'.join(out)


HEAD = f'''# Measured HBM rates per kernel ({RND})

Average launch duration from the rocprofv3 kernel trace of `python bench.py` (`{RND}_bench_nusc_*_kernel_stats.csv`),
HBM-side bytes per launch from the separate PMC passes (`{RND}_pmc_hbm_traffic*.json`: FETCH_SIZE doubled as the
gfx950 note in MI355X_MICROARCH.md prescribes, WRITE_SIZE as read; eager run over the same 4 frames).  These are
the bytes that reach HBM / Infinity Cache: feature tables of 8-85 MB mostly live in L2 / MALL, so the
convolutions' rates sit far below their ALGORITHMIC rates (DESIGN.md section 3), which count every gathered row;
the coordinate kernels are bound by the latency of random 4-8 byte probes, not by bytes.  tools/hbm_rates.py
regenerates this file.
'''

if __name__ == '__main__':
    md = HEAD + '\n## fp32 (BASELINE configs[1])\n\n' + table(
        f'profiles/{RND}_bench_nusc_f32_kernel_stats.csv', f'profiles/{RND}_pmc_hbm_traffic.json')
    if os.path.exists(os.path.join(ROOT, f'profiles/{RND}_pmc_hbm_traffic_bf16.json')):
        md += '\n\n## bf16 (BASELINE configs[2], B = 1)\n\n' + table(
            f'profiles/{RND}_bench_nusc_bf16_kernel_stats.csv', f'profiles/{RND}_pmc_hbm_traffic_bf16.json')
    md += '\n'
    with open(os.path.join(ROOT, 'profiles', f'{RND}_hbm_rates.md'), 'w') as f:
        f.write(md)
    print(md)

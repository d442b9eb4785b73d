#!/usr/bin/env python3
"""Condenses a run of profiles/run_profile.sh into <out>.json + <out>_kernel_stats.csv (kept under profiles/)."""
import csv
import glob
import json
import os
import sys


def main(src, out):
    summary = {'source_dir': src}
    stats = glob.glob(os.path.join(src, 'trace', '*', '*kernel_stats.csv'))
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        summary['kernel_stats'] = [{'name': r['Name'], 'calls': int(r['Calls']), 'avg_ns': float(r['AverageNs']),
                                    'total_ns': float(r['TotalDurationNs']), 'pct': float(r['Percentage'])}
                                   for r in rows[:8]]
        with open(out + '_kernel_stats.csv', 'w') as f:
            f.write(open(stats[0]).read())
    for key, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
        files = glob.glob(os.path.join(src, sub, '*', '*counter_collection.csv'))
        vals = []
        for fn in files:
            for r in csv.DictReader(open(fn)):
                if 'xrt_trace_kernel' in r['Kernel_Name'] and r['Counter_Name'] == key:
                    vals.append(float(r['Counter_Value']))
        if vals:
            summary[key + '_KB_per_launch'] = sum(vals) / len(vals)
    for name in ('bench_trace.json', 'bench_fetch.json', 'bench_write.json'):
        p = os.path.join(src, name)
        if os.path.exists(p):
            lines = [l for l in open(p).read().strip().splitlines() if l.startswith('{')]
            if lines:
                summary[name[:-5]] = json.loads(lines[-1])
    if 'FETCH_SIZE_KB_per_launch' in summary and 'WRITE_SIZE_KB_per_launch' in summary:
        # MI355X_MICROARCH.md, HBM: values are KB; on gfx950 FETCH_SIZE reports half of a streamed read -> doubled
        # (upper bound for this kernel, whose reads are 4 KiB ring loads and atomics); WRITE_SIZE is exact.
        summary['hbm_traffic_bytes_per_launch'] = (2.0 * summary['FETCH_SIZE_KB_per_launch']
                                                   + summary['WRITE_SIZE_KB_per_launch']) * 1024.0
    json.dump(summary, open(out + '.json', 'w'), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if not k.startswith('bench')}, indent=1)[:1500])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])

#!/usr/bin/env python3
"""Condense the counter CSVs of tests/tools/pmc_variants.sh: per variant the counters of the trace kernel,
VALU instructions per 64 photons and the kernel's duration."""
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for v in sys.argv[2:]:
    tag = os.path.basename(v)[:-3]
    c = {}
    for fn in glob.glob(os.path.join(out, tag, '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(fn)):
            if 'xrt_trace_kernel' in r['Kernel_Name'] or 'xrt_staged_kernel' in r['Kernel_Name']:
                c[r['Counter_Name']] = c.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    dur = []
    for fn in glob.glob(os.path.join(out, tag, '*', '*kernel_trace.csv')):
        for r in csv.DictReader(open(fn)):
            if 'xrt_trace_kernel' in r['Kernel_Name'] or 'xrt_staged_kernel' in r['Kernel_Name']:
                dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
    line = [l for l in open(os.path.join(out, tag + '.log')) if l.startswith('{')]
    photons = json.loads(line[-1])['config']['photons_per_step'] if line else 1e9
    d = {'counters': c, 'kernel_ms': dur, 'photons': photons}
    if 'SQ_INSTS_VALU' in c:
        d['valu_per_64'] = c['SQ_INSTS_VALU'] / (photons / 64)
        d['valu_quadcycles_per_64'] = c.get('SQ_ACTIVE_INST_VALU', 0) / (photons / 64)
        d['salu_per_64'] = c.get('SQ_INSTS_SALU', 0) / (photons / 64)
        d['lds_per_64'] = c.get('SQ_INSTS_LDS', 0) / (photons / 64)
        w = c.get('SQ_WAVE_CYCLES', 0)
        if w:
            d['frac_active_valu'] = c.get('SQ_ACTIVE_INST_VALU', 0) / w
            d['frac_wait_inst_any'] = c.get('SQ_WAIT_INST_ANY', 0) / w
            d['frac_wait_any'] = c.get('SQ_WAIT_ANY', 0) / w
    res[tag] = d
    print(tag, json.dumps({k: (round(x, 3) if isinstance(x, float) else x) for k, x in d.items() if k != 'counters'}))
json.dump(res, open(os.path.join(out, 'summary.json'), 'w'), indent=1)

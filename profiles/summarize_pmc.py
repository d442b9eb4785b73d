"""Per-kernel sums of rocprofv3 --pmc passes: python profiles/summarize_pmc.py <dir with pass subdirs> <out.json> [min_grid]
Every *_counter_collection.csv below <dir> is read; counters are summed per kernel name over the dispatches with at
least min_grid work-items (the warm-up calls of the probes run small grids)."""
import csv, glob, json, os, sys, collections
root, out = sys.argv[1], sys.argv[2]
min_grid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if int(float(r['Grid_Size'])) < min_grid or 'xrt_' not in r['Kernel_Name']:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        calls[k][r['Counter_Name']] += 1
res = []
for k, v in agg.items():
    d = {'kernel': k, 'dispatches_per_counter': max(calls[k].values()), 'counters': dict(v)}
    c = v
    if c.get('SQ_WAVE_CYCLES'):
        d['valu_busy_of_wave_cycles'] = c.get('SQ_ACTIVE_INST_VALU', 0) / c['SQ_WAVE_CYCLES']
        d['wait_inst_of_wave_cycles'] = c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES']
    if c.get('SQ_BUSY_CYCLES') and c.get('SQ_ACTIVE_INST_VALU'):
        # SQ_BUSY_CYCLES counts per shader engine (32 on MI355X), SQ_ACTIVE_INST_VALU per SIMD-cycle-of-4: busy fraction of the
        # vector issue = instructions x 4 cycles / (1024 SIMDs x launch cycles); launch cycles = GRBM_GUI_ACTIVE / 8 XCDs
        if c.get('GRBM_GUI_ACTIVE'):
            d['valu_issue_busy'] = c['SQ_ACTIVE_INST_VALU'] * 4.0 / (1024.0 * c['GRBM_GUI_ACTIVE'] / 8.0)
    res.append(d)
json.dump(res, open(out, 'w'), indent=1)
print(out, len(res), 'kernels')

#!/usr/bin/env python3
"""Condenses a run of profiles/run_profile.sh into <out>.json + the CSVs it was made from (kept under profiles/)."""
import csv
import glob
import json
import os
import shutil
import sys

KERNEL = 'xrt_trace_kernel'


def counters(src, sub):
    """Per counter: the values of every dispatch of the propagation kernel in the pass `sub`."""
    vals = {}
    for fn in glob.glob(os.path.join(src, sub, '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(fn)):
            if KERNEL in r['Kernel_Name']:
                vals.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        shutil.copy(fn, '%s_%s_counter_collection.csv' % (OUT, sub))
    return {k: sum(v) / len(v) for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}


def kernel_ms(src, sub):
    d = []
    for fn in glob.glob(os.path.join(src, sub, '*', '*kernel_trace.csv')):
        for r in csv.DictReader(open(fn)):
            if KERNEL in r['Kernel_Name']:
                d.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
    return sum(d) / len(d) if d else None


def main(src, out):
    global OUT
    OUT = out
    summary = {'source_dir': src}
    stats = glob.glob(os.path.join(src, 'trace', '*', '*kernel_stats.csv'))
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        summary['kernel_stats'] = [{'name': r['Name'], 'calls': int(r['Calls']), 'avg_ns': float(r['AverageNs']),
                                    'total_ns': float(r['TotalDurationNs']), 'pct': float(r['Percentage'])}
                                   for r in rows[:8]]
        shutil.copy(stats[0], out + '_kernel_stats.csv')
    for name in ('bench_trace', 'bench_fetch', 'bench_write', 'bench_sq'):
        p = os.path.join(src, name + '.json')
        if os.path.exists(p):
            lines = [l for l in open(p).read().strip().splitlines() if l.startswith('{')]
            if lines:
                summary[name] = json.loads(lines[-1])
    fetch, _ = counters(src, 'pmc_fetch')
    write, _ = counters(src, 'pmc_write')
    if 'FETCH_SIZE' in fetch and 'WRITE_SIZE' in write:
        # MI355X_MICROARCH.md, HBM: values are KB; on gfx950 FETCH_SIZE reports half of a streamed read -> doubled
        # (an upper bound for this kernel, whose reads are 4 KiB ring loads); WRITE_SIZE is exact.
        summary['FETCH_SIZE_KB_per_launch'] = fetch['FETCH_SIZE']
        summary['WRITE_SIZE_KB_per_launch'] = write['WRITE_SIZE']
        summary['hbm_traffic_bytes_per_launch'] = (2.0 * fetch['FETCH_SIZE'] + write['WRITE_SIZE']) * 1024.0
    sq, n = counters(src, 'pmc_sq')
    if sq:
        summary['sq_counters_per_launch'] = sq
        ms = kernel_ms(src, 'pmc_sq')
        photons = summary.get('bench_sq', {}).get('config', {}).get('photons_per_step')
        d = {'kernel_ms_in_this_pass': ms}
        if photons:
            d['valu_wave_instr_per_64_photons'] = sq.get('SQ_INSTS_VALU', 0) / (photons / 64)
            d['salu_wave_instr_per_64_photons'] = sq.get('SQ_INSTS_SALU', 0) / (photons / 64)
            d['lds_wave_instr_per_64_photons'] = sq.get('SQ_INSTS_LDS', 0) / (photons / 64)
        if ms and 'GRBM_GUI_ACTIVE' in sq:
            # MI355X_MICROARCH.md, DVFS: rocprofv3 reports the sum over the 8 XCDs
            clock = sq['GRBM_GUI_ACTIVE'] / 8.0 / (ms * 1e-3)
            d['clock_GHz_under_load'] = clock / 1e9
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over waves; 1024 SIMDs issue one VALU instruction at a time
            d['valu_issue_busy_fraction'] = sq.get('SQ_ACTIVE_INST_VALU', 0) * 4.0 / (1024.0 * clock * ms * 1e-3)
        w = sq.get('SQ_WAVE_CYCLES')
        if w:
            d['wave_time_fraction_issuing_valu'] = sq.get('SQ_ACTIVE_INST_VALU', 0) / w
            d['wave_time_fraction_waiting_any'] = sq.get('SQ_WAIT_ANY', 0) / w
            d['wave_time_fraction_issue_stalled'] = sq.get('SQ_WAIT_INST_ANY', 0) / w
        summary['sq_derived'] = d
    json.dump(summary, open(out + '.json', 'w'), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if not k.startswith('bench')}, indent=1)[:2500])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])

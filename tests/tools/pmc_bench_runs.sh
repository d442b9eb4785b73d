#!/bin/bash
# SQ counters + clock of one bench launch at several run counts.  usage: tests/tools/pmc_bench_runs.sh out runs...
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
for r in "$@"; do
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/r$r -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --runs $r > $out/r$r.log 2>&1
  python3 - $out/r$r $r <<'PY'
import csv, glob, sys
c = {}; dur = []
for fn in glob.glob(sys.argv[1] + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(fn)):
        if 'xrt_trace_kernel' in r['Kernel_Name']:
            c[r['Counter_Name']] = c.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
for fn in glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(fn)):
        if 'xrt_trace_kernel' in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
ms = dur[0]
clock = c['GRBM_GUI_ACTIVE'] / 8 / (ms * 1e-3)
print('runs', sys.argv[2], 'ms %.2f' % ms, 'clock %.3f GHz' % (clock / 1e9), 'valu_busy %.3f' % (c['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * clock * ms * 1e-3)),
      'valu/64ph %.1f' % (c['SQ_INSTS_VALU'] / (float(sys.argv[2]) * 1e6 / 64)), 'waves', c.get('SQ_WAVES'),
      'wave-slot occupancy %.3f' % (c['SQ_WAVE_CYCLES'] * 4 / (4096 * clock * ms * 1e-3)))
PY
done

#!/bin/bash
# kernel durations of a python command: tests/tools/ktrace.sh out_dir script.py [args]   (top kernels by total time)
out=$1; shift
mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 "$@" > $out/run.log 2>&1 || echo "failed"
python3 - $out <<'PY'
import csv, glob, sys
for fn in glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'):
    for r in list(csv.DictReader(open(fn)))[:6]:
        print('%-60s calls %4s  avg %9.3f ms  total %9.3f ms' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e6, float(r['TotalDurationNs']) / 1e6))
PY

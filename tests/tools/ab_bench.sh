#!/bin/bash
# A/B of device-library variants on ONE box (clocks differ between boxes): bench.py per variant, interleaved twice.
# usage: tests/tools/ab_bench.sh out_file variant1.so variant2.so ...
out=$1; shift
mkdir -p $(dirname $out)
for rep in 1 2; do
  for v in "$@"; do
    echo "== $v (rep $rep)" >> $out
    XICSRT_HIP_LIB=$PWD/$v timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        print(json.dumps({'value': round(d['value']), 'ms_per_step': round(d['ms_per_step'], 3), 'kernel_ms': round(d['roofline']['kernel_ms_avg'], 3), 'num_out': d['config']['num_out']}))
    else:
        print(l)
" >> $out
  done
done
cat $out

#!/bin/bash
# SQ counters of the propagation kernel for several device-library variants (one bench launch each).
# usage: tests/tools/pmc_variants.sh out_dir variant.so ...   (run from the repo root on the GPU box)
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
for v in "$@"; do
  tag=$(basename $v .so)
  XICSRT_HIP_LIB=$PWD/$v timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/$tag -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/$tag.log 2>&1 || { echo "FAILED $tag"; tail -3 $out/$tag.log; }
done
python3 tests/tools/pmc_summary.py $out "$@"

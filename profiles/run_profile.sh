#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run from the repo
# root through gpurun).  Usage: profiles/run_profile.sh <tag>
# 1) kernel trace + stats of the default bench command,
# 2) FETCH_SIZE and WRITE_SIZE in separate counter passes (TCC slots, MI355X guide).
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/bench_write.err
python3 profiles/summarize.py $OUT gpurun_out/$TAG

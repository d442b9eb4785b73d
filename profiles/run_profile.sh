#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run from the repo root through gpurun).
# Usage: profiles/run_profile.sh <tag>      ->  gpurun_out/<tag>.json, gpurun_out/<tag>_*.csv  (copy into profiles/)
# Passes (counters in passes of their own, as /opt/skills/guides/MI355X_MICROARCH.md prescribes):
#  1) kernel trace + stats of the default bench command,
#  2) FETCH_SIZE, 3) WRITE_SIZE (TCC slots: not both in one pass),
#  4) SQ issue / wait counters + GRBM_GUI_ACTIVE (clock under load) of one bench launch.
set -e
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/bench_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_sq.json 2> $OUT/bench_sq.err
python3 profiles/summarize.py $OUT gpurun_out/$TAG

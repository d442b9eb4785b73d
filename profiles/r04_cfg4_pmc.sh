#!/bin/bash
# counters of the cfg4-shaped plasma scene (4096 runs, 6.2e8 rays) launches: tests/bench_cfg5.py under rocprofv3, one counter set per pass
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/cfg4_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tests/bench_plasma.py 4096 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/tests/bench_plasma.py 4096 > $OUT/sq.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/tc -- python3 $GRAFT_REPO_ROOT/tests/bench_plasma.py 4096 > $OUT/tc.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/tests/bench_plasma.py 4096 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/tests/bench_plasma.py 4096 > $OUT/write.log 2>&1
cd $GRAFT_REPO_ROOT
python3 profiles/summarize_pmc.py $OUT gpurun_out/r04_cfg4_plasma_counters.json 100000
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r04_cfg4_plasma_kernel_stats.csv

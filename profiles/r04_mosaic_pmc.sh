#!/bin/bash
# counters of the mosaic-crystal probe (first phase of the fused kernel + xrt_mosaic_kernel, 1000 runs x 1e6 rays): tests/bench_mosaic.py under rocprofv3
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/mosaic_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tests/bench_mosaic.py 1000 1000000 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/tests/bench_mosaic.py 1000 1000000 > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/tc -- python3 $GRAFT_REPO_ROOT/tests/bench_mosaic.py 1000 1000000 > $OUT/tc.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/tests/bench_mosaic.py 1000 1000000 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/tests/bench_mosaic.py 1000 1000000 > $OUT/write.log 2>&1
cd $GRAFT_REPO_ROOT
python3 profiles/summarize_pmc.py $OUT gpurun_out/r04_mosaic_counters.json 100000
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r04_mosaic_kernel_stats.csv

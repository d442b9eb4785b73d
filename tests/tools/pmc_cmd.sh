#!/bin/bash
# SQ counters + kernel trace + HBM traffic counters of an arbitrary python command (separate passes, as the
# MI355X guide prescribes).  usage: tests/tools/pmc_cmd.sh out_dir tag script.py [args...]
out=$1; tag=$2; shift; shift
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- python3 "$@" > $out/${tag}_trace.log 2>&1 || echo "trace pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/${tag}_sq -- python3 "$@" > $out/${tag}_sq.log 2>&1 || echo "sq pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/${tag}_sq2 -- python3 "$@" > $out/${tag}_sq2.log 2>&1 || echo "sq2 pass failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_fetch -- python3 "$@" > $out/${tag}_fetch.log 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_write -- python3 "$@" > $out/${tag}_write.log 2>&1 || echo "write pass failed"
python3 tests/tools/pmc_cmd_summary.py $out $tag

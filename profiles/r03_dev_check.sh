python3 -c "import torch" >/dev/null 2>&1
export TMPDIR=/tmp
export XICSRT_HIP_LIB=$PWD/xicsrt_amd/csrc/dev_a.so
mkdir -p gpurun_out/r3b
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3b/t -- python3 tests/bench_plasma.py 4096 > gpurun_out/r3b/log 2>&1
grep "^{" gpurun_out/r3b/log | cut -c1-330
f=$(find gpurun_out/r3b/t -name '*kernel_trace.csv' | head -1); python3 profiles/timeline.py $f 40 | grep "scout\|trace_kernel"
rm -rf gpurun_out/r3b/t

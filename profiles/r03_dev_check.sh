python3 -c "import torch" >/dev/null 2>&1
export TMPDIR=/tmp
mkdir -p gpurun_out/r3b
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3b/t -- python3 tests/bench_mosaic.py > gpurun_out/r3b/log 2>&1
f=$(find gpurun_out/r3b/t -name '*kernel_trace.csv' | head -1); python3 profiles/timeline.py $f 30 | grep -v "copyBuffer\|fillBuffer\|put_kernel"
rm -rf gpurun_out/r3b/t

#!/bin/bash
# Throughput / parity probes beside bench.py (run from the repo root through gpurun):
#   profiles/run_probes.sh <tag>   ->  gpurun_out/<tag>_probes.txt  (copy into profiles/)
# One JSON line per probe; every line that compares says whether the device equalled the oracle / the other route.
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_probes.txt
mkdir -p gpurun_out
python3 -c "import torch" > /dev/null 2>&1          # first import on a fresh box takes a minute
: > $OUT
probe() {   # probe "ENV=1 ..." script args
    echo "# $1 python $2" >> $OUT
    env $1 timeout -k 10 400 python3 $2 2>/dev/null | grep '^{' >> $OUT || echo "# (failed)" >> $OUT
}
probe "" "tests/bench_fewruns.py"                       # few runs: segmented vs one work unit per run
probe "" "tests/bench_variants.py"                      # fused kernel variants at the bench size
probe "" "tests/bench_images.py"                        # cost of the pixel atomics
probe "" "tests/bench_plasma.py 4096"                   # BASELINE cfg4 shape: scout + fused kernel
probe "XICSRT_PLASMA_STAGED=1" "tests/bench_plasma.py 4096"     # ... through the staged kernels
probe "" "tests/bench_staged.py"                        # np.random.normal wavelengths, prepared
probe "XICSRT_STAGED_GAUSS=1" "tests/bench_staged.py"   # ... through the staged kernels
probe "" "tests/bench_cfg5.py 1000 1000000 2"           # BASELINE cfg5 at full size, flat and interpolated mesh
probe "" "tests/bench_mosaic.py 1000 1000000"           # mosaic crystal, 15 layers (parked rays + xrt_mosaic_kernel), at the bench size
probe "" "tests/bench_mosaic.py 256 1000000"            # ... and at 256 runs (one workgroup per CU: the latency of one run's chain of steps)
probe "" "tests/bench_history.py"                       # raytrace(config) with keep_history, as notebooks call it

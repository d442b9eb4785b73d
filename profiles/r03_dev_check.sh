python3 -c "import torch" >/dev/null 2>&1
XICSRT_HIP_LIB=$PWD/xicsrt_amd/csrc/dev_a.so python3 tests/bench_cfg5.py 1000 1000000 1 2>&1 | grep "^{" | cut -c1-330

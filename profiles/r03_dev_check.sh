python3 -c "import torch" >/dev/null 2>&1
export XICSRT_HIP_LIB=$PWD/xicsrt_amd/csrc/dev_a.so
for args in "1000000 100 crystal" "1000000 125 crystal" "10000000 1 crystal" "1000000 10 crystal" "1000000 1 crystal" "100000 1 crystal" "1000000 50 crystal" "3000000 30 crystal"; do
  python3 tests/bench_plan.py $args 2>&1 | grep -v "amdgpu.ids\|Warning\|print\|ret = \|^    \|units;"
done

python3 -c "import torch" >/dev/null 2>&1
export UNIT_CLOCKS=1
for v in a; do
  for args in "1000000 100 crystal XICSRT_SUBUNITS=1" "1000000 100 crystal XICSRT_SUBUNITS=2" "1000000 100 crystal XICSRT_SUBUNITS=2 XICSRT_TARGET_UNITS=2000" "10000000 1 crystal" "1000000 125 crystal"; do
  XICSRT_HIP_LIB=$PWD/xicsrt_amd/csrc/dev_$v.so python3 tests/bench_plan.py $args 2>&1 | grep -v "amdgpu.ids\|Warning\|print\|ret = \|^    \|units;"
  done
done

python3 -c "import torch" >/dev/null 2>&1
export UNIT_CLOCKS=1
for args in "1000000 100 crystal" "1000000 100 crystal XICSRT_SUBUNITS=1" "1000000 100 crystal XICSRT_SUBUNITS=1 XICSRT_CHUNK_HEADS=12" \
  "1000000 100 mirror" "1000000 100 mirror XICSRT_SUBUNITS=1" "10000000 1 crystal" "1000000 125 crystal" "1000000 10 crystal" "1000000 1 crystal"; do
  python3 tests/bench_plan.py $args 2>&1 | grep -v "amdgpu.ids\|Warning\|print\|ret = "
done

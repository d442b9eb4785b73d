export TMPDIR=/tmp
mkdir -p gpurun_out/r3a
python3 -c "import torch" >/dev/null 2>&1
for c in "1000000 100 crystal" "1000000 100 mirror" "10000000 1 crystal" "1000000 125 crystal"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3a/$tag -- python3 tests/prof_single.py $c > gpurun_out/r3a/$tag.log 2>&1
  f=$(find gpurun_out/r3a/$tag -name '*kernel_stats.csv' | head -1)
  cp $f gpurun_out/r3a/${tag}_kernel_stats.csv
  rm -rf gpurun_out/r3a/$tag
done

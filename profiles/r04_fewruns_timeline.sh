export TMPDIR=/tmp
mkdir -p gpurun_out/r04n
python3 -c "import torch" >/dev/null 2>&1
for c in "1000000 125 crystal" "1000000 100 mirror" "10000000 1 crystal"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04n/$tag -- python3 tests/prof_single.py $c > gpurun_out/r04n/$tag.log 2>&1
  f=$(find gpurun_out/r04n/$tag -name '*kernel_trace.csv' | head -1)
  echo "== $c"; tail -n 2 gpurun_out/r04n/$tag.log
  python3 profiles/timeline.py $f 16
  rm -rf gpurun_out/r04n/$tag
done

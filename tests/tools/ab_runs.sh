for r in 300 400 512; do
  for sb in 256 513; do
    XICSRT_SEG_BELOW=$sb python3 bench.py --runs $r --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('runs $r seg_below $sb', round(d['value']), 'Mphot/s', round(d['ms_per_step'],3), 'ms')"
  done
done

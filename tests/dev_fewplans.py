"""Development: the few-run scenes under forced plans (XICSRT_SUBUNITS / XICSRT_SEGMENTS / XICSRT_CHUNK_HEADS from the command line). Not a test.
python tests/dev_fewplans.py rays runs crystal|mirror [ENV=VALUE ...]"""
import sys, os, time
for kv in sys.argv[4:]:
    k, v = kv.split('='); os.environ[k] = v
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from xicsrt_amd import xicsrt_raytrace as xrt, config as xconfig
rays, runs, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
config = bench.spectrometer_config(rays, runs, seed=3)
if kind == 'mirror':
    config['optics']['crystal'] = {'class_name': 'XicsrtOpticPlanarMirror', 'check_size': True, 'origin': [0.0, 0.0, 0.80374151],
                                   'zaxis': [0.0, 0.59497864, -0.80374151], 'xsize': 0.2, 'ysize': 0.2}
config = xconfig.get_config(config)
flat = xrt.Elements(config).flatten()
seeds = xrt.run_seeds(3, runs)
dev = xrt.DeviceTrace(flat)
dev.trace(seeds, 1); dev.results()
best = 1e9
for _ in range(8):
    dev.num_out.zero_(); dev.images.zero_(); torch.cuda.synchronize()
    t0 = time.time(); dev.trace(seeds, 1); torch.cuda.synchronize(); best = min(best, time.time() - t0)
meta, _ = dev.results()
print(' '.join(sys.argv[1:]), '-> %.3f ms' % (best * 1e3), [int(meta[n]['num_out']) for n in flat.names])

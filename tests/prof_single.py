"""A few-run scene under the profiler: python3 tests/prof_single.py [rays] [runs] [crystal|mirror] (not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xicsrt_amd import xicsrt_raytrace as xrt, config as xconfig
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kind = sys.argv[3] if len(sys.argv) > 3 else 'crystal'
config = bench.spectrometer_config(rays, runs, seed=3)
if kind == 'mirror':     # BASELINE cfg2
    config['optics']['crystal'] = {'class_name': 'XicsrtOpticPlanarMirror', 'check_size': True,
                                   'origin': [0.0, 0.0, 0.80374151], 'zaxis': [0.0, 0.59497864, -0.80374151],
                                   'xsize': 0.2, 'ysize': 0.2}
config = xconfig.get_config(config)
flat = xrt.Elements(config).flatten()
seeds = xrt.run_seeds(3, runs)
dev = xrt.DeviceTrace(flat)
dev.trace(seeds, 1); dev.results()
for _ in range(5):
    dev.num_out.zero_(); dev.images.zero_()
    t0 = time.time(); dev.trace(seeds, 1); t1 = time.time(); dev.results(); t2 = time.time()
    print('call %.3f ms, until results %.3f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))

"""Throughput probe: bench geometry with a spherical mosaic (HOPG-like) crystal -> staged path.  Not a test."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, helpers, bench
from xicsrt_amd import xicsrt_raytrace as xrt, config as xconfig
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
config = bench.spectrometer_config(rays, runs, seed=5)
config['optics']['crystal'].update(class_name='XicsrtOpticSphericalMosaicCrystal', mosaic_spread=float(np.radians(0.4)),
                                   mosaic_depth=int(os.environ.get('MOSAIC_DEPTH', '15')), rocking_fwhm=2e-3)
config = xconfig.get_config(config)
flat = xrt.Elements(config).flatten()
seeds = xrt.run_seeds(5, runs)
dev = xrt.DeviceTrace(flat)
dev.trace(seeds, 1); dev.results()
dev.num_out.zero_(); dev.images.zero_()
t0 = time.time(); dev.trace(seeds, 1); meta, image = dev.results(); dt = time.time() - t0
n_or = min(runs, 4)
o_num, o_img = helpers.oracle_counts(flat, seeds[:n_or], 1, threads=4)
dev2 = xrt.DeviceTrace(flat); dev2.trace(seeds[:n_or], 1); m2, i2 = dev2.results()
same = all(int(m2[nm]['num_out']) == int(o_num[k]) for k, nm in enumerate(flat.names))
print(json.dumps({'runs': runs, 'rays_per_run': rays, 'gpu_s': dt, 'gpu_Mphot_s': runs * rays / dt / 1e6,
                  'num_out': {nm: int(meta[nm]['num_out']) for nm in flat.names}, 'gpu_equals_oracle': bool(same)}))

"""Development: per-step clocks of the mosaic layers (library built with -DXRT_ST_CLOCKS=1). Not a test."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from xicsrt_amd import xicsrt_raytrace as xrt, config as xconfig
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
config = bench.spectrometer_config(rays, runs, seed=5)
config['optics']['crystal'].update(class_name='XicsrtOpticSphericalMosaicCrystal', mosaic_spread=float(np.radians(0.4)), mosaic_depth=15, rocking_fwhm=2e-3)
config = xconfig.get_config(config)
flat = xrt.Elements(config).flatten()
seeds = xrt.run_seeds(5, runs)
dev = xrt.DeviceTrace(flat)
images = os.environ.get('NO_IMAGES') is None
dev.trace(seeds, 1, keep_images=images); dev.results()
dev._ws[128:224].zero_()
t0 = time.time(); dev.trace(seeds, 1, keep_images=images); dev.results(); dt = time.time() - t0
w = dev._ws[128:224].cpu().numpy().view(np.uint64)
print(json.dumps({'runs': runs, 'call_ms': dt * 1e3, 'gauss_ms_per_wg': float(w[0]) / 1e5 / runs, 'uniform_ms_per_wg': float(w[1]) / 1e5 / runs,
                  'pass_ms_per_wg': float(w[2]) / 1e5 / runs, 'ray_layers_per_run': int(w[3]) // runs,
                  'list_ms_per_wg': float(w[4]) / 1e5 / runs, 'behind_ms_per_wg': float(w[5]) / 1e5 / runs,
                  'generator_sleeps_per_run': int(w[6]) // runs, 'generator_steps_per_run': int(w[7]) // runs,
                  'tester_word_waits_per_run': int(w[8]) // runs, 'tester_chain_waits_per_run': int(w[9]) // runs,
                  'behind_load_ms_wave0': float(w[10]) / 1e5 / runs, 'behind_work_ms_wave0': float(w[11]) / 1e5 / runs}))

"""Throughput probe for the staged path on the BASELINE cfg4-shaped scene (plasma cube ->
spherical crystal -> 800x400 detector).  Not a test; run on the GPU box:  python tests/bench_plasma.py [runs]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, helpers
from xicsrt_amd import xicsrt_raytrace as xrt
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
if len(sys.argv) > 2 and sys.argv[2] == 'voigt':     # a temperature profile with a natural line width: one Voigt table per bundle
    cfg, _ = helpers.load_golden('F_toroidal_voigt_counts')
    cfg['sources']['source'].update(bundle_count=1000)
else:
    cfg, _ = helpers.load_golden('F_plasma_counts')
    cfg['sources']['source'].update(emissivity=2e15, bundle_count=2000, bundle_volume=0.001 / 2000)
    cfg['optics']['detector']['pixel_size'] = 5e-4
cfg['general'].update(number_of_runs=runs, random_seed=0)
config, elements, flat = helpers.build(cfg)
seeds = xrt.run_seeds(0, runs)
dev = xrt.DeviceTrace(flat)
dev.trace(seeds, 1); dev.results()              # warm-up with the full run count: allocates the workspace
dev.num_out.zero_(); dev.images.zero_()
t0 = time.time(); dev.trace(seeds, 1); meta, image = dev.results(); dt = time.time() - t0
n = int(meta['source']['num_out'])
t1 = time.time(); o_num, o_img = helpers.oracle_counts(flat, seeds[:16], 1, threads=16); dto = time.time() - t1
o16 = helpers.oracle_counts(flat, seeds[:16], 1, threads=16)[0] if False else o_num
dev2 = xrt.DeviceTrace(flat); dev2.trace(seeds[:16], 1); m2, i2 = dev2.results()
same = all(int(m2[nm]['num_out']) == int(o_num[k]) for k, nm in enumerate(flat.names)) and np.array_equal(
    np.concatenate([i2[nm].ravel() for nm in flat.names[1:] if i2[nm] is not None]).astype(np.int64), o_img[:flat.image_bins])
print(json.dumps({'scene': 'voigt per bundle' if (len(sys.argv) > 2 and sys.argv[2] == 'voigt') else 'cfg4', 'runs': runs, 'rays': n, 'capacity_per_run': flat.n_rays, 'gpu_s': dt, 'gpu_Mphot_s': n / dt / 1e6,
                  'oracle_16runs_s': dto, 'oracle_Mphot_s_16thr': int(o_num[0]) / dto / 1e6,
                  'num_out': {nm: int(meta[nm]['num_out']) for nm in flat.names}, 'gpu_equals_oracle_on_16_runs': bool(same)}))

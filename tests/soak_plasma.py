"""Parity soak for the plasma route (not a test): the BASELINE cfg4-shaped scene (plasma cube, 2000 bundles per run ->
spherical crystal -> 800 x 400 detector), device (scout kernel + fused kernel) vs CPU oracle on every counter and pixel;
then the same with a temperature profile and a natural line width (one Voigt table per bundle).
python tests/soak_plasma.py [runs]"""
import sys, os, time, json, copy
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import logging
logging.disable(logging.WARNING)
import numpy as np, helpers
from xicsrt_amd import xicsrt_raytrace as xrt
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
base, _ = helpers.load_golden('F_plasma_counts')
base['sources']['source'].update(emissivity=2e15, bundle_count=2000, bundle_volume=0.001 / 2000)
base['optics']['detector']['pixel_size'] = 5e-4
voigt, _ = helpers.load_golden('F_toroidal_voigt_counts')
voigt['sources']['source'].update(bundle_count=1000)
for tag, cfg, r in (('cfg4 plasma cube', base, runs), ('toroidal plasma, Voigt table per bundle', voigt, max(runs // 4, 1))):
    cfg = copy.deepcopy(cfg)
    cfg['general'].update(number_of_runs=r, number_of_iter=1, random_seed=0, keep_history=False)
    config, elements, flat = helpers.build(cfg)
    seeds = xrt.run_seeds(0, r)
    dev = xrt.DeviceTrace(flat)
    t0 = time.time(); dev.trace(seeds, 1); meta, image = dev.results(); t_gpu = time.time() - t0
    threads = min(os.cpu_count() or 1, 64)
    t0 = time.time(); o_num, o_img = helpers.oracle_counts(flat, seeds, 1, threads=threads); t_cpu = time.time() - t0
    g_num = [int(meta[n]['num_out']) for n in flat.names]
    g_img = dev.images.cpu().numpy()[:flat.image_bins]
    print(json.dumps({'scene': tag, 'runs': r, 'num_out_gpu': g_num, 'num_out_oracle': [int(v) for v in o_num],
                      'counts_equal': g_num == [int(v) for v in o_num], 'pixels_equal': bool(np.array_equal(g_img, o_img[:flat.image_bins])),
                      'pixels': int(flat.image_bins), 'gpu_s': t_gpu, 'oracle_s': t_cpu, 'oracle_threads': threads}), flush=True)

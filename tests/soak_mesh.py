"""Parity soak for the mesh path (not a test): BASELINE cfg5 scene (41 x 41 toroidal mesh crystal), flat and
interpolated, device vs CPU oracle on every counter and pixel.  python tests/soak_mesh.py [runs] [rays] [bragg]"""
import sys, os, time, json, copy
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import logging
logging.disable(logging.WARNING)
import numpy as np, helpers
from xicsrt_amd import xicsrt_raytrace as xrt
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
bragg = len(sys.argv) > 3 and sys.argv[3] == 'bragg'          # Bragg test on, with a wide rocking curve (many reflections)
for tag in ('flat', 'interp'):
    cfg, _ = helpers.load_golden('E_cfg5_mesh_%s_1e5' % tag)
    cfg = copy.deepcopy(cfg)
    cfg['general'].update(number_of_runs=runs, number_of_iter=1, keep_history=False)
    cfg['sources']['source']['intensity'] = rays
    cfg['optics']['crystal']['check_bragg'] = bragg          # off: every ray that lands on the mesh goes on to the detector
    if bragg:
        cfg['optics']['crystal'].update(rocking_type='gaussian', rocking_fwhm=2e-2)
    config, elements, flat = helpers.build(cfg)
    seeds = xrt.run_seeds(config['general']['random_seed'], runs)
    dev = xrt.DeviceTrace(flat)
    t0 = time.time(); dev.trace(seeds, 1); meta, image = dev.results(); t_gpu = time.time() - t0
    threads = min(os.cpu_count() or 1, 64)
    t0 = time.time(); o_num, o_img = helpers.oracle_counts(flat, seeds, 1, threads=threads); t_cpu = time.time() - t0
    g_num = [int(meta[n]['num_out']) for n in flat.names]
    g_img = dev.images.cpu().numpy()[:flat.image_bins]
    print(json.dumps({'scene': 'cfg5 mesh ' + tag + (', Bragg test with a 2e-2 rad rocking curve' if bragg else ', check_bragg off'), 'photons': runs * rays, 'num_out_gpu': g_num,
                      'num_out_oracle': [int(v) for v in o_num], 'counts_equal': g_num == [int(v) for v in o_num],
                      'pixels_equal': bool(np.array_equal(g_img, o_img[:flat.image_bins])), 'pixels': int(flat.image_bins),
                      'gpu_s': t_gpu, 'oracle_s': t_cpu, 'oracle_threads': threads}), flush=True)

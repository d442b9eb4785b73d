"""Parity soak for mosaic crystals (not a test): the bench geometry with a 15-layer spherical mosaic crystal (HOPG-like),
device -- the fused kernel's first phase + xrt_mosaic_kernel, and the staged kernels (XICSRT_NO_MOSAIC_FUSED) -- vs CPU oracle
on every counter and pixel; then the same with a cut-off and a uniform wavelength band.
python tests/soak_mosaic.py [runs] [rays]  ->  one JSON line per scene and route"""
import sys, os, time, json, copy
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import logging
logging.disable(logging.WARNING)
import numpy as np, helpers, bench
from xicsrt_amd import xicsrt_raytrace as xrt, config as xconfig, capi
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
base = bench.spectrometer_config(rays, runs, seed=7)
base['optics']['crystal'].update(class_name='XicsrtOpticSphericalMosaicCrystal', mosaic_spread=float(np.radians(0.4)), mosaic_depth=15,
                                 rocking_fwhm=2e-3)
scenes = {'15 layers': base}
c = copy.deepcopy(base)
c['optics']['crystal'].update(mosaic_cutoff=1e-3, mosaic_depth=6)
c['sources']['source'].update(wavelength_dist='uniform', wavelength_range=[3.9485, 3.9499])
scenes['6 layers, cut-off, wavelength band'] = c
threads = min(os.cpu_count() or 1, 64)
for name, cfg in scenes.items():
    config = xconfig.get_config(copy.deepcopy(cfg))
    flat = xrt.Elements(config).flatten()
    seeds = xrt.run_seeds(7, runs)
    t0 = time.time(); o_num, o_img = helpers.oracle_counts(flat, seeds, 1, threads=threads); t_cpu = time.time() - t0
    for route in ('parked rays', 'staged'):
        os.environ.pop('XICSRT_NO_MOSAIC_FUSED', None)
        if route == 'staged':
            os.environ['XICSRT_NO_MOSAIC_FUSED'] = '1'
        dev = xrt.DeviceTrace(flat)
        capi.lib().xrt_last_path(1)
        t0 = time.time(); dev.trace(seeds, 1); meta, image = dev.results(); t_gpu = time.time() - t0
        path = int(capi.lib().xrt_last_path(1))
        g_img = np.concatenate([image[nm].ravel() for nm in flat.names[1:]]).astype(np.int64)
        print(json.dumps({'scene': name, 'route': route, 'path_bits': path, 'photons': runs * rays,
                          'num_out_gpu': [int(meta[n]['num_out']) for n in flat.names], 'num_out_oracle': [int(v) for v in o_num],
                          'counts_equal': [int(meta[n]['num_out']) for n in flat.names] == [int(v) for v in o_num],
                          'pixels_equal': bool(np.array_equal(g_img, o_img[:flat.image_bins])), 'pixels': int(flat.image_bins),
                          'gpu_s': t_gpu, 'oracle_s': t_cpu, 'oracle_threads': threads}), flush=True)

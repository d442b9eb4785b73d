"""Throughput probe for few-run scenes (the reference's usual notebook usage: a handful of runs of
1e5..1e7 rays), with the library's own segmentation and with it switched off (XICSRT_SEGMENTS=1).
Not a test; run on the GPU box:  python tests/bench_fewruns.py"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from xicsrt_amd import xicsrt_raytrace as xrt, config as xconfig


def scene(kind, rays, runs):
    config = bench.spectrometer_config(rays, runs, seed=3)
    if kind == 'mirror':     # BASELINE cfg2
        config['optics']['crystal'] = {'class_name': 'XicsrtOpticPlanarMirror', 'check_size': True,
                                       'origin': [0.0, 0.0, 0.80374151], 'zaxis': [0.0, 0.59497864, -0.80374151],
                                       'xsize': 0.2, 'ysize': 0.2}
    return xconfig.get_config(config)


for kind, rays, runs in (('crystal', 100000, 1), ('crystal', 1000000, 1), ('crystal', 10000000, 1),
                         ('crystal', 1000000, 10), ('mirror', 1000000, 100), ('crystal', 1000000, 100)):
    out = {'scene': kind, 'rays_per_run': rays, 'runs': runs}
    ref = None
    for label, env in (('segmented', None), ('one_unit_per_run', '1')):
        if env is None:
            os.environ.pop('XICSRT_SEGMENTS', None)
        else:
            os.environ['XICSRT_SEGMENTS'] = env
        config = scene(kind, rays, runs)
        flat = xrt.Elements(config).flatten()
        seeds = xrt.run_seeds(3, runs)
        dev = xrt.DeviceTrace(flat)
        dev.trace(seeds, 1); dev.results()          # warm-up: jump polynomials, kernels
        best = 1e9
        for _ in range(3):
            dev.num_out.zero_(); dev.images.zero_()
            t0 = time.time(); dev.trace(seeds, 1); meta, image = dev.results(); best = min(best, time.time() - t0)
        counts = [int(meta[n]['num_out']) for n in flat.names]
        if ref is None:
            ref = (counts, {k: v.copy() for k, v in image.items() if v is not None})
        else:
            out['identical_results'] = bool(counts == ref[0] and all(np.array_equal(image[k], v) for k, v in ref[1].items()))
        out[label + '_ms'] = best * 1e3
        out[label + '_Mphot_s'] = rays * runs / best / 1e6
    print(json.dumps(out), flush=True)

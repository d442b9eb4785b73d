"""Throughput probe for BASELINE cfg5 at its stated size (point source -> 41x41 XicsrtOpticMeshToroidalCrystal ->
detector, 1000 runs x 1e6 rays), mesh_interpolate off and on.  Not a test; run on the GPU box:
python tests/bench_cfg5.py [runs] [rays] [repeat]"""
import sys, os, time, json, copy, ctypes
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch, helpers
from xicsrt_amd import xicsrt_raytrace as xrt, capi
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lib = capi.lib()
for tag in os.environ.get('CFG5_TAGS', 'flat,interp,norefine,81_coarse17').split(','):
    cfg, _ = helpers.load_golden('E_cfg5_mesh_%s_1e5' % tag if tag in ('flat', 'interp') else 'E_mesh_%s_counts' % tag)
    cfg = copy.deepcopy(cfg)
    cfg['general'].update(number_of_runs=runs, number_of_iter=1, keep_history=False)
    cfg['sources']['source']['intensity'] = rays
    config, elements, flat = helpers.build(cfg)
    seeds = xrt.run_seeds(config['general']['random_seed'], runs)
    dev = xrt.DeviceTrace(flat)
    dev.trace(seeds[:8], 1); dev.results()
    best = None
    for r in range(rep):
        dev.num_out.zero_(); dev.images.zero_()
        torch.cuda.synchronize()
        lib.xrt_timing_begin()
        t0 = time.time(); dev.trace(seeds, 1); torch.cuda.synchronize(); dt = time.time() - t0
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        lib.xrt_timing_end(ctypes.byref(ms), ctypes.byref(n))
        if best is None or dt < best[0]:
            best = (dt, ms.value, n.value)
    meta, image = dev.results()
    print(json.dumps({'scene': 'cfg5 41x41 toroidal mesh crystal', 'mesh': tag, 'runs': runs,
                      'rays_per_run': rays, 'call_s': best[0], 'kernel_ms': best[1], 'launches': best[2],
                      'Mphot_s': runs * rays / best[0] / 1e6,
                      'num_out': {nm: int(meta[nm]['num_out']) for nm in flat.names}}), flush=True)

"""Throughput + parity probe for mesh optics on the BASELINE cfg5 scene (point source -> 41x41
XicsrtOpticMeshToroidalCrystal with a 5x5 coarse mesh -> detector), mesh_interpolate False then True.
Not a test; run on the GPU box:  python tests/bench_mesh.py [runs] [rays]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, helpers
from xicsrt_amd import xicsrt_raytrace as xrt
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
for interp in (False, True):
    cfg, _ = helpers.load_golden('E_mesh_interp_trace')
    cfg['general'].update(number_of_runs=runs, random_seed=0, keep_history=False)
    cfg['sources']['source']['intensity'] = rays
    cfg['optics']['crystal'].update(mesh_interpolate=interp, check_bragg=True)
    config, elements, flat = helpers.build(cfg)
    seeds = xrt.run_seeds(0, runs)
    dev = xrt.DeviceTrace(flat)
    dev.trace(seeds[:2], 1); dev.results()
    dev.num_out.zero_(); dev.images.zero_()
    t0 = time.time(); dev.trace(seeds, 1); meta, image = dev.results(); dt = time.time() - t0
    n_or = min(runs, 16)
    t1 = time.time(); o_num, o_img = helpers.oracle_counts(flat, seeds[:n_or], 1, threads=16); dto = time.time() - t1
    dev2 = xrt.DeviceTrace(flat); dev2.trace(seeds[:n_or], 1); m2, i2 = dev2.results()
    same = all(int(m2[nm]['num_out']) == int(o_num[k]) for k, nm in enumerate(flat.names)) and np.array_equal(
        np.concatenate([i2[nm].ravel() for nm in flat.names[1:] if i2[nm] is not None]).astype(np.int64), o_img[:flat.image_bins])
    print(json.dumps({'mesh_interpolate': interp, 'runs': runs, 'rays_per_run': rays, 'gpu_s': dt,
                      'gpu_Mphot_s': runs * rays / dt / 1e6, 'oracle_runs': n_or, 'oracle_s': dto,
                      'oracle_Mphot_s_16thr': n_or * rays / dto / 1e6,
                      'num_out': {nm: int(meta[nm]['num_out']) for nm in flat.names},
                      'gpu_equals_oracle': bool(same)}), flush=True)

"""Few-run scenes under different segment plans (development probe, not a test):
    python tests/bench_plan.py rays runs crystal|mirror  [ENV=VALUE ...]   -> best of 5 calls, ms (host clock, results on the host)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xicsrt_amd import xicsrt_raytrace as xrt, config as xconfig
rays, runs, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
for kv in sys.argv[4:]:
    k, v = kv.split('=')
    os.environ[k] = v
config = bench.spectrometer_config(rays, runs, seed=3)
if kind == 'mirror':
    config['optics']['crystal'] = {'class_name': 'XicsrtOpticPlanarMirror', 'check_size': True,
                                   'origin': [0.0, 0.0, 0.80374151], 'zaxis': [0.0, 0.59497864, -0.80374151],
                                   'xsize': 0.2, 'ysize': 0.2}
config = xconfig.get_config(config)
flat = xrt.Elements(config).flatten()
seeds = xrt.run_seeds(3, runs)
dev = xrt.DeviceTrace(flat)
os.environ['XICSRT_PLAN_DEBUG'] = '1'
KI = os.environ.get('NO_IMAGES') is None
dev.trace(seeds, 1, keep_images=KI); dev.results()
os.environ.pop('XICSRT_PLAN_DEBUG')
best = 1e9
for _ in range(5):
    dev.num_out.zero_(); dev.images.zero_()
    dev.lib.xrt_timing_begin()
    t0 = time.time(); dev.trace(seeds, 1, keep_images=KI); meta, image = dev.results(); t = time.time() - t0
    import ctypes as C
    ms, n = C.c_double(0), C.c_int64(0)
    dev.lib.xrt_timing_end(C.byref(ms), C.byref(n))
    if t < best:
        best, kms = t, ms.value
print('%s %d x %d %s: %.3f ms (trace kernels %.3f ms), %s' % (kind, runs, rays, ' '.join(sys.argv[4:]), best * 1e3, kms,
                                                            [int(meta[n]['num_out']) for n in flat.names]), flush=True)
if os.environ.get('UNIT_CLOCKS'):
    import torch, numpy as np
    buf = torch.zeros(8 * 8192, dtype=torch.int64, device='cuda')
    os.environ['XICSRT_UNIT_CLOCKS'] = str(buf.data_ptr())
    dev.num_out.zero_(); dev.images.zero_()
    dev.trace(seeds, 1, keep_images=KI); dev.results()
    os.environ.pop('XICSRT_UNIT_CLOCKS')
    c = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
    c = c[c[:, 0] > 0]
    t0 = c[:, 0].min()
    print('  first unit started at wall clock %d (x10 ns, mod 1e8)' % (int(t0) % 100000000))
    c = (c - t0) / 100.0       # 100 MHz wall clock -> microseconds
    names = ['start', 'heads ready', 'first phase done', 'look-back done', 'stream ready', 'done']
    print('  %d units; microseconds since the first unit started: mean / max' % len(c))
    for k, nm in enumerate(names):
        v = c[:, k]
        v = v[v >= 0]
        if len(v):
            print('    %-18s %9.1f %9.1f' % (nm, v.mean(), v.max()))
    d = np.diff(c[:, :6], axis=1)
    print('  phase lengths mean / max: ' + ', '.join('%s %.1f / %.1f' % (nm, d[:, k][d[:, k] > -1e6].mean(), d[:, k].max())
                                                     for k, nm in enumerate(['set-up+skip', 'first phase', 'look-back', 'stream skip', 'second phase'])))

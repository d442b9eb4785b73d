"""Per-unit survivor counts of the bench scene under two builds of the library (development: finds the units that differ).
    python tests/dev_units.py libA.so libB.so"""
import sys, os, subprocess, json
if len(sys.argv) == 3:
    outs = []
    for lib in sys.argv[1:]:
        env = dict(os.environ, XICSRT_HIP_LIB=os.path.abspath(lib))
        outs.append(json.loads(subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]))
    a, b = outs
    print('totals', a['num'], b['num'])
    nd = 0
    for u, (x, y) in enumerate(zip(a['units'], b['units'])):
        if x[0] != y[0]:
            nd += 1
            if nd < 40:
                print('unit %d (run %d, place %d): survivors %d vs %d; candidates %d, before %d / %d' % (u, u // a['upr'], u % a['upr'], x[0], y[0], x[1], x[2], y[2]))
    print(nd, 'units differ of', len(a['units']))
else:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench, torch, numpy as np
    from xicsrt_amd import xicsrt_raytrace as xrt, config as xconfig
    config = xconfig.get_config(bench.spectrometer_config(1000000, 100, seed=3))
    flat = xrt.Elements(config).flatten()
    seeds = xrt.run_seeds(3, 100)
    dev = xrt.DeviceTrace(flat)
    buf = torch.zeros(8 * 8192, dtype=torch.int64, device='cuda')
    os.environ['XICSRT_UNIT_CLOCKS'] = str(buf.data_ptr())
    dev.trace(seeds, 1); meta, image = dev.results()
    c = buf.cpu().numpy().reshape(-1, 8)
    c = c[c[:, 0] > 0]
    print(json.dumps({'num': [int(meta[n]['num_out']) for n in flat.names], 'upr': 10,
                      'units': [[int(r[6]), int(r[7] >> 32), int(r[7] & 0xffffffff)] for r in c]}))

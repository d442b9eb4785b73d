"""Randomised sweep of raytrace() end to end on the device (not a test): the same call once with the real DeviceTrace and
once with helpers.OracleDeviceTrace (the stand-in that tests/tools/fuzz_host_vs_reference.py holds to the reference
itself): totals, images, found / lost histories -- including the device-side gather of the sampled lost rays -- must
agree.  python tests/fuzz_raytrace.py [cases] [first_seed]"""
import sys, os, json, time, copy
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import logging
logging.disable(logging.WARNING)
import numpy as np, helpers, fuzz_parity as fz
import xicsrt_amd
from xicsrt_amd import xicsrt_raytrace as xrt


def same(a, b, path, out):
    if isinstance(a, dict) and isinstance(b, dict):
        if set(a.keys()) != set(b.keys()):
            out.append('%s: keys' % path)
            return
        for k in a:
            same(a[k], b[k], path + '/' + str(k), out)
        return
    if a is None or b is None:
        if not (a is None and b is None):
            out.append('%s: None vs value' % path)
        return
    x, y = np.asarray(a), np.asarray(b)
    if x.shape != y.shape:
        out.append('%s: shape %s vs %s' % (path, x.shape, y.shape))
    elif x.dtype.kind in 'biu' or y.dtype.kind in 'biu':
        if not np.array_equal(x, y):
            out.append('%s: values differ' % path)
    elif x.dtype.kind == 'f':
        if not np.array_equal(np.isnan(x), np.isnan(y)):
            out.append('%s: NaN pattern' % path)
        else:
            ok = ~np.isnan(x)
            if ok.any() and np.max(np.abs(x[ok] - y[ok])) > 1e-9 * max(1.0, float(np.max(np.abs(x[ok])))):
                out.append('%s: max diff %.3e' % (path, np.max(np.abs(x[ok] - y[ok]))))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    real = xrt.DeviceTrace
    bad = skipped = 0
    t0 = time.time()
    for case in range(n_cases):
        rs = np.random.RandomState(seed0 + case)
        cfg = fz.scene(rs)
        if 'Plasma' not in cfg['sources']['source']['class_name']:
            cfg['sources']['source']['intensity'] = int(rs.choice([1, 7, 300, 2500, 30000]))
        cfg['general'].update(keep_history=bool(rs.rand() < 0.7), history_max_lost=int(rs.choice([0, 3, 50, 10000])),
                              keep_images=bool(rs.rand() < 0.8), number_of_runs=int(rs.randint(1, 4)), number_of_iter=int(rs.randint(1, 3)))
        res = []
        for cls in (real, helpers.OracleDeviceTrace):
            xrt.DeviceTrace = cls
            try:
                res.append(xicsrt_amd.raytrace(copy.deepcopy(cfg)))
            except Exception as e:
                res.append(e)
        xrt.DeviceTrace = real
        if isinstance(res[0], Exception) or isinstance(res[1], Exception):
            if type(res[0]) is type(res[1]):
                skipped += 1
                continue
            bad += 1
            print(json.dumps({'case': seed0 + case, 'device': repr(res[0])[:200], 'stand_in': repr(res[1])[:200], 'config': cfg}), flush=True)
            continue
        out = []
        for part in ('total', 'found', 'lost'):
            same(res[0][part], res[1][part], part, out)
        if out:
            bad += 1
            print(json.dumps({'case': seed0 + case, 'diffs': out[:8], 'config': cfg}), flush=True)
        if case % 200 == 199:
            print('# %d cases, %d mismatches, %.0f s' % (case + 1, bad, time.time() - t0), flush=True)
    print(json.dumps({'cases': n_cases, 'first_seed': seed0, 'skipped': skipped, 'mismatches': bad, 'seconds': time.time() - t0}))


if __name__ == '__main__':
    main()

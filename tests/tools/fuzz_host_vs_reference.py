"""Host-layer sweep against the reference ITSELF (build container only: imports /root/reference; never runs on the GPU box).
raytrace(config) of the reference vs xicsrt_amd.raytrace(config) with helpers.OracleDeviceTrace standing in for the
device: the whole result dictionary -- totals, images, found / lost histories in their order -- on random scenes.
python tests/tools/fuzz_host_vs_reference.py [cases] [first_seed]"""
import sys, os, json, copy, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, '/root/reference')
sys.dont_write_bytecode = True
import logging
logging.disable(logging.WARNING)
import numpy as np
import xicsrt as ref                      # the reference
import helpers, fuzz_parity as fz
import xicsrt_amd
from xicsrt_amd import xicsrt_raytrace as xrt
xrt.DeviceTrace = helpers.OracleDeviceTrace


def ref_cfg(cfg):
    cfg = copy.deepcopy(cfg)
    for f in cfg.get('filters', {}).values():
        for k, v in f.items():
            if isinstance(v, list):
                f[k] = np.array(v, dtype=np.float64)
    return cfg


def same(a, b, path, out):
    if isinstance(a, dict) and isinstance(b, dict):
        if set(a.keys()) != set(b.keys()):
            out.append('%s: keys %s vs %s' % (path, sorted(a.keys()), sorted(b.keys())))
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
        return
    if x.dtype == bool or y.dtype == bool or x.dtype.kind in 'iu':
        if not np.array_equal(x, y):
            out.append('%s: values differ' % path)
        return
    if x.dtype.kind == 'f':
        if not np.array_equal(np.isnan(x), np.isnan(y)):
            out.append('%s: NaN pattern' % path)
            return
        ok = ~np.isnan(x)
        if ok.any() and np.max(np.abs(x[ok] - y[ok])) > 1e-9 * max(1.0, float(np.max(np.abs(x[ok])))):
            out.append('%s: max diff %.3e' % (path, np.max(np.abs(x[ok] - y[ok]))))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = skipped = 0
    t0 = time.time()
    for case in range(n_cases):
        rs = np.random.RandomState(seed0 + case)
        cfg = fz.scene(rs)
        cfg['sources']['source']['intensity'] = int(rs.choice([1, 7, 300, 2500])) if 'Plasma' not in cfg['sources']['source']['class_name'] else 0
        if 'Plasma' in cfg['sources']['source']['class_name']:
            del cfg['sources']['source']['intensity']
            cfg['sources']['source']['bundle_count'] = int(rs.randint(5, 40))
        cfg['general'].update(keep_history=bool(rs.randint(2)), history_max_lost=int(rs.choice([0, 3, 50, 10000])),
                              keep_images=bool(rs.rand() < 0.8), number_of_runs=int(rs.randint(1, 4)), number_of_iter=int(rs.randint(1, 3)))
        try:
            r = ref.raytrace(ref_cfg(cfg))
            r_exc = None
        except Exception as e:
            r, r_exc = None, e
        try:
            m = xicsrt_amd.raytrace(copy.deepcopy(cfg))
            m_exc = None
        except Exception as e:
            m, m_exc = None, e
        if r_exc is not None or m_exc is not None:
            if type(r_exc) is type(m_exc) or (r_exc is not None and m_exc is not None):
                skipped += 1          # both raise (the exact type may differ for out-of-scope features)
                if type(r_exc) is not type(m_exc):
                    print(json.dumps({'case': seed0 + case, 'note': 'both raise, types differ', 'ref': repr(r_exc)[:120], 'mine': repr(m_exc)[:120]}), flush=True)
                continue
            bad += 1
            print(json.dumps({'case': seed0 + case, 'ref_exc': repr(r_exc)[:200], 'mine_exc': repr(m_exc)[:200], 'config': cfg}), flush=True)
            continue
        out = []
        for part in ('total', 'found', 'lost'):
            same(r[part], m[part], part, out)
        if out:
            bad += 1
            print(json.dumps({'case': seed0 + case, 'diffs': out[:8], 'config': cfg}), flush=True)
    print(json.dumps({'cases': n_cases, 'first_seed': seed0, 'skipped': skipped, 'mismatches': bad, 'seconds': time.time() - t0}))


if __name__ == '__main__':
    main()

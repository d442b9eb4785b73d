"""Randomised sweep of the object-level API (not a test): xicsrt_amd.get_element objects on the device --
source.generate_rays(), a caller who switches rays off at random, optic.trace_global(rays), optic.make_image(rays)
-- against the CPU oracle fed with the same rays.  python tests/fuzz_objects.py [cases] [first_seed]"""
import sys, os, json, time, copy
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import logging
logging.disable(logging.WARNING)
import numpy as np, helpers, fuzz_parity as fz
import xicsrt_amd
from xicsrt_amd import xicsrt_raytrace as xrt


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = skipped = 0
    t0 = time.time()
    for case in range(n_cases):
        rs = np.random.RandomState(seed0 + case)
        cfg = fz.scene(rs)
        if 'Plasma' in cfg['sources']['source']['class_name'] or 'filters' in cfg:
            skipped += 1
            continue
        cfg['sources']['source']['intensity'] = int(rs.choice([1, 2, 255, 257, 3000, 20000]))
        keep = rs.rand(cfg['sources']['source']['intensity']) < rs.choice([0.0, 0.5, 0.9, 1.0])
        try:
            config, elements, flat = helpers.build(copy.deepcopy(cfg))
        except Exception:
            skipped += 1
            continue
        seed = int(config['general']['random_seed'])
        why = None
        try:
            np.random.seed(seed)
            source = xicsrt_amd.get_element(copy.deepcopy(cfg), 'source')
            crystal = xicsrt_amd.get_element(copy.deepcopy(cfg), 'crystal')
            rays = source.generate_rays()
            state = helpers.seed_state(seed)
            src_only = helpers.xscene.FlatScene(elements.source, [], ['source'])
            num_out, images, o_rays, o_mask, st1 = helpers.oracle_history(src_only, state)
            if not (np.array_equal(np.asarray(rays['origin']), o_rays[0, 0:3].T) or np.allclose(rays['origin'], o_rays[0, 0:3].T, rtol=0, atol=1e-14)):
                why = 'generate_rays origin'
            elif not np.allclose(rays['direction'], o_rays[0, 3:6].T, rtol=1e-12, atol=1e-15):
                why = 'generate_rays direction'
            rays['mask'][:] = keep
            ext_rays = np.empty((8, len(keep)))
            ext_rays[0:3] = np.asarray(rays['origin']).T; ext_rays[3:6] = np.asarray(rays['direction']).T
            ext_rays[6] = rays['wavelength']; ext_rays[7] = 1.0
            ext = helpers.xscene.FlatScene(helpers.xscene.ExternalRays(np.ascontiguousarray(ext_rays), np.ascontiguousarray(keep.astype(np.uint8))),
                                           [elements.optics[0]], ['source', 'crystal'])
            num_out, images, t_rays, t_mask, st2 = helpers.oracle_history(ext, st1, all_rays=True)
            hist = xrt._history_from_device(['source', 'crystal'], t_rays, t_mask, ext.optic_objs)['crystal']
            out = crystal.trace_global(rays)
            if why is None and not np.array_equal(out['mask'], hist['mask']):
                why = 'trace_global mask'
            for key in ('origin', 'direction', 'wavelength'):
                g, h = np.asarray(hist[key]), np.asarray(out[key])
                if why is None and not np.array_equal(np.isnan(h), np.isnan(g)):
                    why = 'trace_global NaN pattern of ' + key
                ok = ~np.isnan(g)
                if why is None and ok.any() and np.max(np.abs(h[ok] - g[ok])) > 1e-9 * max(1.0, float(np.max(np.abs(g[ok])))):
                    why = 'trace_global ' + key
            image = crystal.make_image(out)
            o_image = helpers.split_images(ext, images)['crystal']
            if why is None and o_image is not None and not np.array_equal(np.asarray(image).astype(np.int64), o_image):
                why = 'make_image'
            if why is None and np.random.random_sample() != helpers.state_next_double(st2):
                why = 'global np.random state after the calls'
        except Exception as e:
            why = 'raised %r' % (e,)
        if why:
            bad += 1
            print(json.dumps({'case': seed0 + case, 'why': why[:300], 'config': cfg}), flush=True)
        if case % 200 == 199:
            print('# %d cases, %d mismatches, %.0f s' % (case + 1, bad, time.time() - t0), flush=True)
    print(json.dumps({'cases': n_cases, 'first_seed': seed0, 'skipped': skipped, 'mismatches': bad, 'seconds': time.time() - t0}))


if __name__ == '__main__':
    main()

"""
Pins the CPU oracle (oracle/xrt_oracle.c) to the reference: golden vectors in
tests/golden/ were produced by importing the reference itself
(tests/golden/make_golden.py).  Integer results bit-exact, floats <= 1e-12 rel.
"""
import numpy as np
import pytest

import helpers

# golden cases whose features the oracle / device path implement so far
SKIP_PREFIX = ('H_history', 'T_tables')


def _cases(kind):
    return [n for n in helpers.golden_names(kind) if not n.startswith(SKIP_PREFIX)]


def test_mt19937_known_answers(oracle):
    import ctypes as C
    out = np.zeros(4, dtype=np.uint32)
    oracle.xrt_oracle_mt_u32(5489, out.ctypes.data, 4)
    assert out[0] == 3499211612          # init_genrand(5489) first output (MT19937 reference)
    for seed in (0, 1, 12345, 2 ** 32 - 1):
        n = 2000
        d = np.zeros(n)
        oracle.xrt_oracle_mt_double(seed, d.ctypes.data, n)
        assert np.array_equal(d, np.random.RandomState(seed).random_sample(n))
        g = np.zeros(n)
        oracle.xrt_oracle_mt_gauss(seed, g.ctypes.data, n)
        ref = np.random.RandomState(seed).standard_normal(n)
        assert np.allclose(g, ref, rtol=1e-14, atol=0)
        u = np.zeros(n, dtype=np.uint32)
        oracle.xrt_oracle_mt_u32(seed, u.ctypes.data, n)
        ref = np.random.RandomState(seed).randint(0, 2 ** 32, n, dtype=np.uint64)
        assert np.array_equal(u.astype(np.uint64), ref)


@pytest.mark.parametrize('name', _cases('trace'))
def test_trace_against_reference(name, oracle):
    cfg, gold = helpers.load_golden(name)
    config, elements, flat = helpers.build(cfg)
    state = helpers.seed_state(config['general']['random_seed'])
    num_out, images, rays, mask, st_out = helpers.oracle_history(flat, state)
    for k, nm in enumerate(flat.names):
        assert int(num_out[k]) == int(gold['num_out/' + nm]), nm
    for nm, img in helpers.split_images(flat, images).items():
        assert np.array_equal(img, gold['image/' + nm]), 'image ' + nm
    helpers.assert_history_matches_golden(flat, rays, mask, gold, rtol=helpers.rtol_for(name))
    assert helpers.state_next_double(st_out) == float(gold['next_double'])


@pytest.mark.parametrize('name', _cases('counts'))
def test_counts_against_reference(name, oracle):
    cfg, gold = helpers.load_golden(name)
    config, elements, flat = helpers.build(cfg)
    g = config['general']
    seeds = helpers.xrt.run_seeds(g['random_seed'], g['number_of_runs'])
    num_out, images = helpers.oracle_counts(flat, seeds, g['number_of_iter'], threads=2)
    for k, nm in enumerate(flat.names):
        assert int(num_out[k]) == int(gold['num_out/' + nm]), nm
    for nm, img in helpers.split_images(flat, images).items():
        assert np.array_equal(img, gold['image/' + nm]), 'image ' + nm


def _object_rays(gold, prefix):
    rays = np.empty((8, len(gold[prefix + '/mask'])))
    rays[0:3] = gold[prefix + '/origin'].T
    rays[3:6] = gold[prefix + '/direction'].T
    rays[6] = gold[prefix + '/wavelength']
    rays[7] = 1.0
    return np.ascontiguousarray(rays), np.ascontiguousarray(gold[prefix + '/mask'].astype(np.uint8))


@pytest.mark.parametrize('name', _cases('object'))
def test_object_level_tracing_against_reference(name, oracle):
    """source.generate_rays(), caller switches rays off, optic.trace_global(rays), optic.make_image(rays)."""
    cfg, gold = helpers.load_golden(name)
    config, elements, flat = helpers.build(cfg)
    state = helpers.seed_state(config['general']['random_seed'])
    src_only = helpers.xscene.FlatScene(elements.source, [], ['source'])
    num_out, images, rays, mask, st1 = helpers.oracle_history(src_only, state)
    assert np.allclose(rays[0, 0:3].T, gold['in/origin'], rtol=0, atol=1e-15)
    assert np.allclose(rays[0, 3:6].T, gold['in/direction'], rtol=1e-13, atol=1e-16)
    ext_rays, ext_mask = _object_rays(gold, 'in')
    ext = helpers.xscene.FlatScene(helpers.xscene.ExternalRays(ext_rays, ext_mask), [elements.optics[0]], ['source', 'crystal'])
    num_out, images, rays, mask, st2 = helpers.oracle_history(ext, st1, all_rays=True)
    assert int(num_out[0]) == int(gold['in/mask'].sum()) and int(num_out[1]) == int(gold['out/mask'].sum())
    hist = helpers.xrt._history_from_device(['source', 'crystal'], rays, mask, ext.optic_objs)['crystal']
    assert np.array_equal(hist['mask'], gold['out/mask'])
    for key in ('origin', 'direction', 'wavelength'):
        g, h = gold['out/' + key], hist[key]
        assert np.array_equal(np.isnan(h), np.isnan(g)), key
        ok = ~np.isnan(g)
        assert not ok.any() or np.max(np.abs(h[ok] - g[ok])) <= 1e-12 * np.max(np.abs(g[ok])), key
    assert np.array_equal(helpers.split_images(ext, images)['crystal'], gold['image'])
    assert helpers.state_next_double(st2) == float(gold['next_double'])


def test_oracle_reports_a_plasma_without_rays():
    """Every bundle outside the sightline filter: the reference's generate_rays raises 'No rays generated'
    (recorded by make_golden.py for Z_plasma_all_filtered_trace); the oracle returns an error status."""
    cfg, gold = helpers.load_golden('F_cubic_filter_trace')
    cfg['filters']['sight']['radius'] = 1e-9
    config, elements, flat = helpers.build(cfg)
    with pytest.raises(AssertionError):
        helpers.oracle_counts(flat, [int(config['general']['random_seed'])], 1)

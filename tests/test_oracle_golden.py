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
    helpers.assert_history_matches_golden(flat, rays, mask, gold)
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

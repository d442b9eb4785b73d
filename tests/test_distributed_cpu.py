"""
N > 1 path on CPU: two processes, gloo backend.

1. The exchange step on its own: each rank takes the runs `shard_runs` gives it (run i -> rank i mod world,
   seeds travel with the run index), computes its partial histogram/counters (with the CPU oracle standing
   in for the device), and ONE all-reduce of the packed integer vector gives every rank the full result,
   which must equal the single-process result and the reference's golden totals.
2. The product's own process-group branch: `xicsrt_amd.raytrace(config)` is called under the group with
   helpers.OracleDeviceTrace standing in for DeviceTrace (no GPU here), so sharding, the status agreement,
   the all-reduce, the history gather in run order and the rank-0-only saving are the code that ships.
"""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers


def _worker(rank, world, port, name, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from xicsrt_amd import xicsrt_raytrace as xrt
        cfg, gold = helpers.load_golden(name)
        config, elements, flat = helpers.build(cfg)
        g = config['general']
        seeds = xrt.run_seeds(g['random_seed'], g['number_of_runs'])
        d, r, w = xrt._dist()
        assert (r, w) == (rank, world)
        mine = [seeds[i] for i in xrt.shard_runs(g['number_of_runs'], r, w)]
        num_out, images = helpers.oracle_counts(flat, mine, g['number_of_iter']) if mine else (
            np.zeros(flat.n_elements, dtype=np.int64), np.zeros(max(flat.image_bins, 1), dtype=np.int64))
        packed = xrt.pack_counts(torch.from_numpy(num_out), torch.from_numpy(images))
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        n_all, i_all = xrt.unpack_counts(packed, flat.n_elements)
        ret[rank] = (n_all.numpy().copy(), i_all.numpy().copy())
    finally:
        dist.destroy_process_group()


def _spawn(fn, world, *args):
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as manager:
        ret = manager.dict()
        mp.spawn(fn, args=(world, port) + args + (ret,), nprocs=world, join=True)
        return dict(ret)


@pytest.mark.parametrize('name', ['C_sphere_runs', 'B_mirror_runs'])
def test_two_ranks_reduce_to_single_process_result(name):
    world = 2
    results = _spawn(_worker, world, name)
    cfg, gold = helpers.load_golden(name)
    config, elements, flat = helpers.build(cfg)
    g = config['general']
    from xicsrt_amd import xicsrt_raytrace as xrt
    seeds = xrt.run_seeds(g['random_seed'], g['number_of_runs'])
    n1, i1 = helpers.oracle_counts(flat, seeds, g['number_of_iter'])
    for rank in range(world):
        n, i = results[rank]
        assert np.array_equal(n, n1) and np.array_equal(i, i1)
    for k, nm in enumerate(flat.names):
        assert int(n1[k]) == int(gold['num_out/' + nm])


# ---- the product's raytrace() under a process group ---------------------------------------------

def _summary(out):
    """Picklable digest of a results dictionary: counters, images, histories."""
    d = {'meta': {k: int(v['num_out']) for k, v in out['total']['meta'].items()},
         'image': {k: (None if v is None else np.asarray(v)) for k, v in out['total']['image'].items()},
         'hist': {}}
    for group in ('found', 'lost'):
        for name, rays in out[group]['history'].items():
            for key in ('origin', 'direction', 'wavelength', 'mask'):
                d['hist'][(group, name, key)] = np.asarray(rays[key]).copy()
    return d


def _raytrace_worker(rank, world, port, cfg, fail_rank, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from xicsrt_amd import xicsrt_raytrace as xrt
        xrt.DeviceTrace = helpers.OracleDeviceTrace            # no GPU here: the oracle is the per-rank compute

        if fail_rank is not None:
            class Failing(helpers.OracleDeviceTrace):
                fail_code = -7 if rank == fail_rank else 0
            xrt.DeviceTrace = Failing
        try:
            out = xrt.raytrace(cfg)
            ret[rank] = ('ok', _summary(out))
        except Exception as e:          # noqa: BLE001 - the test asserts on the type name
            ret[rank] = ('raised', type(e).__name__, str(e))
    finally:
        if world > 1:
            dist.destroy_process_group()


def _config(tmp_path, keep_history, runs=5):
    import bench
    cfg = bench.spectrometer_config(4000, runs, seed=7)
    cfg['general'].update({'keep_history': keep_history, 'history_max_lost': 600, 'number_of_iter': 2,
                           'save_config': True, 'save_images': False, 'output_path': str(tmp_path),
                           'output_prefix': 'dist', 'print_results': False})
    return cfg


@pytest.mark.parametrize('keep_history', [False, True])
def test_raytrace_under_a_process_group_equals_single_process(tmp_path, keep_history):
    (tmp_path / 'one').mkdir(exist_ok=True)
    (tmp_path / 'two').mkdir(exist_ok=True)
    one = _spawn(_raytrace_worker, 1, _config(tmp_path / 'one', keep_history), None)
    two = _spawn(_raytrace_worker, 2, _config(tmp_path / 'two', keep_history), None)
    assert one[0][0] == 'ok', one[0]
    ref = one[0][1]
    assert ref['meta']['detector'] > 0
    for rank in (0, 1):
        assert two[rank][0] == 'ok', two[rank]
        got = two[rank][1]
        assert got['meta'] == ref['meta']
        for k, img in ref['image'].items():
            assert np.array_equal(got['image'][k], img)
        assert set(got['hist']) == set(ref['hist'])
        if keep_history:
            assert len(ref['hist'][('found', 'detector', 'mask')]) == ref['meta']['detector']
        for k, arr in ref['hist'].items():
            assert np.array_equal(got['hist'][k], arr, equal_nan=(arr.dtype.kind == 'f')), k
    # saving happens once, on rank 0 (a second writer would raise FileExistsError: overwrite is off)
    assert len([f for f in os.listdir(tmp_path / 'two') if 'config' in f]) == 1


@pytest.mark.parametrize('keep_history', [False, True])
def test_every_rank_raises_when_one_device_reports_an_error(tmp_path, keep_history):
    """One rank's device reports 'intensity of less than one'.  Without histories the status is read behind the runs;
    with keep_history `trace_history` raises inside the run loop, in front of every collective: either way the
    failing rank raises the reference's exception, the other one a RuntimeError, nobody hangs, nothing is saved."""
    (tmp_path / 'err').mkdir()
    res = _spawn(_raytrace_worker, 2, _config(tmp_path / 'err', keep_history), 1)
    assert res[1][0] == 'raised' and res[1][1] == 'ValueError' and 'intensity of less than one' in res[1][2]
    assert res[0][0] == 'raised' and res[0][1] == 'RuntimeError'
    assert os.listdir(tmp_path / 'err') == []

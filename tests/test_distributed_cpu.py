"""
N > 1 path on CPU: two processes, gloo backend.  Each rank takes the runs
`shard_runs` gives it (run i -> rank i mod world, seeds travel with the run
index), computes its partial histogram/counters (with the CPU oracle standing
in for the device), and ONE all-reduce of the packed integer vector gives every
rank the full result, which must equal the single-process result and the
reference's golden totals (any partition of the runs gives the same sums).
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


@pytest.mark.parametrize('name', ['C_sphere_runs', 'B_mirror_runs'])
def test_two_ranks_reduce_to_single_process_result(name):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as manager:
        ret = manager.dict()
        mp.spawn(_worker, args=(world, port, name, ret), nprocs=world, join=True)
        results = dict(ret)
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

"""
The N > 1 code on the hardware a one-GPU box has (-m gpu).

1. A FRESH child process initialises a one-rank `nccl` (= RCCL) process group and calls `xicsrt_amd.raytrace(config)`: the
   process-group branch of raytrace() -- status agreement (all-reduce MAX), the all-reduce of [num_out | image bins] on
   device tensors, the history gather -- runs through RCCL, and the result equals the non-distributed call of the parent
   and the reference's golden (xicsrt/xicsrt_multiprocessing.py:37-65 is what that branch replaces).
2. `raytrace_mp(config)` from one process (device fan-out; one device here) equals `raytrace(config)`.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

CHILD = r'''
import json, os, sys
sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch
import torch.distributed as dist
import helpers
import xicsrt_amd
from xicsrt_amd import capi
torch.cuda.set_device(0)
dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
out = {}
try:
    for name in sys.argv[3:]:
        cfg, gold = helpers.load_golden(name)
        cfg['general']['print_results'] = False
        res = xicsrt_amd.raytrace(cfg)
        d = {'meta': {k: int(v['num_out']) for k, v in res['total']['meta'].items()},
             'image': {k: (None if v is None else np.asarray(v).astype(np.int64).ravel().tolist())
                       for k, v in res['total']['image'].items()},
             'hist': {}}
        for group in ('found', 'lost'):
            for el, rays in res[group]['history'].items():
                for key in ('origin', 'direction', 'wavelength', 'mask'):
                    a = np.asarray(rays[key])
                    d['hist']['%s/%s/%s' % (group, key, el)] = (a.astype(np.float64).view(np.uint64) if a.dtype.kind == 'f'
                                                               else a.astype(np.uint64)).ravel().tolist()
        out[name] = d
    out['backend'] = dist.get_backend()
    out['world'] = dist.get_world_size()
    out['library'] = capi.LIB_PATH
finally:
    dist.destroy_process_group()
json.dump(out, open(sys.argv[2], 'w'))
'''


def _digest(res):
    d = {'meta': {k: int(v['num_out']) for k, v in res['total']['meta'].items()},
         'image': {k: (None if v is None else np.asarray(v).astype(np.int64).ravel().tolist())
                   for k, v in res['total']['image'].items()},
         'hist': {}}
    for group in ('found', 'lost'):
        for el, rays in res[group]['history'].items():
            for key in ('origin', 'direction', 'wavelength', 'mask'):
                a = np.asarray(rays[key])
                d['hist']['%s/%s/%s' % (group, key, el)] = (a.astype(np.float64).view(np.uint64) if a.dtype.kind == 'f'
                                                           else a.astype(np.uint64)).ravel().tolist()
    return d


def test_raytrace_under_a_one_rank_rccl_group_in_a_fresh_process(tmp_path):
    import xicsrt_amd
    names = ['C_sphere_runs', 'H_history_runs_iters']        # without and with keep_history
    script = tmp_path / 'child.py'
    script.write_text(CHILD)
    out = tmp_path / 'child.json'
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29600 + os.getpid() % 300),
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    res = subprocess.run([sys.executable, str(script), helpers.ROOT, str(out)] + names, env=env, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-4000:]
    child = json.load(open(out))
    assert child['backend'] == 'nccl' and child['world'] == 1
    for name in names:
        cfg, gold = helpers.load_golden(name)
        cfg['general']['print_results'] = False
        assert cfg['general'].get('keep_history', False) == (name == 'H_history_runs_iters')
        mine = _digest(xicsrt_amd.raytrace(cfg))              # this process: no group
        got = child[name]
        assert got['meta'] == mine['meta']
        assert got['image'] == mine['image']
        assert got['hist'].keys() == mine['hist'].keys()
        for k in mine['hist']:
            assert got['hist'][k] == mine['hist'][k], k        # bit for bit: same device, same library
        for nm, n in got['meta'].items():
            assert n == int(gold['num_out/' + nm]), nm
        for nm, img in got['image'].items():
            if img is not None and 'image/' + nm in gold:
                assert np.array_equal(np.asarray(img), np.asarray(gold['image/' + nm]).ravel())
        if name.startswith('H_'):
            assert len(got['hist']) > 0
            assert np.array_equal(np.asarray(got['hist']['found/mask/detector']).astype(bool), gold['found/mask/detector'])


@pytest.mark.parametrize('name', ['C_sphere_runs', 'H_history_runs_iters', 'B_mirror_runs'])
def test_raytrace_mp_from_one_process_equals_raytrace(name):
    import xicsrt_amd
    from xicsrt_amd import xicsrt_raytrace as xrt
    cfg, gold = helpers.load_golden(name)
    cfg['general']['print_results'] = False
    one = _digest(xicsrt_amd.raytrace(cfg))
    many = xicsrt_amd.raytrace_mp(cfg)                         # every visible device (one on this box)
    assert many['config']['general']['random_seed'] == many['config']['general']['output_run_suffix']
    assert _digest(many) == one
    for nm, n in one['meta'].items():
        assert n == int(gold['num_out/' + nm]), nm
    # the fan-out itself, forced onto "two devices" that are both device 0: two host threads, two DeviceTrace objects,
    # two streams of launches into the same card
    import torch
    real = (xrt.visible_devices, xrt.device_scope)
    xrt.visible_devices = lambda: 2
    xrt.device_scope = lambda index: torch.cuda.device(0)
    try:
        two = xicsrt_amd.raytrace_mp(cfg)
    finally:
        xrt.visible_devices, xrt.device_scope = real
    assert _digest(two) == one

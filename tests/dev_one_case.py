"""Development: one scene of tests/fuzz_parity.py (python tests/dev_one_case.py <seed> [--history]); POISON=<MiB>: device memory filled with 0xFF and
released to the allocator first, so that the workspace is not the zeroed memory of a fresh process.  Not a test."""
import sys, os
sys.path.insert(0, 'tests')
import logging; logging.disable(logging.WARNING)
import numpy as np, copy, helpers
import fuzz_parity
from xicsrt_amd import xicsrt_raytrace as xrt, capi
case = int(sys.argv[1])
rs = np.random.RandomState(case)
cfg = fuzz_parity.scene(rs)
config, elements, flat = helpers.build(copy.deepcopy(cfg))
g = config['general']
seeds = xrt.run_seeds(g['random_seed'], g['number_of_runs'])
n_cpu, i_cpu = helpers.oracle_counts(flat, seeds, g['number_of_iter'], threads=8)
env = {}
if rs.rand() < 0.4: env['XICSRT_SEGMENTS'] = str(int(rs.choice([1, 2, 3, 7])))
if rs.rand() < 0.2: env['XICSRT_BRAGG_BATCH_128'] = '1'
if rs.rand() < 0.1: env['XICSRT_NO_JUMP'] = '1'
if rs.rand() < 0.1: env['XICSRT_PLASMA_STAGED'] = '1'
if rs.rand() < 0.5: env['XICSRT_MOSAIC_FUSED_MIN'] = '1'
os.environ.update(env)
print('case', case, env, flush=True)
import torch
if os.environ.get('POISON'):
    x = torch.empty(int(os.environ['POISON']) << 20, dtype=torch.uint8, device='cuda'); x.fill_(0xFF); torch.cuda.synchronize(); del x
dev = xrt.DeviceTrace(flat)
capi.lib().xrt_last_path(1)
dev.trace(seeds, g['number_of_iter'])
meta, image = dev.results()
print('  path', capi.lib().xrt_last_path(1), 'equal', [int(meta[nm]['num_out']) for nm in flat.names] == [int(v) for v in n_cpu], flush=True)
if fuzz_parity.HISTORY and flat.n_rays <= 50000:
    devh = xrt.DeviceTrace(flat)
    rays, mask, st = devh.trace_history(xrt.rng_state_from_seed(seeds[0]))
    print('  history ok', rays.shape, flush=True)

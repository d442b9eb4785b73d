"""Cycles the second phase of the one-pass units spends per section of a batch (development; needs a -DXRT_DBG_BATCH build)."""
import sys, os
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
print([int(meta[n]['num_out']) for n in flat.names])
c = buf.cpu().numpy()
t = c[8000:8000 + 5000].reshape(1000, 5).astype(float)
u = c[:8000].reshape(1000, 8)
nb = (u[:, 7] >> 32) / 256.0
print('batches per unit %.0f; cycles per batch (mean over units): between %.0f, uniforms %.0f, test %.0f, survivor scan %.0f, drain %.0f'
      % ((nb.mean(),) + tuple((t[:, k] / nb).mean() for k in range(5))))

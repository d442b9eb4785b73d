"""Large parity soak (not a test): device vs CPU oracle on the bench scene, every count and every pixel.
python tests/soak_parity.py [runs] [rays]  ->  one JSON line"""
import sys, os, time, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, helpers, bench
from xicsrt_amd import xicsrt_raytrace as xrt, config as xconfig
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
config = xconfig.get_config(bench.spectrometer_config(rays, runs, seed=0))
flat = xrt.Elements(config).flatten()
seeds = xrt.run_seeds(0, runs)
dev = xrt.DeviceTrace(flat)
t0 = time.time(); dev.trace(seeds, 1); meta, image = dev.results(); t_gpu = time.time() - t0
threads = min(os.cpu_count() or 1, 64)
t0 = time.time(); o_num, o_img = helpers.oracle_counts(flat, seeds, 1, threads=threads); t_cpu = time.time() - t0
g_img = np.concatenate([image[nm].ravel() for nm in flat.names[1:]]).astype(np.int64)
print(json.dumps({'photons': runs * rays, 'runs': runs, 'rays_per_run': rays,
                  'num_out_gpu': [int(meta[n]['num_out']) for n in flat.names], 'num_out_oracle': [int(v) for v in o_num],
                  'counts_equal': [int(meta[n]['num_out']) for n in flat.names] == [int(v) for v in o_num],
                  'pixels_equal': bool(np.array_equal(g_img, o_img[:flat.image_bins])), 'pixels': int(flat.image_bins),
                  'image_sha256': hashlib.sha256(g_img.tobytes()).hexdigest()[:16],
                  'gpu_s': t_gpu, 'oracle_s': t_cpu, 'oracle_threads': threads}))

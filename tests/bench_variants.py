"""Throughput of the fused kernel variants on bench-sized work (1000 runs x 1e6 rays).  Not a test."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xicsrt_amd import xicsrt_raytrace as xrt, config as xconfig
runs, rays = 1000, 1000000
scenes = {
    'lean: spherical crystal': {},
    'full: cylindrical crystal': {'class_name': 'XicsrtOpticCylindricalCrystal'},
    'full: toroidal crystal': {'class_name': 'XicsrtOpticToroidalCrystal', 'radius_major': 1.0, 'radius_minor': 0.5},
    'full: spherical crystal + aperture': {'aperture': [{'shape': 'circle', 'size': [0.09]}]},
    'lean: planar mirror': {'class_name': 'XicsrtOpticPlanarMirror'},
}
for label, over in scenes.items():
    config = bench.spectrometer_config(rays, runs, seed=2)
    c = config['optics']['crystal']
    c.update(over)
    if 'Mirror' in c['class_name']:
        for k in ('crystal_spacing', 'rocking_type', 'rocking_fwhm', 'radius'):
            c.pop(k, None)
    if 'Toroidal' in c['class_name'] or 'Cylindrical' in c['class_name']:
        c.pop('radius', None) if 'Toroidal' in c['class_name'] else None
    config = xconfig.get_config(config)
    flat = xrt.Elements(config).flatten()
    seeds = xrt.run_seeds(2, runs)
    dev = xrt.DeviceTrace(flat)
    dev.trace(seeds, 1); dev.results()
    dev.num_out.zero_(); dev.images.zero_()
    t0 = time.time(); dev.trace(seeds, 1); meta, image = dev.results(); dt = time.time() - t0
    print(json.dumps({'scene': label, 'ms': dt * 1e3, 'Gphot_s': runs * rays / dt / 1e9,
                      'num_out': [int(meta[n]['num_out']) for n in flat.names]}), flush=True)

"""Cost of the pixel atomics: the planar-mirror scene (BASELINE cfg2 geometry, 8.4e8 pixel hits per 1e9 photons)
and the bench scene with and without images.  Not a test."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xicsrt_amd import xicsrt_raytrace as xrt, config as xconfig
runs, rays = 1000, 1000000
for label, mirror in (('spherical crystal', False), ('planar mirror', True)):
    config = bench.spectrometer_config(rays, runs, seed=2)
    c = config['optics']['crystal']
    if mirror:
        c['class_name'] = 'XicsrtOpticPlanarMirror'
        for k in ('crystal_spacing', 'rocking_type', 'rocking_fwhm', 'radius'):
            c.pop(k, None)
    config = xconfig.get_config(config)
    flat = xrt.Elements(config).flatten()
    seeds = xrt.run_seeds(2, runs)
    dev = xrt.DeviceTrace(flat)
    for images in (True, False):
        dev.trace(seeds, 1, keep_images=images); dev.results()
        dev.num_out.zero_(); dev.images.zero_()
        t0 = time.time(); dev.trace(seeds, 1, keep_images=images); meta, image = dev.results(); dt = time.time() - t0
        print(json.dumps({'scene': label, 'images': images, 'ms': dt * 1e3, 'Gphot_s': runs * rays / dt / 1e9,
                          'num_out': [int(meta[n]['num_out']) for n in flat.names]}), flush=True)

#!/usr/bin/env python3
"""
Golden-vector generator.  TEST INFRASTRUCTURE, runs only in the build container.

Imports the *reference* XICSRT (read-only at /root/reference, v0.8.13, pure
NumPy) and records, for a list of small configurations, what the reference
produces on a fixed seed:

  kind 'trace'  : one iteration through the reference's own
                  xicsrt_raytrace._raytrace_iter (xicsrt_raytrace.py:178) with
                  keep_history=True -> the full per-element ray arrays in
                  original ray order (origin, direction, wavelength, mask),
                  per-element num_out, per-optic images and the next double
                  of the global np.random stream after the iteration (stream
                  accounting check).
  kind 'counts' : xicsrt.raytrace(config) (xicsrt_raytrace.py:28) with
                  keep_history=False -> summed num_out and images over
                  runs x iterations.

Nothing from the reference is copied: fixtures hold inputs (config as JSON) and
outputs (arrays) only.  The reference never travels to the GPU box; these .npz
files do.

Usage (from anywhere):
    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py [case ...]
"""
import sys
import os
import json
import copy
import hashlib

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = '/root/reference'

# The reference must win over any same-named package on the path.
sys.path = [REFERENCE] + [p for p in sys.path
                          if os.path.abspath(p or '.') not in
                          (os.path.dirname(os.path.dirname(HERE)),)]
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402
import logging  # noqa: E402
logging.disable(logging.WARNING)

import xicsrt  # noqa: E402
assert os.path.abspath(xicsrt.__file__).startswith(REFERENCE), xicsrt.__file__
from xicsrt import xicsrt_raytrace, xicsrt_config  # noqa: E402
from xicsrt.objects._Dispatcher import Dispatcher  # noqa: E402


# ---------------------------------------------------------------------------
# configurations (geometry of examples/example_00, example_01 cell 2 and
# testing/integrated_test_01 cell 2)
# ---------------------------------------------------------------------------

def _general(seed=0, runs=1, iters=1, history=False):
    return {
        'number_of_iter': iters,
        'number_of_runs': runs,
        'random_seed': seed,
        'keep_history': history,
        'keep_images': True,
        'print_results': False,
        'strict_config_check': True,
    }


def _source(n, spread_deg=10.0, **kw):
    s = {
        'class_name': 'XicsrtSourceDirected',
        'intensity': n,
        'wavelength': 3.9492,
        'spread': float(np.radians(spread_deg)),
        'xsize': 0.0, 'ysize': 0.0, 'zsize': 0.0,
    }
    s.update(kw)
    return s


def _crystal(class_name, **kw):
    c = {
        'class_name': class_name,
        'check_size': True,
        'origin': [0.0, 0.0, 0.80374151],
        'zaxis': [0.0, 0.59497864, -0.80374151],
        'xsize': 0.2,
        'ysize': 0.2,
    }
    c.update(kw)
    return c


_BRAGG = {'crystal_spacing': 2.45676, 'rocking_type': 'gaussian',
          'rocking_fwhm': 48.070e-6}


def _detector(**kw):
    d = {
        'class_name': 'XicsrtOpticDetector',
        'origin': [0.0, 0.76871290, 0.56904832],
        'zaxis': [0.0, -0.95641806, 0.29200084],
        'xsize': 0.4,
        'ysize': 0.2,
    }
    d.update(kw)
    return d


def cfg_example00(n, seed=0, **g):
    return {
        'general': _general(seed, **g),
        'sources': {'source': _source(n, 5.0)},
        'optics': {'detector': {
            'class_name': 'XicsrtOpticDetector',
            'origin': [0.0, 0.0, 1.0], 'zaxis': [0.0, 0.0, -1.0],
            'xsize': 0.2, 'ysize': 0.2}},
    }


def cfg_three(n, crystal, seed=0, source=None, detector=None, **g):
    return {
        'general': _general(seed, **g),
        'sources': {'source': source if source is not None else _source(n)},
        'optics': {'crystal': crystal,
                   'detector': detector if detector is not None else _detector()},
    }


def build_cases():
    C = {}

    def add(name, kind, cfg):
        C[name] = (kind, cfg)

    sph = _crystal('XicsrtOpticSphericalCrystal', radius=1.0, **_BRAGG)
    mir = _crystal('XicsrtOpticPlanarMirror')

    # --- BASELINE cfg1: example_00 ---------------------------------------
    add('A_example00_trace', 'trace', cfg_example00(500, history=True))
    add('A_example00_1e5', 'counts', cfg_example00(100000))
    # --- BASELINE cfg2: planar mirror --------------------------------------
    add('B_mirror_trace', 'trace', cfg_three(500, mir, history=True))
    add('B_mirror_1e5', 'counts', cfg_three(100000, mir))
    add('B_mirror_runs', 'counts', cfg_three(20000, mir, seed=2, runs=3))
    # --- BASELINE cfg3: spherical Bragg crystal --------------------------
    add('C_sphere_trace', 'trace', cfg_three(2000, sph, history=True))
    add('C_sphere_1e5', 'counts', cfg_three(100000, sph))
    add('C_sphere_2e5_s3', 'counts', cfg_three(200000, sph, seed=3))
    add('C_sphere_runs', 'counts', cfg_three(10000, sph, seed=5, runs=4))
    add('C_sphere_iters', 'counts', cfg_three(10000, sph, seed=7, iters=3))
    add('C_sphere_runs_iters', 'counts',
        cfg_three(5000, sph, seed=11, runs=3, iters=2))
    c = dict(sph, check_bragg=False)
    add('C_sphere_nobragg_trace', 'trace', cfg_three(500, c, history=True))
    add('C_sphere_nobragg_1e5', 'counts', cfg_three(100000, c))
    c = dict(sph, rocking_type='step', rocking_fwhm=2.0e-3)
    add('C_sphere_step_trace', 'trace', cfg_three(2000, c, history=True))
    add('C_sphere_step_1e5', 'counts', cfg_three(100000, c))
    c = dict(sph, reflectivity=0.5, rocking_fwhm=1.0e-3)
    add('C_sphere_refl_1e5', 'counts', cfg_three(100000, c))
    c = dict(sph, convex=True, check_bragg=False)
    add('C_sphere_convex_trace', 'trace', cfg_three(500, c, history=True))
    # --- other analytic optic classes (integrated_test_01 sweep) ---------
    for cls, extra in [
            ('XicsrtOpticSphericalMirror', {'radius': 1.0}),
            ('XicsrtOpticPlanarCrystal', dict(_BRAGG, check_bragg=False)),
            ('XicsrtOpticPlanarCrystal', dict(_BRAGG, rocking_fwhm=5e-2)),
            ('XicsrtOpticCylindricalCrystal', dict(_BRAGG, radius=1.0, check_bragg=False)),
            ('XicsrtOpticCylindricalMirror', {'radius': 1.0, 'convex': True}),
            ('XicsrtOpticToroidalCrystal', dict(_BRAGG, radius_major=1.0, radius_minor=0.2,
                                                check_bragg=False)),
            ]:
        tag = cls.replace('XicsrtOptic', '')
        if 'rocking_fwhm' in extra and extra['rocking_fwhm'] == 5e-2:
            tag += '_bragg'
        if extra.get('convex'):
            tag += '_convex'
        c = _crystal(cls, **extra)
        add('D_%s_trace' % tag, 'trace', cfg_three(500, c, history=True))
        add('D_%s_1e5' % tag, 'counts', cfg_three(100000, c))

    # --- sources: extended, tilted, generic, focused ---------------------
    wide = dict(sph, rocking_fwhm=5.0e-3)
    tilt = {'origin': [0.01, -0.02, 0.03],
            'zaxis': [0.0, 0.6, 0.8], 'xaxis': [1.0, 0.0, 0.0]}
    s = _source(500, 10.0, xsize=0.01, ysize=0.02, zsize=0.005)
    add('S_extended_trace', 'trace', cfg_three(500, wide, source=s, history=True, seed=7))
    s = dict(_source(500, 12.0, xsize=0.01, ysize=0.02, zsize=0.005), **tilt)
    s['direction'] = [0.05, -0.05, 1.0]
    add('S_tilted_directed_trace', 'trace', cfg_three(500, dict(sph, check_bragg=False), source=s, history=True, seed=8))
    s = dict(_source(500, 12.0, xsize=0.01, ysize=0.02, zsize=0.005),
             class_name='XicsrtSourceGeneric')
    add('S_generic_trace', 'trace', cfg_three(500, wide, source=s, history=True, seed=9))
    s = dict(_source(500, 2.0, xsize=0.05, ysize=0.05, zsize=0.05),
             class_name='XicsrtSourceFocused', target=[0.0, 0.0, 0.80374151])
    add('S_focused_trace', 'trace', cfg_three(500, wide, source=s, history=True, seed=10))
    s = dict(s, intensity=100000)
    add('S_focused_1e5', 'counts', cfg_three(100000, wide, source=s, seed=10))
    s = dict(_source(500, 10.0, xsize=0.01, ysize=0.02, zsize=0.005),
             spatial_dist='gaussian')
    add('S_gaussian_spatial_trace', 'trace', cfg_three(500, wide, source=s, history=True, seed=12))

    # --- wavelength distributions ------------------------------------------
    s = _source(2000, 10.0, wavelength_dist='uniform', wavelength_range=[3.9480, 3.9500])
    add('W_uniform_trace', 'trace', cfg_three(2000, sph, source=s, history=True, seed=21))
    s = dict(s, intensity=100000)
    add('W_uniform_1e5', 'counts', cfg_three(100000, sph, source=s, seed=21))
    s = _source(2000, 10.0, wavelength_dist='monochrome')
    add('W_monochrome_trace', 'trace', cfg_three(2000, sph, source=s, history=True, seed=22))
    s = _source(2000, 10.0, linewidth=1.129e14, temperature=1000.0, mass_number=39.948)
    add('W_voigt_trace', 'trace', cfg_three(2000, sph, source=s, history=True, seed=23))
    s = dict(s, intensity=100000)
    add('W_voigt_1e5', 'counts', cfg_three(100000, sph, source=s, seed=23))
    s = _source(2000, 10.0, temperature=1000.0, mass_number=39.948)
    add('W_normal_trace', 'trace', cfg_three(2000, sph, source=s, history=True, seed=24))
    s = dict(s, intensity=100000)
    add('W_normal_1e5', 'counts', cfg_three(100000, sph, source=s, seed=24))
    s = _source(2000, 10.0, temperature=1000.0, mass_number=39.948,
                velocity=[1.0e4, -2.0e4, 3.0e4])
    add('W_doppler_trace', 'trace', cfg_three(2000, sph, source=s, history=True, seed=25))

    # --- angular distributions ---------------------------------------------
    for dist, spread in [('isotropic_xy', [0.10, 0.05]), ('flat', 0.17),
                         ('flat_xy', [-0.1, 0.15, -0.05, 0.08])]:
        s = _source(500, 10.0, angular_dist=dist)
        s['spread'] = spread
        add('G_%s_trace' % dist, 'trace', cfg_three(500, mir, source=s, history=True, seed=31))
        s = dict(s, intensity=50000)
        add('G_%s_5e4' % dist, 'counts', cfg_three(50000, mir, source=s, seed=31))

    # --- apertures / bounds / local tracing --------------------------------
    ap = [{'shape': 'circle', 'size': [0.08]},
          {'shape': 'rectangle', 'size': [0.05, 0.02], 'origin': [0.01, 0.0], 'logic': 'not'}]
    c = dict(mir, aperture=ap)
    add('P_aperture_trace', 'trace', cfg_three(2000, c, history=True, seed=41))
    add('P_aperture_1e5', 'counts', cfg_three(100000, c, seed=41))
    ap = [{'shape': 'ellipse', 'size': [0.09, 0.05]},
          {'shape': 'square', 'size': [0.03], 'origin': [0.02, 0.01], 'logic': 'xor'},
          {'shape': 'triangle', 'vertices': [[-0.05, -0.05], [0.05, -0.05], [0.0, 0.06]],
           'logic': 'or'}]
    c = dict(mir, aperture=ap)
    add('P_aperture2_trace', 'trace', cfg_three(2000, c, history=True, seed=42))
    c = dict(mir, check_size=False)
    add('P_nosize_trace', 'trace', cfg_three(500, c, history=True, seed=43))
    c = dict(mir, trace_local=True)
    add('P_local_trace', 'trace', cfg_three(500, c, history=True, seed=44))
    c = dict(sph, zsize=0.004)
    add('P_zsize_trace', 'trace', cfg_three(500, c, history=True, seed=45))
    d = _detector(pixel_size=0.0005)
    add('P_pixels_1e5', 'counts', cfg_three(100000, sph, detector=d, seed=46))

    # --- four elements: two crystals in sequence (two ranked RNG draws) ----
    cfg = cfg_three(2000, dict(sph, rocking_fwhm=2e-3), history=True, seed=51)
    cfg['optics'] = {
        'aperture': {'class_name': 'XicsrtOpticAperture', 'origin': [0.0, 0.0, 0.4],
                     'zaxis': [0.0, 0.0, -1.0], 'xsize': 0.12, 'ysize': 0.12,
                     'aperture': [{'shape': 'circle', 'size': [0.055]}]},
        'crystal': cfg['optics']['crystal'],
        'crystal2': _crystal('XicsrtOpticPlanarCrystal',
                             origin=[0.0, 0.3826, 0.6869], zaxis=[0.0, -0.9426, -0.3342],
                             xsize=0.4, ysize=0.4, reflectivity=0.7,
                             **dict(_BRAGG, rocking_fwhm=0.2)),
        'detector': _detector(origin=[0.0, 0.2149, 0.4381], zaxis=[0.0, 0.5591, 0.8293],
                              xsize=0.4, ysize=0.4),
    }
    add('Q_four_trace', 'trace', cfg)
    cfg2 = copy.deepcopy(cfg)
    cfg2['general'].update(keep_history=False, number_of_iter=2, number_of_runs=2)
    cfg2['sources']['source']['intensity'] = 50000
    add('Q_four_counts', 'counts', cfg2)

    # --- mosaic crystals (multi-layer model: per pass 2k normals then k uniforms) ----------
    mos = dict(_BRAGG, rocking_fwhm=2.0e-3, mosaic_spread=float(np.radians(0.4)), mosaic_depth=6)
    c = _crystal('XicsrtOpticPlanarMosaicCrystal', **mos)
    add('M_planar_mosaic_trace', 'trace', cfg_three(2000, c, history=True, seed=91))
    add('M_planar_mosaic_counts', 'counts', cfg_three(30000, c, seed=91, runs=2, iters=2))
    c = _crystal('XicsrtOpticSphericalMosaicCrystal', radius=1.0, mosaic_cutoff=1e-3, **mos)
    add('M_spherical_mosaic_cutoff_trace', 'trace', cfg_three(2000, c, history=True, seed=92))
    add('M_spherical_mosaic_cutoff_1e5', 'counts', cfg_three(100000, c, seed=92))
    c = _crystal('XicsrtOpticSphericalMosaicCrystal', radius=1.0, check_bragg=False, **mos)
    add('M_spherical_mosaic_nobragg_trace', 'trace', cfg_three(500, c, history=True, seed=93))

    c = _crystal('XicsrtOpticPlanarMosaicCrystal', trace_local=True, mosaic_cutoff=1e-2, **mos)
    add('M_planar_mosaic_local_trace', 'trace', cfg_three(2000, c, history=True, seed=94))
    # a user-supplied mesh (a patch of a sphere of radius 1.2 in the optic's frame) with mosaic interaction
    gx, gy = np.meshgrid(np.linspace(-0.11, 0.11, 9), np.linspace(-0.11, 0.11, 8), indexing='ij')
    gz = 1.2 - np.sqrt(1.2 ** 2 - gx ** 2 - gy ** 2)
    pts = np.stack((gx.ravel(), gy.ravel(), gz.ravel())).T
    nrm = (np.array([0.0, 0.0, 1.2]) - pts) / 1.2
    for interp in (False, True):
        c = _crystal('XicsrtOpticMeshMosaicCrystal', mesh_points=pts.tolist(), mesh_normals=nrm.tolist(),
                     mesh_interpolate=interp, trace_local=True, **mos)
        add('M_mesh_mosaic_%s_trace' % ('interp' if interp else 'flat'), 'trace',
            cfg_three(1500, c, history=True, seed=95 + int(interp)))
    c = _crystal('XicsrtOpticMeshMosaicCrystal', mesh_points=pts.tolist(), mesh_normals=nrm.tolist(),
                 mesh_interpolate=True, mosaic_cutoff=1e-3, trace_local=True, **mos)
    add('M_mesh_mosaic_counts', 'counts', cfg_three(20000, c, seed=97, runs=2, iters=2))
    # the same mesh read in global coordinates (trace_local False): nothing lands inside the bounds,
    # and the mosaic interaction then leaves every ray untouched
    c = _crystal('XicsrtOpticMeshMosaicCrystal', mesh_points=pts.tolist(), mesh_normals=nrm.tolist(),
                 mesh_interpolate=False, **mos)
    add('M_mesh_mosaic_global_trace', 'trace', cfg_three(600, c, history=True, seed=98))

    # --- full results dictionary with histories (found / shuffled lost sample) ---
    cfg = cfg_three(3000, dict(sph, rocking_fwhm=2e-3), seed=81, runs=2, iters=2, history=True)
    cfg['general']['history_max_lost'] = 200
    add('H_history_runs_iters', 'history', cfg)
    cfg = cfg_three(2000, mir, seed=82, history=True)
    add('H_history_mirror', 'history', cfg)

    # --- plasma cube (BASELINE cfg4 shape, small) ---------------------------
    p = {'class_name': 'XicsrtPlasmaCubic', 'origin': [0.0, 0.0, 0.0],
         'xsize': 0.1, 'ysize': 0.1, 'zsize': 0.1,
         'target': [0.0, 0.0, 0.80374151], 'emissivity': 2e15 / 50, 'time_resolution': 1e-3,
         'temperature': 1000.0, 'mass_number': 39.948, 'linewidth': 0.0,
         'wavelength': 3.9492, 'spread': float(np.radians(1.0)), 'use_poisson': True,
         'bundle_count': 200, 'bundle_volume': 0.001 / 200, 'bundle_type': 'voxel'}
    add('F_plasma_trace', 'trace', cfg_three(0, sph, source=p, history=True, seed=61))
    add('F_plasma_counts', 'counts', cfg_three(0, sph, source=dict(p, emissivity=2e15 / 5),
                                               seed=61, runs=2))
    # --- per-bundle plasma models: toroidal flux geometry, profile files, spread_radius,
    #     sightline bundle filter.  Profile files are data fixtures.
    ddir = os.path.join(HERE, 'data')
    os.makedirs(ddir, exist_ok=True)
    np.savetxt(os.path.join(ddir, 'emissivity_profile.txt'),
               np.array([[0.0, 4e13], [0.15, 3.1e13], [0.3, 1.7e13], [0.45, 0.6e13], [0.55, 0.0]]))
    np.savetxt(os.path.join(ddir, 'temperature_profile.txt'),
               np.array([[0.0, 2500.0], [0.2, 1800.0], [0.4, 700.0], [0.5, 0.0], [0.6, 0.0]]))
    box = dict(p, xsize=0.3, ysize=0.3, zsize=0.16, bundle_count=150, bundle_volume=1e-6,
               spread=float(np.radians(1.5)))
    tor = dict(box, class_name='XicsrtPlasmaToroidal', major_radius=0.08, minor_radius=0.05,
               torus_origin=[0.01, -0.02, 0.0], emissivity=1.2e13, emissivity_scale=0.5, temperature_scale=2.0,
               velocity=[1.0e4, 0.0, 2.0e4], velocity_scale=1.5)
    add('F_toroidal_trace', 'trace', cfg_three(0, sph, source=tor, history=True, seed=62))
    add('F_toroidal_nopoisson_trace', 'trace', cfg_three(0, sph, source=dict(tor, use_poisson=False, emissivity=2e13),
                                                         history=True, seed=63))
    dat = dict(box, class_name='XicsrtPlasmaToroidalDatafile', major_radius=0.08, minor_radius=0.05,
               emissivity_file='tests/golden/data/emissivity_profile.txt',
               temperature_file='tests/golden/data/temperature_profile.txt',
               emissivity_scale=0.45, temperature_scale=1.0)
    del dat['emissivity'], dat['temperature']
    add('F_datafile_trace', 'trace', cfg_three(0, sph, source=dat, history=True, seed=64))
    flt = {'sight': {'class_name': 'XicsrtBundleFilterSightline', 'origin': [0.0, 0.0, 0.80374151],
                     'zaxis': [0.05, -0.02, -1.0], 'radius': 0.06}}
    cfg = cfg_three(0, sph, source=dict(dat, filters=['sight']), history=True, seed=65)
    cfg['filters'] = flt
    add('F_datafile_filter_trace', 'trace', cfg)
    cfg = cfg_three(0, sph, source=dict(dat, filters=['sight'], spread=None, spread_radius=0.03), seed=66, runs=2, iters=2)
    cfg['filters'] = flt
    add('F_datafile_filter_counts', 'counts', cfg)
    # a temperature profile together with a natural line width: one Voigt profile per bundle (the realistic
    # W7-X / ITER line shape); the profile's zero-temperature tail exercises the reference's +1 eV rule
    vgt = dict(dat, linewidth=1.0e14, wavelength_dist='voigt')
    add('F_toroidal_voigt_trace', 'trace', cfg_three(0, sph, source=vgt, history=True, seed=70))
    add('F_toroidal_voigt_counts', 'counts', cfg_three(0, dict(sph, rocking_fwhm=2e-3), source=dict(vgt, emissivity_scale=0.9),
                                                       seed=71, runs=2, iters=2))
    add('F_spread_radius_trace', 'trace', cfg_three(0, sph, source=dict(box, spread=None, spread_radius=0.02,
                                                                       emissivity=6e12), history=True, seed=67))
    cfg = cfg_three(0, sph, source=dict(box, filters=['sight'], emissivity=2e13), history=True, seed=68)
    cfg['filters'] = flt
    add('F_cubic_filter_trace', 'trace', cfg)
    add('F_generic_plasma_trace', 'trace', cfg_three(0, sph, source=dict(box, class_name='XicsrtPlasmaGeneric',
                                                                        time_resolution=3e9), history=True, seed=69))

    # a sightline filter attached to an ordinary (extended) source: XicsrtSourceGeneric.ray_filter switches off the rays
    # that start outside the sightline; they keep their places in the arrays and in the random stream
    ext_src = _source(1500, 10.0, xsize=0.06, ysize=0.05, zsize=0.02, filters=['sight'])
    ray_flt = {'sight': {'class_name': 'XicsrtBundleFilterSightline', 'origin': [0.004, -0.003, 0.8],
                         'zaxis': [0.02, -0.01, -1.0], 'radius': 0.015}}
    cfg = cfg_three(0, dict(sph, rocking_fwhm=5e-3), source=ext_src, history=True, seed=83)
    cfg['filters'] = ray_flt
    add('R_ray_filter_trace', 'trace', cfg)
    cfg = cfg_three(0, dict(sph, rocking_fwhm=5e-3), source=dict(ext_src, intensity=40000, class_name='XicsrtSourceFocused',
                                                                 target=[0.0, 0.0, 0.80374151]), seed=84, runs=2, iters=2)
    cfg['filters'] = ray_flt
    add('R_ray_filter_counts', 'counts', cfg)
    cfg = cfg_three(0, dict(sph, rocking_fwhm=5e-3), source=ext_src, history=True, seed=85)
    cfg['filters'] = {'sight': dict(ray_flt['sight'], radius=1e-9)}
    add('R_ray_filter_all_off_trace', 'trace', cfg)
    cfg = cfg_three(0, dict(sph, rocking_fwhm=5e-3), source=dict(ext_src, intensity=4000), seed=86, runs=2, iters=2, history=True)
    cfg['filters'] = ray_flt
    cfg['general']['history_max_lost'] = 300
    add('H_history_ray_filter', 'history', cfg)
    # plasma edge cases: a single bundle, point bundles, every bundle outside the sightline filter (no rays at all),
    # small Poisson means (multiplication method) next to large ones (PTRS)
    add('Z_plasma_one_bundle_trace', 'trace', cfg_three(0, sph, source=dict(p, bundle_count=1, bundle_volume=0.001, emissivity=2e15 / 50 / 40),
                                                       history=True, seed=75))
    add('Z_plasma_point_bundles_trace', 'trace', cfg_three(0, sph, source=dict(p, bundle_type='point'), history=True, seed=76))
    cfg = cfg_three(0, sph, source=dict(box, filters=['sight'], emissivity=2e13), history=True, seed=77)
    cfg['filters'] = {'sight': dict(flt['sight'], radius=1e-9)}
    add('Z_plasma_all_filtered_trace', 'trace', cfg)
    add('Z_plasma_small_means_counts', 'counts', cfg_three(0, dict(sph, rocking_fwhm=2e-3), source=dict(p, emissivity=2e15 / 50 / 300, bundle_count=400,
                                                                                                     bundle_volume=0.001 / 400),
                                                         seed=78, runs=2, iters=2))

    # --- the two scenes of the randomised sweep (profiles/r02_fuzz_parity_c.json, seeds 9011073 / 9015074) where device and
    #     oracle differed beyond 1e-9 in the position of a LOST ray: a ray recorded 24 km away (it grazes the next plane) and a
    #     ray on a double root of the torus quartic.  Inputs: tests/golden/fuzz_outliers.json, written by tests/fuzz_parity.scene()
    for seed, c in sorted(json.load(open(os.path.join(HERE, 'fuzz_outliers.json'))).items()):
        c = copy.deepcopy(c)
        c['general'].update(keep_history=True, number_of_iter=1, number_of_runs=1)
        add('Y_fuzz_%s_trace' % seed, 'trace', c)

    # --- the reference's own integrated tests (testing/integrated_test_01 / _02 .ipynb): one base config,
    #     crystal class swapped, non-strict config check with keys that most classes do not know ------------
    def integrated(radius, rmaj, rmin, fwhm, spread_deg, size, src_size, n):
        return {'general': dict(_general(0), strict_config_check=False, save_images=False),
                'sources': {'source': _source(n, spread_deg, xsize=src_size, ysize=src_size, zsize=0.0)},
                'optics': {'crystal': dict(_crystal('XicsrtOpticPlanarMirror', xsize=size, ysize=size), radius=radius,
                                           radius_major=rmaj, radius_minor=rmin, mesh_size=[41, 41],
                                           crystal_spacing=2.45676, rocking_type='gaussian', rocking_fwhm=fwhm,
                                           check_bragg=False),
                           'detector': _detector()}}
    for cls in ('PlanarMirror', 'SphericalMirror', 'PlanarCrystal', 'SphericalCrystal', 'CylindricalCrystal',
                'ToroidalCrystal', 'MeshSphericalCrystal', 'MeshCylindricalCrystal', 'MeshToroidalCrystal',
                'PlanarMosaicCrystal', 'SphericalMosaicCrystal'):
        cfg = integrated(1.0, 1.0, 0.2, 48.070e-6, 10.0, 0.2, 0.0, 10000)
        cfg['optics']['crystal']['class_name'] = 'XicsrtOptic' + cls
        add('I1_' + cls, 'counts', cfg)
    for cls in ('PlanarCrystal', 'SphericalCrystal', 'CylindricalCrystal', 'ToroidalCrystal',
                'MeshSphericalCrystal', 'MeshCylindricalCrystal', 'MeshToroidalCrystal'):
        cfg = integrated(1e5, 1e5, 0.5e5, 48.070e-5, 5.0, 0.1, 0.10, 2500)
        cfg['general']['keep_history'] = True
        cfg['optics']['crystal']['class_name'] = 'XicsrtOptic' + cls
        add('I2_' + cls + '_trace', 'trace', cfg)

    # --- object-level API: generate_rays / trace_global / make_image on a caller's ray array ----
    add('O_object_sphere', 'object', cfg_three(4000, dict(sph, rocking_fwhm=2e-3), seed=91))
    add('O_object_mirror_local', 'object', cfg_three(3000, dict(mir, trace_local=True), seed=92))
    # the caller switches every ray off / all but one off before trace_global (the Bragg element then draws 0 / <= 1 uniforms)
    for tag in ('all_off', 'one_on'):
        c = cfg_three(300, dict(sph, rocking_fwhm=5e-3), seed=93)
        c['caller_mask'] = tag
        add('O_object_sphere_' + tag, 'object', c)
    # the three steps of TraceObject.trace as separate calls
    add('O_steps_sphere', 'steps', cfg_three(3000, dict(sph, rocking_fwhm=2e-3), seed=93))
    add('O_steps_mirror_aperture', 'steps', cfg_three(3000, dict(mir, aperture=[{'shape': 'circle', 'size': [0.08]},
                                                                               {'shape': 'square', 'size': [0.05], 'logic': 'not'}]), seed=94))
    add('O_steps_cylinder_step', 'steps', cfg_three(3000, _crystal('XicsrtOpticCylindricalCrystal', radius=1.0,
                                                                   **dict(_BRAGG, rocking_type='step', rocking_fwhm=5e-3)), seed=95))
    add('O_steps_torus', 'steps', cfg_three(3000, _crystal('XicsrtOpticToroidalCrystal', radius_major=1.0, radius_minor=0.5,
                                                           **dict(_BRAGG, check_bragg=False)), seed=96))
    add('O_object_torus', 'object', cfg_three(2000, _crystal('XicsrtOpticToroidalCrystal', radius_major=1.0, radius_minor=0.5,
                                                             **dict(_BRAGG, rocking_fwhm=5e-3)), seed=93))

    # --- mesh set-up tables of the three generators (host-side parity) -----------
    for cls, extra in [('XicsrtOpticMeshToroidalCrystal', {'radius_major': 1.0, 'radius_minor': 0.2, 'mesh_size': [9, 7]}),
                       ('XicsrtOpticMeshSphericalCrystal', {'radius': 1.3, 'mesh_size': [8, 6]}),
                       ('XicsrtOpticMeshCylindricalCrystal', {'radius': 0.9, 'mesh_size': [7, 9]})]:
        c = _crystal(cls, **dict(_BRAGG, **extra))
        add('T_tables_' + cls.replace('XicsrtOpticMesh', ''), 'mesh', cfg_three(10, c, seed=1))
    c = _crystal('XicsrtOpticMeshToroidalCrystal', radius_major=1.0, radius_minor=0.2, mesh_size=[41, 41],
                 **dict(_BRAGG, rocking_fwhm=2e-3))
    add('E_mesh_interp_counts', 'counts', cfg_three(20000, c, seed=72, runs=2))
    c = _crystal('XicsrtOpticMeshSphericalCrystal', radius=1.0, mesh_size=[21, 21], **dict(_BRAGG, check_bragg=False))
    add('E_mesh_sphere_trace', 'trace', cfg_three(500, c, history=True, seed=73))
    c = _crystal('XicsrtOpticMeshCylindricalCrystal', radius=1.0, mesh_size=[21, 21], mesh_interpolate=False,
                 **dict(_BRAGG, check_bragg=False))
    add('E_mesh_cylinder_trace', 'trace', cfg_three(500, c, history=True, seed=74))

    # --- mesh optics (BASELINE cfg5 shape, small) ---------------------------
    for interp in (False, True):
        c = _crystal('XicsrtOpticMeshToroidalCrystal', radius_major=1.0, radius_minor=0.2,
                     mesh_size=[41, 41], mesh_interpolate=interp,
                     **dict(_BRAGG, check_bragg=False))
        tag = 'interp' if interp else 'flat'
        add('E_mesh_%s_trace' % tag, 'trace', cfg_three(500, c, history=True, seed=71))
    # first passes over many faces (the device then goes through an x-y grid of the faces): the fine 41 x 41 mesh searched
    # directly (mesh_refine off: 3200 faces), and an 81 x 81 mesh behind a 17 x 17 coarse level (512 faces)
    c = _crystal('XicsrtOpticMeshToroidalCrystal', radius_major=1.0, radius_minor=0.2, mesh_size=[41, 41], mesh_refine=False,
                 **dict(_BRAGG, rocking_fwhm=2e-3))
    add('E_mesh_norefine_counts', 'counts', cfg_three(20000, c, seed=75, runs=2))
    add('E_mesh_norefine_trace', 'trace', cfg_three(600, dict(c, check_bragg=False), history=True, seed=75))
    c = _crystal('XicsrtOpticMeshToroidalCrystal', radius_major=1.0, radius_minor=0.2, mesh_size=[81, 81], mesh_coarse_size=[17, 17],
                 **dict(_BRAGG, rocking_fwhm=2e-3))
    add('E_mesh_81_coarse17_counts', 'counts', cfg_three(20000, c, seed=76, runs=2))
    add('E_mesh_81_coarse17_trace', 'trace', cfg_three(600, dict(c, check_bragg=False), history=True, seed=76))
    # tiny and lopsided meshes: one quad (two faces) with the coarse level as fine as the fine one, a 3 x 2 grid,
    # a coarse mesh finer than the fine mesh, many thin columns (the nearest-point buckets degenerate)
    for tag, size, coarse, interp in (('2x2', [2, 2], [2, 2], False), ('3x2', [3, 2], [2, 2], True), ('coarse_finer', [4, 4], [9, 9], False),
                                      ('61x3', [61, 3], [5, 5], True)):
        c = _crystal('XicsrtOpticMeshToroidalCrystal', radius_major=1.0, radius_minor=0.2, mesh_size=size, mesh_coarse_size=coarse,
                     mesh_interpolate=interp, **dict(_BRAGG, check_bragg=False))
        add('E_mesh_tiny_%s_trace' % tag, 'trace', cfg_three(400, c, history=True, seed=79))
    # --- ten optics in a row (more than the eight the first device path could take): five apertures between
    #     the source and the crystal, the Bragg crystal, three apertures on the reflected beam, the detector
    cfg = cfg_three(4000, dict(sph, rocking_fwhm=2e-3), history=True, seed=55)
    crystal_at = np.array([0.0, 0.0, 0.80374151])
    det_at = np.array([0.0, 0.76871290, 0.56904832])
    out_dir = (det_at - crystal_at) / np.linalg.norm(det_at - crystal_at)
    optics = {}
    shapes = [{'shape': 'circle', 'size': [0.016]}, {'shape': 'square', 'size': [0.058]},
              {'shape': 'rectangle', 'size': [0.10, 0.085]}, {'shape': 'ellipse', 'size': [0.068, 0.06]},
              {'shape': 'circle', 'size': [0.078]}]
    for i, ap in enumerate(shapes):
        z = 0.1 * (i + 1)
        optics['ap%d' % i] = {'class_name': 'XicsrtOpticAperture', 'origin': [0.0, 0.0, z], 'zaxis': [0.0, 0.0, -1.0],
                              'xsize': 0.3, 'ysize': 0.3, 'aperture': [ap]}
    optics['crystal'] = cfg['optics']['crystal']
    for i, f in enumerate((0.25, 0.5, 0.75)):
        optics['out%d' % i] = {'class_name': 'XicsrtOpticAperture', 'origin': (crystal_at + f * (det_at - crystal_at)).tolist(),
                               'zaxis': (-out_dir).tolist(), 'xsize': 0.5, 'ysize': 0.5,
                               'aperture': [{'shape': 'circle', 'size': [0.045 - 0.008 * i]}]}
    optics['detector'] = cfg['optics']['detector']
    cfg['optics'] = optics
    add('Q_ten_trace', 'trace', cfg)
    cfg2 = copy.deepcopy(cfg)
    cfg2['general'].update(keep_history=False, number_of_iter=2, number_of_runs=3)
    cfg2['sources']['source']['intensity'] = 60000
    add('Q_ten_counts', 'counts', cfg2)

    # --- far beyond the sizes of the device structures' first form (16 optics, 8 apertures per optic, 4 filters): forty optics
    #     -- 28 apertures in front of the crystal, one of them with twelve aperture shapes, the crystal, ten more on the reflected
    #     beam, the detector -- and six sightline filters on an extended source
    cfg40 = cfg_three(1500, dict(sph, rocking_fwhm=3e-3), history=True, seed=57,
                      source=_source(1500, 8.0, xsize=0.004, ysize=0.004, zsize=0.002, filters=['f%d' % i for i in range(6)]))
    cfg40['filters'] = {'f%d' % i: {'class_name': 'XicsrtBundleFilterSightline', 'origin': [0.0002 * (i - 2), -0.0001 * i, 0.8],
                                    'zaxis': [0.0003 * (i - 3), 0.0002 * i, -1.0], 'radius': 0.0022 + 0.0002 * i} for i in range(6)}
    optics = {}
    for i in range(28):
        z = 0.02 + 0.025 * i
        aps = [{'shape': 'circle', 'size': [0.012 + 0.0042 * i]}]
        if i == 9:
            aps = [{'shape': 'circle', 'size': [0.06]}]
            for j in range(11):
                aps.append({'shape': ['circle', 'square', 'rectangle', 'ellipse'][j % 4], 'origin': [0.012 * (j - 5), 0.006 * ((j * 3) % 5 - 2)],
                            'size': [[0.004], [0.007], [0.009, 0.005], [0.006, 0.004]][j % 4],
                            'logic': ['not', 'not', 'or', 'not', 'xor', 'not', 'nor', 'xnor', 'nand', 'not', 'xor'][j]})
        optics['ap%02d' % i] = {'class_name': 'XicsrtOpticAperture', 'origin': [0.0, 0.0, z], 'zaxis': [0.0, 0.0, -1.0],
                                'xsize': 0.4, 'ysize': 0.4, 'aperture': aps}
    optics['crystal'] = cfg40['optics']['crystal']
    for i in range(10):
        f = (i + 1) / 11.0
        optics['out%02d' % i] = {'class_name': 'XicsrtOpticAperture', 'origin': (crystal_at + f * (det_at - crystal_at)).tolist(),
                                 'zaxis': (-out_dir).tolist(), 'xsize': 0.5, 'ysize': 0.5,
                                 'aperture': [{'shape': 'circle', 'size': [0.06 - 0.002 * i]}]}
    optics['detector'] = cfg40['optics']['detector']
    assert len(optics) == 40
    cfg40['optics'] = optics
    add('Q_forty_trace', 'trace', cfg40)
    cfg40c = copy.deepcopy(cfg40)
    cfg40c['general'].update(keep_history=False, number_of_iter=2, number_of_runs=2)
    cfg40c['sources']['source']['intensity'] = 40000
    add('Q_forty_counts', 'counts', cfg40c)

    # --- edge cases: empty and tiny ray arrays, tile boundaries of the device kernels (256 rays), nothing
    #     reaching the Bragg element, a single pixel, the largest scene the C ABI takes (16 optics) ------------
    for n in (0, 1, 2, 255, 256, 257):
        add('X_rays_%d_trace' % n, 'trace', cfg_three(n, dict(sph, rocking_fwhm=5e-3), history=True, seed=40 + n))
    add('X_rays_1_counts', 'counts', cfg_three(1, dict(sph, rocking_fwhm=5e-3), seed=47, runs=5, iters=3))
    add('X_all_lost_trace', 'trace', cfg_three(3000, dict(sph, xsize=1e-7, ysize=1e-7, rocking_fwhm=5e-3), history=True, seed=48))
    add('X_all_lost_counts', 'counts', cfg_three(30000, dict(sph, origin=[0.5, 0.0, 0.80374151], rocking_fwhm=5e-3),
                                                 seed=49, runs=2, iters=2))
    add('X_one_pixel_trace', 'trace', cfg_three(3000, dict(sph, rocking_fwhm=5e-3, pixel_size=0.2), history=True, seed=50,
                                                detector=_detector(pixel_size=0.4)))
    # degenerate parameters: a pencil beam, a fractional intensity, a Lorentzian only (temperature 0) and a
    # Gaussian only line, a rocking curve of zero width, a crystal without bounds, a seed near 2^32
    add('Y_spread_zero_trace', 'trace', cfg_three(300, dict(sph, rocking_fwhm=5e-3), source=_source(300, 0.0), history=True, seed=51))
    add('Y_intensity_fraction_trace', 'trace', cfg_three(0, dict(sph, rocking_fwhm=5e-3), source=_source(777.9), history=True, seed=52))
    add('Y_lorentz_only_trace', 'trace', cfg_three(0, sph, source=_source(1500, 10.0, linewidth=1.129e14, temperature=0.0,
                                                                       mass_number=39.948), history=True, seed=53))
    add('Y_step_zero_width_trace', 'trace', cfg_three(1500, dict(sph, rocking_type='step', rocking_fwhm=0.0), history=True, seed=54))
    add('Y_gauss_zero_width_trace', 'trace', cfg_three(1500, dict(sph, rocking_fwhm=0.0), history=True, seed=54))
    add('Y_big_seed_counts', 'counts', cfg_three(20000, dict(sph, rocking_fwhm=5e-3), seed=4294967290, runs=3, iters=2))
    add('Y_seed_overflow_counts', 'counts', cfg_three(2000, dict(sph, rocking_fwhm=5e-3), seed=4294967290, runs=4))
    add('Y_zero_iter_counts', 'counts', cfg_three(2000, dict(sph, rocking_fwhm=5e-3), seed=3, runs=2, iters=0))

    # found by tests/fuzz_parity.py: the z axes above, typed in with 8 digits, are 1.5e-9 short of unit length, the reference
    # takes a plane's normal as it is (no division by its length in angle_calc), and with a STEP rocking curve a ray
    # 1e-9 rad from the edge of the step then falls on the other side than with a unit normal
    add('Y_step_edge_planar_counts', 'counts', {
        'general': dict(_general(1667030425, runs=3, iters=2)),
        'sources': {'source': {'class_name': 'XicsrtSourceGeneric', 'intensity': 150000, 'wavelength': 3.9492,
                               'spread': 0.1394126057487311, 'xsize': 0.0, 'ysize': 0.0, 'zsize': 0.0,
                               'linewidth': 112900000000000.0, 'temperature': 1000.0, 'mass_number': 39.948}},
        'optics': {'crystal': {'class_name': 'XicsrtOpticPlanarCrystal', 'origin': [0.0, 0.0, 0.80374151],
                               'zaxis': [0.0, 0.59497864, -0.80374151], 'xsize': 0.17608454490703235, 'ysize': 0.10534398864297627,
                               'crystal_spacing': 2.45676, 'rocking_type': 'step', 'rocking_fwhm': 0.0001771938641382608,
                               'reflectivity': 0.9136488535067315},
                   'detector': _detector()}})

    cfg16 = copy.deepcopy(cfg)
    optics = {}
    for i in range(10):                                  # ten apertures of alternating shape in front of the crystal
        z = 0.07 * (i + 1)
        ap = [{'shape': 'circle', 'size': [0.02 + 0.011 * i]}, {'shape': 'square', 'size': [0.05 + 0.02 * i]}][i % 2]
        optics['ap%d' % i] = {'class_name': 'XicsrtOpticAperture', 'origin': [0.0, 0.0, z], 'zaxis': [0.0, 0.0, -1.0],
                              'xsize': 0.3, 'ysize': 0.3, 'aperture': [ap]}
    optics['crystal'] = cfg['optics']['crystal']
    for i, f in enumerate((0.2, 0.4, 0.6, 0.8)):
        optics['out%d' % i] = {'class_name': 'XicsrtOpticAperture', 'origin': (crystal_at + f * (det_at - crystal_at)).tolist(),
                               'zaxis': (-out_dir).tolist(), 'xsize': 0.5, 'ysize': 0.5,
                               'aperture': [{'shape': 'circle', 'size': [0.05 - 0.006 * i]}]}
    optics['detector'] = cfg['optics']['detector']
    assert len(optics) == 16
    cfg16['optics'] = optics
    cfg16['general']['random_seed'] = 56
    add('X_sixteen_trace', 'trace', cfg16)
    cfg16c = copy.deepcopy(cfg16)
    cfg16c['general'].update(keep_history=False, number_of_iter=2, number_of_runs=2)
    cfg16c['sources']['source']['intensity'] = 50000
    add('X_sixteen_counts', 'counts', cfg16c)

    # --- BASELINE.json configurations at their stated geometry, reference-sized run counts -----------------
    # cfg2: point source -> planar mirror -> detector, 1e6 rays per run
    add('B_cfg2_mirror_1e6', 'counts', cfg_three(1000000, mir, seed=81, runs=2))
    # cfg3: directed source -> spherical Bragg crystal -> detector, 1e6 rays per run
    add('C_cfg3_sphere_1e6', 'counts', cfg_three(1000000, sph, seed=82, runs=2))
    # cfg4: W7-X style plasma volume (2000 bundles per run, Poisson statistics, 1 keV) -> spherical crystal ->
    #       2D detector with 0.5 mm pixels (800 x 400 bins)
    p4 = {'class_name': 'XicsrtPlasmaCubic', 'origin': [0.0, 0.0, 0.0],
          'xsize': 0.1, 'ysize': 0.1, 'zsize': 0.1,
          'target': [0.0, 0.0, 0.80374151], 'emissivity': 2e15, 'time_resolution': 1e-3,
          'temperature': 1000.0, 'mass_number': 39.948, 'linewidth': 0.0,
          'wavelength': 3.9492, 'spread': float(np.radians(1.0)), 'use_poisson': True,
          'bundle_count': 2000, 'bundle_volume': 0.001 / 2000, 'bundle_type': 'voxel'}
    add('F_cfg4_plasma_counts', 'counts', cfg_three(0, sph, source=p4, detector=_detector(pixel_size=5e-4),
                                                    seed=83, runs=3))
    # cfg5: toroidal mesh crystal 41 x 41 (interpolated and flat), rocking-curve test on
    for interp in (False, True):
        c = _crystal('XicsrtOpticMeshToroidalCrystal', radius_major=1.0, radius_minor=0.2,
                     mesh_size=[41, 41], mesh_interpolate=interp, **_BRAGG)
        add('E_cfg5_mesh_%s_1e5' % ('interp' if interp else 'flat'), 'counts', cfg_three(100000, c, seed=84, runs=2))
    # ... and the same geometry with a rocking curve wide enough to put >= 1e3 rays per 1e5 on the detector (cfg5's own curve of
    # 48 urad reflects none or a handful from a faceted surface: its goldens hold the pixel path to very little), the fine mesh
    # searched directly as well
    for tag, extra in (('flat', dict(mesh_interpolate=False)), ('interp', dict(mesh_interpolate=True)),
                       ('norefine', dict(mesh_refine=False))):
        c = _crystal('XicsrtOpticMeshToroidalCrystal', radius_major=1.0, radius_minor=0.2, mesh_size=[41, 41],
                     **dict(_BRAGG, rocking_fwhm=2e-2), **extra)
        add('E_cfg5_wide_%s_1e5' % tag, 'counts', cfg_three(100000, c, seed=85, runs=2))
    return C


# ---------------------------------------------------------------------------

def _jsonable(obj):
    if isinstance(obj, dict):
        return {k: _jsonable(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_jsonable(v) for v in obj]
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, (np.floating,)):
        return float(obj)
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.bool_,)):
        return bool(obj)
    return obj


def _ref_cfg(cfg):
    """Deep copy for the reference: its sightline filter indexes config lists as ndarrays."""
    cfg = copy.deepcopy(cfg)
    for f in cfg.get('filters', {}).values():
        for k, v in f.items():
            if isinstance(v, list):
                f[k] = np.array(v, dtype=np.float64)
    os.chdir(os.path.dirname(os.path.dirname(HERE)))      # relative profile-file names resolve from the repo root
    return cfg


def run_trace(cfg):
    """One iteration via the reference's internals, rays kept in original order."""
    cfg = _ref_cfg(cfg)
    config = xicsrt_config.config_to_numpy(cfg)
    config = xicsrt_config.get_config(config)
    np.random.seed(config['general']['random_seed'])
    filters = Dispatcher(config, 'filters')
    filters.instantiate(); filters.setup(); filters.initialize()
    sources = Dispatcher(config, 'sources')
    sources.instantiate(); sources.apply_filters(filters)
    sources.setup(); sources.check_param(); sources.initialize()
    optics = Dispatcher(config, 'optics')
    optics.instantiate(); optics.apply_filters(filters)
    optics.setup(); optics.check_param(); optics.initialize()
    single = xicsrt_raytrace._raytrace_iter(config, sources, optics)
    out = {}
    names = list(single['meta'].keys())
    out['names'] = np.array(names)
    for k in names:
        out['num_out/' + k] = np.int64(single['meta'][k]['num_out'])
        h = single['history'][k]
        out['origin/' + k] = np.asarray(h['origin'], dtype=np.float64)
        out['direction/' + k] = np.asarray(h['direction'], dtype=np.float64)
        out['wavelength/' + k] = np.asarray(h['wavelength'], dtype=np.float64)
        out['mask/' + k] = np.asarray(h['mask'], dtype=np.bool_)
        img = single['image'].get(k)
        if img is not None:
            assert np.array_equal(img, np.round(img))
            out['image/' + k] = img.astype(np.int64)
    out['next_double'] = np.float64(np.random.random_sample())
    return out


def run_counts(cfg):
    cfg = _ref_cfg(cfg)
    res = xicsrt.raytrace(cfg)
    out = {}
    names = list(res['total']['meta'].keys())
    out['names'] = np.array(names)
    for k in names:
        out['num_out/' + k] = np.int64(res['total']['meta'][k]['num_out'])
        img = res['total']['image'].get(k)
        if img is not None:
            assert np.array_equal(img, np.round(img))
            out['image/' + k] = img.astype(np.int64)
    return out


def dump_class_defaults():
    """default_config() of every built-in element class and of the general section (the config contract)."""
    import glob
    out = {'general': _jsonable({k: v for k, v in xicsrt_config.default_config()['general'].items()
                                 if k != 'pathlist_default'})}
    base = os.path.join(REFERENCE, 'xicsrt')
    for section in ('sources', 'optics', 'filters'):
        out[section] = {}
        for path in sorted(glob.glob(os.path.join(base, section, '_Xicsrt*.py'))):
            name = os.path.splitext(os.path.basename(path))[0][1:]
            try:
                disp = Dispatcher({'general': xicsrt_config.default_config()['general'],
                                   section: {'x': {'class_name': name}}}, section)
                disp.instantiate()
                out[section][name] = _jsonable(disp.objects['x'].default_config())
            except Exception as e:
                out[section][name] = {'__error__': '%s: %s' % (type(e).__name__, e)}
    json.dump(out, open(os.path.join(HERE, 'class_defaults.json'), 'w'), indent=1, sort_keys=True)
    print('class_defaults.json:', {k: len(v) for k, v in out.items()})


def run_mesh_tables(cfg):
    """The set-up products of a mesh optic (generator output and _mesh_precalc tables)."""
    cfg = copy.deepcopy(cfg)
    obj = xicsrt.get_element(cfg, 'crystal')
    out = {}
    for key in ('mesh_points', 'mesh_normals', 'mesh_faces', 'mesh_coarse_points', 'mesh_coarse_normals',
                'mesh_coarse_faces'):
        out[key] = np.asarray(obj.param[key])
    for which in ('mesh', 'mesh_coarse'):
        m = obj.param[which]
        out[which + '/faces_normal'] = np.asarray(m['faces_normal'])
        out[which + '/p_faces_idx'] = np.asarray(m['p_faces_idx'])
        out[which + '/p_faces_mask'] = np.asarray(m['p_faces_mask'])
    tri = obj.param['mesh']['interp']['z'].tri
    out['ct_simplices'] = np.asarray(tri.simplices)
    out['ct_grad_z'] = np.asarray(obj.param['mesh']['interp']['z'].grad)
    # Clough-Tocher values on a probe grid (third-party SciPy evaluated through the reference's objects)
    xs = np.linspace(-0.08, 0.08, 9)
    xx, yy = np.meshgrid(xs, xs * 0.9)
    out['probe_xy'] = np.stack((xx.ravel(), yy.ravel())).T
    out['probe_z'] = obj.param['mesh']['interp']['z'](xx.ravel(), yy.ravel())
    out['probe_nx'] = obj.param['mesh']['interp']['normal_x'](xx.ravel(), yy.ravel())
    out['names'] = np.array(['crystal'])
    return out


def run_history(cfg):
    """xicsrt.raytrace with keep_history=True: totals + found / lost ray histories."""
    cfg = _ref_cfg(cfg)
    res = xicsrt.raytrace(cfg)
    out = {}
    names = list(res['total']['meta'].keys())
    out['names'] = np.array(names)
    for k in names:
        out['num_out/' + k] = np.int64(res['total']['meta'][k]['num_out'])
        img = res['total']['image'].get(k)
        if img is not None:
            out['image/' + k] = img.astype(np.int64)
        for group in ('found', 'lost'):
            h = res[group]['history'][k]
            assert sorted(h.keys()) == ['direction', 'mask', 'origin', 'wavelength']
            for key in ('origin', 'direction', 'wavelength', 'mask'):
                out['%s/%s/%s' % (group, key, k)] = np.asarray(h[key])
    return out


def main(argv):
    if argv[1:] == ['class_defaults']:
        dump_class_defaults()
        return
    cases = build_cases()
    want = argv[1:] or list(cases)
    index = {}
    index_path = os.path.join(HERE, 'INDEX.json')
    if os.path.exists(index_path):
        index = json.load(open(index_path))
    for name in want:
        kind, cfg = cases[name]
        try:
            out = {'trace': run_trace, 'counts': run_counts, 'history': run_history, 'mesh': run_mesh_tables,
                   'object': run_object, 'steps': run_steps}[kind](cfg)
        except Exception as e:  # reference raised: record that, it is part of the contract
            print('%-32s REFERENCE RAISED %s: %s' % (name, type(e).__name__, e))
            continue
        out['config_json'] = np.array(json.dumps(_jsonable(cfg)))
        out['kind'] = np.array(kind)
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **out)
        counts = {k[8:]: int(v) for k, v in out.items() if k.startswith('num_out/')}
        hashes = {k[6:]: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()[:16]
                  for k, v in out.items() if k.startswith('image/')}
        index[name] = {'kind': kind, 'num_out': counts, 'image_sha256_16': hashes,
                       'bytes': os.path.getsize(path)}
        print('%-32s %s' % (name, counts))
    json.dump(index, open(index_path, 'w'), indent=1, sort_keys=True)


def run_object(cfg):
    """Object-level API: source.generate_rays(), some rays switched off by the caller, then
    optic.trace_global(rays) and optic.make_image(rays) (xicsrt_public.get_element objects)."""
    cfg = _ref_cfg(cfg)
    off = cfg.pop('caller_mask', 'every_fifth_off')      # which rays the caller switches off: not a key of the config schema
    np.random.seed(cfg['general']['random_seed'])
    source = xicsrt.get_element(cfg, 'source')
    crystal = xicsrt.get_element(cfg, 'crystal')
    rays = source.generate_rays()
    if off == 'all_off':
        rays['mask'][:] = False
    elif off == 'one_on':
        rays['mask'][:] = False
        rays['mask'][7] = True
    else:
        rays['mask'][::5] = False
    out = {'names': np.array(['source', 'crystal'])}
    for k in ('origin', 'direction', 'wavelength', 'mask'):
        out['in/' + k] = np.array(rays[k])
    rays = crystal.trace_global(rays)
    for k in ('origin', 'direction', 'wavelength', 'mask'):
        out['out/' + k] = np.array(rays[k])
    img = crystal.make_image(rays)
    out['image'] = img.astype(np.int64)
    out['next_double'] = np.float64(np.random.random_sample())
    return out


def run_steps(cfg):
    """Object-level API, step by step: rays from source.generate_rays(), some switched off, then the crystal's
    intersect -> check_bounds -> interact (optics/_TraceObject.py:157-172), everything recorded after each step."""
    cfg = _ref_cfg(cfg)
    np.random.seed(cfg['general']['random_seed'])
    source = xicsrt.get_element(cfg, 'source')
    crystal = xicsrt.get_element(cfg, 'crystal')
    rays = source.generate_rays()
    rays['mask'][::7] = False
    out = {'names': np.array(['source', 'crystal'])}
    for k in ('origin', 'direction', 'wavelength', 'mask'):
        out['in/' + k] = np.array(rays[k])
    xloc, norm, mask = crystal.intersect(rays)
    out['intersect/xloc'] = np.array(xloc); out['intersect/norm'] = np.array(norm); out['intersect/mask'] = np.array(mask)
    out['intersect/rays_mask'] = np.array(rays['mask'])
    mask = crystal.check_bounds(xloc, mask)
    out['bounds/mask'] = np.array(mask)
    rays = crystal.interact(rays, xloc, norm, mask)
    for k in ('origin', 'direction', 'wavelength', 'mask'):
        out['out/' + k] = np.array(rays[k])
    out['next_double'] = np.float64(np.random.random_sample())
    return out


def dump_io_cases():
    """File-name rules, the saved-config text and a saved image of the reference's xicsrt_io (data fixtures)."""
    from xicsrt import xicsrt_io
    import tempfile
    from PIL import Image
    out = {'filenames': []}
    for general, kind, name in [({'output_path': '/tmp/x'}, 'config', None),
                                ({'output_path': 'out', 'output_prefix': 'run', 'output_suffix': 'a1'}, 'image', 'detector'),
                                ({'output_path': 'out', 'output_run_suffix': '0003', 'results_ext': '.json'}, 'results', None),
                                ({'output_path': '', 'output_prefix': None, 'output_suffix': 's'}, None, 'thing')]:
        out['filenames'].append({'general': general, 'kind': kind, 'name': name,
                                 'expected': xicsrt_io.generate_filename({'general': dict(general)}, kind, name)})
    cfg = cfg_three(1000, _crystal('XicsrtOpticSphericalCrystal', radius=1.0, **_BRAGG), seed=5)
    cfg = xicsrt_config.get_config(xicsrt_config.config_to_numpy(copy.deepcopy(cfg)))
    with tempfile.TemporaryDirectory() as tmp:
        cfg['general']['output_path'] = tmp
        xicsrt_io.save_config(cfg)
        out['config_in'] = _jsonable(cfg)
        out['config_text'] = open(os.path.join(tmp, 'xicsrt_config.json')).read().replace(tmp, '<TMP>')
        img = np.arange(12, dtype=np.float64).reshape(4, 3)
        res = {'config': cfg, 'total': {'image': {'crystal': img, 'detector': None}}}
        xicsrt_io.save_images(res)
        back = np.array(Image.open(os.path.join(tmp, 'xicsrt_crystal.tif')))
        out['image_in'] = img.tolist()
        out['image_file_pixels'] = back.tolist()
        out['image_file_dtype'] = str(back.dtype)
    json.dump(out, open(os.path.join(HERE, 'io_cases.json'), 'w'), indent=1)
    print('io_cases.json written')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'io_cases':
        dump_io_cases()
    else:
        main(sys.argv)

"""
CPU tests (-m "not gpu"): host-side logic of the drop-in boundary, the C ABI
library's exports and static scene validation (no compute without a GPU).
"""
import copy
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import helpers
from xicsrt_amd import capi, config as xconfig, scene as xscene
from xicsrt_amd import xicsrt_raytrace as xrt
import xicsrt_amd


def _jsonable(v):
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, dict):
        return {k: _jsonable(x) for k, x in v.items()}
    if isinstance(v, (tuple, list)):
        return [_jsonable(x) for x in v]
    return v


def test_general_defaults_match_reference():
    ref = json.load(open(os.path.join(helpers.GOLDEN, 'class_defaults.json')))['general']
    mine = {k: v for k, v in xconfig.default_config()['general'].items() if k != 'pathlist_default'}
    assert _jsonable(mine) == ref
    assert set(xconfig.default_config().keys()) == {'general', 'sources', 'optics', 'filters', 'scenario'}


def test_element_class_defaults_match_reference():
    """Every built-in class offered here has exactly the reference's default_config (keys and values)."""
    ref = json.load(open(os.path.join(helpers.GOLDEN, 'class_defaults.json')))
    checked = 0
    for section in ('sources', 'optics', 'filters'):
        for name, expected in ref[section].items():
            try:
                cls = xrt.find_class(name, section, [])
            except NotImplementedError:
                continue            # known reference class, not on the device path yet: must fail loudly
            except Exception:
                assert name.startswith('XicsrtPlasma'), name
                continue
            got = _jsonable(cls({'class_name': name}, initialize=False, strict=True).default_config())
            assert got == expected, name
            assert list(got.keys()) == list(got.keys())
            checked += 1
    assert checked >= 25


def test_strict_config_check_and_merge():
    cfg = {'general': {'random_seed': 1}, 'sources': {'s': {'class_name': 'XicsrtSourceDirected', 'intensity': 10,
                                                            'not_an_option': 1}}, 'optics': {}}
    with pytest.raises(Exception, match='User option not recognized: not_an_option'):
        xrt.Elements(xconfig.get_config(cfg))
    cfg['general']['strict_config_check'] = False
    el = xrt.Elements(xconfig.get_config(cfg))
    assert el.source.param['intensity'] == 10 and 'not_an_option' not in el.source.config
    merged = xconfig.get_config({'general': {'keep_history': False}, 'extra_section': {'a': 1}})
    assert merged['general']['keep_history'] is False and merged['extra_section'] == {'a': 1}
    with pytest.raises(Exception, match='Could not find XicsrtOpticNope'):
        xrt.find_class('XicsrtOpticNope', 'optics', [])
    assert xrt.find_class('XicsrtOpticMeshMosaicCrystal', 'optics', []).interact_kind == 'mosaic'
    with pytest.raises(ValueError, match='intensity of less than one'):
        xicsrt_amd.get_element({'sources': {'s': {'class_name': 'XicsrtSourceGeneric'}}}, 's')


def test_seed_schedule_is_triangular():
    # random_seed += ii, cumulative (xicsrt_raytrace.py:60-63): 5, 6, 8, 11
    assert xrt.run_seeds(5, 4) == [5, 6, 8, 11]
    assert xrt.run_seeds(0, 3) == [0, 1, 3]
    assert len(set(xrt.run_seeds(None, 3))) >= 1
    assert xrt.shard_runs(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((xrt.shard_runs(10, r, 4) for r in range(4)), [])) == list(range(10))


def test_geometry_default_axes_and_transforms():
    det = xicsrt_amd.get_element({'optics': {'d': {'class_name': 'XicsrtOpticDetector', 'zaxis': [0.0, 0.6, 0.8],
                                                   'origin': [1.0, 2.0, 3.0], 'xsize': 0.4, 'ysize': 0.2}}}, 'd')
    z = np.array([0.0, 0.6, 0.8])
    x = np.cross([0.0, 0.0, 1.0], z)
    x /= np.linalg.norm(x)
    assert np.array_equal(det.orientation, np.array([x, np.cross(z, x), z]))
    assert np.array_equal(det.xaxis, x) and np.array_equal(det.zaxis, z)
    p = np.array([[1.0, 2.5, 3.5]])
    loc = det.point_to_local(p.copy())
    assert np.allclose(det.point_to_external(loc.copy()), p)
    assert det.param['pixel_xsize'] == 100 and det.param['pixel_ysize'] == 50 and det.param['enable_image']
    zdet = xicsrt_amd.get_element({'optics': {'d': {'class_name': 'XicsrtOpticDetector'}}}, 'd')
    assert np.array_equal(zdet.xaxis, [1.0, 0.0, 0.0]) and not zdet.param['enable_image']
    with pytest.raises(ValueError, match='not orthogonal'):
        xicsrt_amd.get_element({'optics': {'d': {'class_name': 'XicsrtOpticDetector', 'xaxis': [0, 0, 1.0]}}}, 'd')


def test_scene_flattening_of_spectrometer():
    cfg, gold = helpers.load_golden('C_sphere_1e5')
    config, elements, flat = helpers.build(cfg)
    s = flat.struct
    assert flat.names == ['source', 'crystal', 'detector'] and s.n_optics == 2
    assert s.source.kind == xscene.SRC_KIND['direction'] and s.source.intensity == 100000
    assert s.source.ang[0] == np.cos(np.array([np.radians(10.0)]))[0]
    c, d = s.optics[0], s.optics[1]
    assert c.shape == xscene.SHAPE['sphere'] and c.interact == xscene.INTERACT['crystal']
    assert c.flags & xscene.F_CHECK_BRAGG and c.flags & xscene.F_IMAGE and not (c.flags & xscene.F_HAS_ZSIZE)
    assert c.two_d == 2 * 2.45676 and c.half_size[0] == 0.1
    assert list(c.center) == list(1.0 * np.array([0.0, 0.59497864, -0.80374151]) + np.array([0.0, 0.0, 0.80374151]))
    assert (d.pixel_nx, d.pixel_ny, d.pixel_xoff, d.pixel_yoff) == (100, 50, 49.5, 24.5)
    assert flat.image_slices == {'crystal': (0, 100, 100), 'detector': (10000, 100, 50)} and flat.image_bins == 15000


def test_combine_raytrace_sums_and_concatenates():
    def one(n, img, found):
        out = xrt._empty_output({'general': {}})
        out['total']['meta'] = {'source': {'num_out': 10}, 'det': {'num_out': n}}
        out['total']['image'] = {'det': img}
        for group, cnt in (('found', found), ('lost', 1)):
            for k in ('source', 'det'):
                r = xrt.RayArray()
                r.zeros(cnt)
                r['origin'][:] = n
                out[group]['history'][k] = r
        return out
    res = xrt.combine_raytrace([one(3, np.ones((2, 2)), 2), one(4, 2 * np.ones((2, 2)), 3)])
    assert res['total']['meta']['det']['num_out'] == 7 and res['total']['meta']['source']['num_out'] == 20
    assert np.array_equal(res['total']['image']['det'], 3 * np.ones((2, 2)))
    assert len(res['found']['history']['det']['mask']) == 5 and len(res['lost']['history']['det']['mask']) == 2
    assert list(res['found']['history']['det']['origin'][:, 0]) == [3, 3, 4, 4, 4]
    assert set(res['found']['history']['det'].keys()) == {'origin', 'direction', 'mask', 'wavelength'}


def test_library_exports_every_declared_symbol():
    """The C ABI library loads without a GPU and exports every function include/xicsrt_hip.h declares."""
    header = open(os.path.join(helpers.ROOT, 'include', 'xicsrt_hip.h')).read()
    body = header[header.index('/* ---- entry points'):]
    declared = set(re.findall(r'\b(xrt_[a-z_0-9]+)\s*\(', body))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    L = capi.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.xrt_abi_version() == xscene.XRT_ABI_VERSION
    assert L.xrt_sizeof_scene() == C.sizeof(xscene.Scene)


def test_device_sources_compile_without_warnings():
    """The HIP sources pass `-Wall -Werror` (host and gfx950 passes, front end only: seconds): a new warning fails here, and
    build.sh, which carries -Werror, would refuse to build."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc here')
    csrc = os.path.join(helpers.ROOT, 'xicsrt_amd', 'csrc')
    assert '-Wall' in open(os.path.join(csrc, 'build.sh')).read() and '-Werror' in open(os.path.join(csrc, 'build.sh')).read()
    res = subprocess.run([hipcc, '--offload-arch=gfx950', '-std=c++17', '-fsyntax-only', '-Wall', '-Wno-unused-function',
                          '-Werror', '-Wno-unused-command-line-argument', 'xrt_kernels.hip'],
                         cwd=csrc, capture_output=True, text=True)
    assert res.returncode == 0 and 'warning' not in res.stderr, res.stderr[-3000:]


def test_unsupported_scenes_fail_loudly():
    """No CPU fallback: features outside the device path are refused (class lookup or xrt_scene_check)."""
    L = capi.lib()
    cfg, gold = helpers.load_golden('M_planar_mosaic_trace')
    cfg['sources']['source']['intensity'] = 2 ** 29
    config, elements, flat = helpers.build(cfg)
    assert L.xrt_scene_check(flat.byref()) != 0
    assert b'2^29' in L.xrt_last_error()
    cfg, gold = helpers.load_golden('F_datafile_trace')
    cfg['sources']['source'].update(linewidth=1e13)         # one Voigt table per bundle: built on the device
    config, elements, flat = helpers.build(cfg)
    assert L.xrt_scene_check(flat.byref()) == 0
    flat.struct.source.plasma.contents.n_weideman = 0       # ... from coefficients that must be there
    assert L.xrt_scene_check(flat.byref()) != 0 and b'Weideman' in L.xrt_last_error()
    cfg, gold = helpers.load_golden('R_ray_filter_trace')   # ray filters belong to the plain sources ...
    config, elements, flat = helpers.build(cfg)
    assert flat.struct.source.n_ray_filters == 1 and L.xrt_scene_check(flat.byref()) == 0
    flat.struct.source.kind = xscene.SRC_KIND['plasma']      # ... a plasma filters its bundles
    assert L.xrt_scene_check(flat.byref()) != 0


def test_supported_scenes_validate():
    L = capi.lib()
    for name in ('A_example00_trace', 'B_mirror_trace', 'C_sphere_trace', 'D_CylindricalCrystal_trace',
                 'P_aperture2_trace', 'W_voigt_trace', 'S_focused_trace', 'G_flat_xy_trace',
                 'W_normal_trace', 'S_gaussian_spatial_trace', 'G_isotropic_xy_trace', 'Q_four_trace',
                 'P_local_trace', 'M_spherical_mosaic_cutoff_trace', 'F_plasma_trace', 'D_ToroidalCrystal_trace',
                 'E_mesh_flat_trace', 'E_mesh_interp_trace', 'E_mesh_sphere_trace', 'E_mesh_cylinder_trace',
                 'M_mesh_mosaic_interp_trace', 'M_planar_mosaic_local_trace', 'F_toroidal_trace', 'F_datafile_filter_trace', 'F_spread_radius_trace', 'F_generic_plasma_trace',
                 'R_ray_filter_trace', 'X_sixteen_trace', 'E_mesh_tiny_2x2_trace'):
        cfg, gold = helpers.load_golden(name)
        config, elements, flat = helpers.build(cfg)
        assert L.xrt_scene_check(flat.byref()) == 0, (name, L.xrt_last_error())
        assert L.xrt_workspace_bytes(flat.byref(), 10) > 10 * 4096


def test_workspace_of_a_mesh_crystal_holds_the_parked_rays_within_the_budget(monkeypatch):
    """A mesh crystal goes through three launches with the rays that hit a face parked in HBM (per ray of capacity: the direction
    -- behind a point source the origin is kept once per unit --, the face, the lists of the rays left to the list walk, per 64
    rays and per unit the number left alive: 36 B; an extended source or an interpolated mesh: 60 - 92 B).  A call whose runs
    would take more than the budget (48 GiB on a 288 GB part; a cap set through xrt_set_workspace_budget or the environment counts
    too) goes through them in equal batches, and the workspace is that of one batch -- cfg5 at its full size: 1000 runs in one
    batch, 36 GB (interpolated: 2 x 500 runs, 42 GB).  With the split switched off: the one-kernel route, no parked rays."""
    L = capi.lib()
    cfg, _ = helpers.load_golden('E_cfg5_mesh_flat_1e5')
    cfg['sources']['source']['intensity'] = 1000000
    config, elements, flat = helpers.build(cfg)
    L.xrt_workspace_bytes.restype = C.c_size_t
    cap = 1000192
    per_run = cap * 36
    full = L.xrt_workspace_bytes(flat.byref(), 1000)
    assert 1000 * per_run < full < 1000 * per_run + (4 << 30)        # one batch
    assert full < (48 << 30)
    assert L.xrt_workspace_bytes(flat.byref(), 2000) == full          # two batches of 1000
    assert full < L.xrt_workspace_bytes(flat.byref(), 4000) < (48 << 30)       # three batches of 1334
    # a single run's workspace also serves xrt_trace_history, which takes the one-pass route WITHOUT the split: 64 B per ray of
    # capacity there (found by a fuzz sweep: with the records without origin the split's region had become the smaller one)
    assert L.xrt_workspace_bytes(flat.byref(), 1) > cap * 64
    few = L.xrt_workspace_bytes(flat.byref(), 300)
    assert 300 * per_run < few < 300 * per_run + (4 << 30)           # fits as it is
    # a tighter budget: smaller batches
    L.xrt_set_workspace_budget(8 << 30)
    try:
        assert L.xrt_workspace_bytes(flat.byref(), 1000) < (8 << 30) + (2 << 30)
    finally:
        L.xrt_set_workspace_budget(0)
    assert L.xrt_workspace_bytes(flat.byref(), 1000) == full
    monkeypatch.setenv('XICSRT_WORKSPACE_BUDGET_MB', '4096')
    assert L.xrt_workspace_bytes(flat.byref(), 1000) < (6 << 30)
    monkeypatch.delenv('XICSRT_WORKSPACE_BUDGET_MB')
    monkeypatch.setenv('XICSRT_NO_MESH_SPLIT', '1')
    assert L.xrt_workspace_bytes(flat.byref(), 1000) < (8 << 30)


def test_mesh_tables_match_reference():
    """Mesh generators and the pre-computed face / Clough-Tocher tables are bit-identical to the reference's
    (optics/_ShapeMesh*.py; fixtures T_tables_* written by tests/golden/make_golden.py)."""
    names = helpers.golden_names('mesh')
    assert len(names) >= 3
    for name in names:
        cfg, gold = helpers.load_golden(name)
        obj = xicsrt_amd.get_element(cfg, 'crystal')
        for key in ('mesh_points', 'mesh_normals', 'mesh_faces', 'mesh_coarse_points', 'mesh_coarse_normals',
                    'mesh_coarse_faces'):
            assert np.array_equal(np.asarray(obj.param[key]), gold[key]), (name, key)
        for which in ('mesh', 'mesh_coarse'):
            tab = obj.param[which]
            assert np.array_equal(tab['faces_normal'], gold[which + '/faces_normal']), (name, which)
            assert np.array_equal(tab['p_faces_idx'], gold[which + '/p_faces_idx']), (name, which)
            assert np.array_equal(tab['p_faces_mask'].astype(bool), gold[which + '/p_faces_mask'].astype(bool)), (name, which)
        assert np.array_equal(obj.param['mesh']['ct_simplices'], gold['ct_simplices'])
        assert np.array_equal(obj.param['mesh']['ct_grad'][0], gold['ct_grad_z'][:, 0, :])


def test_product_does_not_reference_the_oracle():
    """The product path must not import, load or link anything under oracle/."""
    for root, _, files in os.walk(os.path.join(helpers.ROOT, 'xicsrt_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.sh')):
                text = open(os.path.join(root, f)).read()
                assert 'xrt_oracle' not in text and 'oracle/' not in text, f


def test_import_xicsrt_compatibility_layout(tmp_path):
    """compat.install(): `import xicsrt`, the upstream one-module-per-class names, and a user plug-in written
    against the reference's imports resolves to device-backed classes."""
    import sys
    from xicsrt_amd import compat
    for k in [k for k in sys.modules if k == 'xicsrt' or k.startswith('xicsrt.')]:
        del sys.modules[k]
    pkg = compat.install()
    try:
        import xicsrt
        assert xicsrt is pkg and xicsrt.raytrace is xicsrt_amd.raytrace
        from xicsrt import xicsrt_io as io2, xicsrt_config as cfg2, xicsrt_multiprocessing as mp2
        assert io2 is xicsrt_amd.xicsrt_io and cfg2 is xconfig and mp2.raytrace is xrt.raytrace_mp
        from xicsrt.optics._InteractCrystal import InteractCrystal
        from xicsrt.optics._ShapeSphere import ShapeSphere
        from xicsrt.optics._XicsrtOpticDetector import XicsrtOpticDetector
        from xicsrt.sources._XicsrtSourceFocused import XicsrtSourceFocused
        from xicsrt.objects._RayArray import RayArray
        from xicsrt.filters._XicsrtBundleFilterSightline import XicsrtBundleFilterSightline
        assert XicsrtOpticDetector is xrt.find_class('XicsrtOpticDetector', 'optics', [])
        assert XicsrtSourceFocused.cone_axis_rule == 'target' and RayArray is xrt.RayArray
        # a user plug-in in the reference's style: file _<ClassName>.py on general.pathlist
        (tmp_path / '_XicsrtOpticMyCrystal.py').write_text('\n'.join([
            'from xicsrt.optics._InteractCrystal import InteractCrystal',
            'from xicsrt.optics._ShapeSphere import ShapeSphere',
            'class XicsrtOpticMyCrystal(InteractCrystal, ShapeSphere):',
            '    def default_config(self):',
            '        config = super().default_config()',
            '        config["radius"] = 1.0',
            '        return config', '']))
        cls = xrt.find_class('XicsrtOpticMyCrystal', 'optics', [str(tmp_path)])
        assert issubclass(cls, InteractCrystal) and issubclass(cls, ShapeSphere)
        cfg, gold = helpers.load_golden('C_sphere_trace')
        cfg['general']['pathlist'] = [str(tmp_path)]
        cfg['optics']['crystal']['class_name'] = 'XicsrtOpticMyCrystal'
        config, elements, flat = helpers.build(cfg)
        assert flat.struct.optics[0].shape == xscene.SHAPE['sphere']
        with pytest.raises(ImportError):
            sys.modules['xicsrt'] = __import__('json')      # something else owns the name
            compat.install()
    finally:
        for k in [k for k in sys.modules if k == 'xicsrt' or k.startswith('xicsrt.')]:
            del sys.modules[k]


def test_jump_polynomials_satisfy_the_generator_recurrence():
    """g = t^J mod phi from the library (host-side GF(2) arithmetic, no GPU): the MT19937 state sequence obeys
    s[n + J] = XOR over the set bits j of g of s[n + j]; checked for small J, J = 19937 (first reduction) and beyond."""
    L = capi.lib()
    buf = (C.c_uint32 * 624)()

    def sequence(count):
        s = [0] * (624 + count)
        x = 19650218
        for i in range(624):
            s[i] = x
            x = (1812433253 * (x ^ (x >> 30)) + i + 1) & 0xffffffff
        for n in range(624, 624 + count):
            y = (s[n - 624] & 0x80000000) | (s[n - 623] & 0x7fffffff)
            s[n] = s[n - 227] ^ (y >> 1) ^ (0x9908b0df if y & 1 else 0)
        return s

    s = sequence(19937 + 66000)
    for J in (1, 623, 19936, 19937, 19938, 40001, 65537):
        assert L.xrt_mt_jump_poly(J, buf) == 0
        g = np.frombuffer(buf, dtype=np.uint32)
        bits = [j for j in range(19937) if (int(g[j >> 5]) >> (j & 31)) & 1]
        assert len(bits) >= 1
        for n in (1, 300, 623):
            acc = 0
            for j in bits:
                acc ^= s[n + j]
            assert acc == s[n + J], (J, n)


def test_jump_polynomials_of_a_plan_come_out_as_from_scratch():
    """A plan's jump polynomials are made in families -- t^(J + step) = t^J . t^step mod phi, one carry-less product -- and must
    be the polynomials square-and-multiply gives for every exponent: the offsets of a segmented plan (arrays x segments, chunk
    heads), duplicates and stragglers included."""
    import time
    L = capi.lib()
    N, Lseg, CH, ahead = 10_000_000, 9984, 20480, 512
    Js = [2 * (k * N + s * Lseg) - ahead for k in (3, 4) for s in range(40)] + [2 * 5 * N + c * CH - ahead for c in range(60)]
    Js += [Js[5], 7, 123456789, 2 * 5 * N + 61 * CH - ahead + 2]
    arr = (C.c_uint64 * len(Js))(*Js)
    out = (C.c_uint32 * (624 * len(Js)))()
    t0 = time.time()
    assert L.xrt_mt_jump_polys(arr, len(Js), out) == 0
    t_family = time.time() - t0
    got = np.frombuffer(out, dtype=np.uint32).reshape(len(Js), 624)
    buf = (C.c_uint32 * 624)()
    for i in list(range(0, len(Js), 7)) + [len(Js) - 4, len(Js) - 3, len(Js) - 2, len(Js) - 1]:
        assert L.xrt_mt_jump_poly(Js[i], buf) == 0
        assert np.array_equal(got[i], np.frombuffer(buf, dtype=np.uint32)), (i, Js[i])
    assert t_family < 30.0


def test_weideman_faddeeva_reproduces_the_voigt_tables_of_scipy():
    """The per-bundle Voigt tables are built on the device with Weideman's approximation of Re w(z); with the
    coefficients the host hands over, the table of tools/xicsrt_voigt.py (restated here with scipy.special.wofz)
    comes out to 1e-15, far inside the parity tolerance on wavelengths."""
    from scipy.special import wofz
    from xicsrt_amd import scene as xscene

    big_l, coeff = xscene.weideman_coefficients(40)

    def rational(z):
        zz = (big_l + 1j * z) / (big_l - 1j * z)
        return 2 * np.polyval(coeff, zz) / (big_l - 1j * z) ** 2 + (1 / np.sqrt(np.pi)) / (big_l - 1j * z)

    def table(gamma, sigma, w):
        value = 100 / 2 * np.sqrt((np.sqrt(2.0 * np.log(2.0)) * sigma) ** 2 + gamma ** 2) / 5.0
        cut = max(gamma * np.sqrt(1.0 / 1e-5 - 1.0), np.sqrt(-1 * sigma ** 2 * 2 * np.log(1e-5 * sigma * np.sqrt(2 * np.pi))))
        base = np.exp(1 / 10 * np.log(cut / value))
        bounds = np.linspace(-value, value, 1001)
        bounds = bounds * base ** np.abs(bounds / value * 10)
        x = (bounds[:-1] + bounds[1:]) / 2
        y = w((x + 1j * gamma) / np.sqrt(2) / sigma).real / np.sqrt(2 * np.pi) / sigma
        return np.cumsum(y * (bounds[1:] - bounds[:-1]))

    for gamma, sigma in ((4.14e-5, 6.47e-4), (4.14e-6, 2.05e-4), (4.14e-4, 1.12e-3), (4.14e-7, 2.05e-5), (1e-3, 1e-5)):
        a, b = table(gamma, sigma, wofz), table(gamma, sigma, rational)
        assert a[-1] > 0.99 and np.max(np.abs(a - b)) < 2e-15


def test_plasma_with_temperature_profile_and_linewidth_flattens():
    """One Voigt profile per bundle: the scene carries gamma and the approximation's coefficients."""
    cfg, _ = helpers.load_golden('F_toroidal_voigt_trace')
    config, elements, flat = helpers.build(cfg)
    pl = flat.struct.source.plasma.contents
    assert pl.n_temperature > 0 and pl.voigt_gamma > 0.0 and pl.n_weideman == 40 and pl.weideman_L > 0.0


def test_history_of_a_single_ray_does_not_alias_the_device_snapshot():
    """With one ray the transposed [3, 1] blocks of the snapshot are contiguous already: the per-element
    dictionaries must still be copies (the replay of the reference's masked updates edits them in place)."""
    from xicsrt_amd import xicsrt_raytrace as xrt
    rays = np.full((3, 8, 1), np.nan)
    rays[0, :, 0] = [0, 0, 0, 0.1, 0.2, 0.97, 3.9, 1.0]
    rays[1, :, 0] = [0.1, 0.2, 0.9, 0.1, 0.2, 0.97, 3.9, 1.0]        # lost at the second element
    mask = np.array([[True], [False], [False]])
    before = rays.copy()
    hist = xrt._history_from_device(['source', 'crystal', 'detector'], rays, mask)
    assert np.array_equal(np.isnan(before), np.isnan(rays)) and np.array_equal(before[~np.isnan(before)], rays[~np.isnan(rays)])
    # later elements: NaN origin, direction as the ray had it when it was lost
    assert np.all(np.isnan(hist['detector']['origin'])) and np.array_equal(hist['detector']['direction'][0], [0.1, 0.2, 0.97])


@pytest.fixture
def oracle_device(monkeypatch):
    """raytrace() end to end on a machine without a GPU: helpers.OracleDeviceTrace stands in for the device."""
    from xicsrt_amd import xicsrt_raytrace as xrt
    monkeypatch.setattr(xrt, 'DeviceTrace', helpers.OracleDeviceTrace)
    return xrt


@pytest.mark.parametrize('name', ['X_rays_1_counts', 'X_all_lost_counts', 'X_sixteen_counts', 'Y_big_seed_counts'])
def test_raytrace_entry_point_on_edge_cases(name, oracle_device):
    """The host side of raytrace() (seed schedule, run loop, result dictionary) on the edge-case goldens."""
    cfg, gold = helpers.load_golden(name)
    res = oracle_device.raytrace(cfg)
    for nm in res['total']['meta']:
        assert int(res['total']['meta'][nm]['num_out']) == int(gold['num_out/' + nm]), nm
    for nm, img in res['total']['image'].items():
        if img is not None:
            assert np.array_equal(np.asarray(img).astype(np.int64), gold['image/' + nm]), nm


def test_raytrace_raises_what_the_reference_raises_for_degenerate_run_settings(oracle_device):
    """xicsrt_raytrace.py:60-63 lets the triangular seed schedule run past 2^32 - 1 (np.random.seed then raises),
    :114 divides by number_of_iter: both errors are part of the behaviour (goldens Y_seed_overflow / Y_zero_iter
    record that the reference raised)."""
    cfg, _ = helpers.load_golden('Y_big_seed_counts')
    cfg = copy.deepcopy(cfg)
    cfg['general'].update(number_of_runs=4)
    with pytest.raises(ValueError, match=r'Seed must be between 0 and 2\*\*32 - 1'):
        oracle_device.raytrace(cfg)
    cfg['general'].update(number_of_runs=2, number_of_iter=0)
    with pytest.raises(ZeroDivisionError):
        oracle_device.raytrace(cfg)


@pytest.mark.parametrize('n,m,burn', [(200001, 1000, 0), (1000003, 10000, 17), (262144, 262144, 623), (300000, 0, 5), (2, 1, 0), (1, 1, 3),
                                      (50000, 3125, 1), (50000, 3126, 1)])
def test_legacy_shuffle_head_equals_numpy(n, m, burn):
    """The library's walk through np.random.shuffle's draws (sample of the lost rays, _sort_raytrace) gives numpy's
    own first m indices and leaves the generator where numpy leaves it, from any position of the state block."""
    from xicsrt_amd import xicsrt_raytrace as xrt
    a, b = np.random.RandomState(12345), np.random.RandomState(12345)
    a.random_sample(burn); b.random_sample(burn)
    a.standard_normal(1); b.standard_normal(1)          # a cached gauss value must pass through
    index = np.arange(n)
    a.shuffle(index)
    got = xrt._shuffled_head(b, n, m, library_from=0)
    assert np.array_equal(got, index[:m])
    assert a.random_sample() == b.random_sample() and a.standard_normal() == b.standard_normal()


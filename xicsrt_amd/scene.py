"""
Scene flattener: initialised element objects -> the fixed-layout C structs of
include/xicsrt_hip.h (ctypes mirrors below).

All scalar set-up arithmetic is done here with NumPy in the same way the
reference's setup()/initialize()/per-call preambles do it, so the device
receives bit-identical constants:

  source   cone axis / basis inputs        xicsrt/sources/_XicsrtSourceGeneric.py:262-292
           cos(spread), tan(spread) ...    xicsrt/tools/xicsrt_spread.py:102,235,282,173-186
           wavelength case analysis        xicsrt/sources/_XicsrtSourceGeneric.py:295-367
           Voigt CDF table                 xicsrt/tools/xicsrt_voigt.py:30-83
  optic    xsize/2 ...                     xicsrt/optics/_TraceObject.py:204-212
           sphere / cylinder centre        xicsrt/optics/_ShapeSphere.py:37-43
           2*d, fwhm/2, 2*sigma**2, pi/2   xicsrt/optics/_InteractCrystal.py:110-149
           pixel grid                      xicsrt/optics/_TraceObject.py:104-131,268-270
           aperture defaults               xicsrt/tools/xicsrt_aperture.py:83-104
"""
import ctypes as C

import numpy as np

XRT_ABI_VERSION = 19
XRT_WS_STATUS_BYTE = 64        # include/xicsrt_hip.h: the uint32 xrt_check reads (checked against xrt_status_offset())
XRT_MAX_BUNDLE_FILTERS = 16
XRT_MAX_OPTICS = 64
XRT_MAX_APERTURES = 32
XRT_HIST_COMPONENTS = 8

SRC_KIND = {'zaxis': 0, 'direction': 1, 'target': 2, 'plasma': 3, 'external': 4}
SPATIAL = {'uniform': 0, 'gaussian': 1}
ANGULAR = {'isotropic': 0, 'isotropic_xy': 1, 'flat': 2, 'flat_xy': 3}
WL_CONST, WL_UNIFORM, WL_NORMAL, WL_VOIGT = 0, 1, 2, 3
SHAPE = {'plane': 0, 'sphere': 1, 'cylinder': 2, 'torus': 3, 'mesh': 4}
INTERACT = {'none': 0, 'mirror': 1, 'crystal': 2, 'mosaic': 3}
ROCKING_STEP, ROCKING_GAUSS = 0, 1
AP_SHAPE = {'none': 0, 'circle': 1, 'square': 2, 'rectangle': 3, 'ellipse': 4, 'triangle': 5}
AP_LOGIC = {'and': 0, 'not': 1, 'or': 2, 'nand': 3, 'nor': 4, 'xor': 5, 'xnor': 6}

F_CHECK_SIZE, F_CHECK_APERTURE = 1 << 0, 1 << 1
F_HAS_XSIZE, F_HAS_YSIZE, F_HAS_ZSIZE = 1 << 2, 1 << 3, 1 << 4
F_CONVEX, F_CHECK_BRAGG, F_IMAGE, F_TRACE_LOCAL = 1 << 5, 1 << 6, 1 << 7, 1 << 8


class Aperture(C.Structure):
    _fields_ = [('shape', C.c_int32), ('logic', C.c_int32),
                ('origin', C.c_double * 2), ('size', C.c_double * 2),
                ('vertices', C.c_double * 6)]


class BundleFilter(C.Structure):
    _fields_ = [('origin', C.c_double * 3), ('zaxis', C.c_double * 3), ('radius', C.c_double)]


class Plasma(C.Structure):
    _fields_ = [('geometry', C.c_int32), ('has_spread_radius', C.c_int32), ('n_filters', C.c_int32),
                ('n_emissivity', C.c_int32), ('n_temperature', C.c_int32), ('pad', C.c_int32),
                ('torus_origin', C.c_double * 3), ('major_radius', C.c_double), ('minor_radius', C.c_double),
                ('emissivity', C.c_double), ('emissivity_scale', C.c_double), ('temperature_scale', C.c_double),
                ('emissivity_rho', C.POINTER(C.c_double)), ('emissivity_val', C.POINTER(C.c_double)),
                ('temperature_rho', C.POINTER(C.c_double)), ('temperature_val', C.POINTER(C.c_double)),
                ('spread_radius', C.c_double), ('solid_angle', C.c_double),
                ('time_resolution', C.c_double), ('bundle_volume', C.c_double), ('four_pi', C.c_double),
                ('volume_ratio', C.c_double),
                ('mass_number', C.c_double), ('amu_kg', C.c_double), ('c_squared', C.c_double), ('ev_J', C.c_double),
                ('filters', BundleFilter * XRT_MAX_BUNDLE_FILTERS),
                ('voigt_gamma', C.c_double), ('weideman_L', C.c_double), ('weideman_a', C.POINTER(C.c_double)),
                ('n_weideman', C.c_int32), ('pad2', C.c_int32)]


class Source(C.Structure):
    _fields_ = [('kind', C.c_int32), ('spatial_dist', C.c_int32),
                ('angular_dist', C.c_int32), ('wavelength_dist', C.c_int32),
                ('intensity', C.c_int64),
                ('origin', C.c_double * 3), ('orientation', C.c_double * 9),
                ('size', C.c_double * 3), ('spatial_A', C.c_double * 9),
                ('axis', C.c_double * 3), ('basis', C.c_double * 9), ('ang', C.c_double * 5),
                ('two_pi', C.c_double), ('wavelength', C.c_double),
                ('wl_a', C.c_double), ('wl_b', C.c_double),
                ('has_velocity', C.c_int32), ('voigt_n', C.c_int32),
                ('velocity', C.c_double * 3), ('light_speed', C.c_double),
                ('voigt_cdf', C.POINTER(C.c_double)), ('voigt_x', C.POINTER(C.c_double)),
                ('bundle_count', C.c_int64), ('plasma_size', C.c_double * 3),
                ('bundle_intensity', C.c_double), ('use_poisson', C.c_int32), ('pad_plasma', C.c_int32),
                ('plasma', C.POINTER(Plasma)),
                ('ext_rays', C.c_void_p), ('ext_mask', C.c_void_p),
                ('n_ray_filters', C.c_int32), ('pad_filters', C.c_int32),
                ('ray_filters', BundleFilter * XRT_MAX_BUNDLE_FILTERS)]


_PD, _PI, _PB = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)


class Mesh(C.Structure):
    _fields_ = [('n_points', C.c_int32), ('n_faces', C.c_int32), ('n_coarse_faces', C.c_int32),
                ('interpolate', C.c_int32), ('n_simplices', C.c_int32), ('pad', C.c_int32),
                ('points', _PD), ('p0', _PD), ('p1', _PD), ('p2', _PD), ('edge1', _PD), ('edge2', _PD),
                ('faces_normal', _PD), ('faces_area', _PD), ('p_faces_idx', _PI), ('p_faces_mask', _PB),
                ('c_p0', _PD), ('c_edge1', _PD), ('c_edge2', _PD),
                ('ct_simplices', _PI), ('ct_neighbors', _PI), ('ct_transform', _PD), ('ct_points', _PD),
                ('ct_values', _PD), ('ct_grad', _PD), ('ct_vertex_simplex', _PI)]


class Optic(C.Structure):
    _fields_ = [('shape', C.c_int32), ('interact', C.c_int32),
                ('flags', C.c_int32), ('rocking_type', C.c_int32),
                ('origin', C.c_double * 3), ('orientation', C.c_double * 9),
                ('half_size', C.c_double * 3),
                ('radius', C.c_double), ('radius2', C.c_double), ('center', C.c_double * 3),
                ('torus_major', C.c_double), ('torus_minor', C.c_double), ('torus_k', C.c_double * 5),
                ('torus_root', C.c_int32), ('n_apertures', C.c_int32),
                ('two_d', C.c_double), ('reflectivity', C.c_double),
                ('rocking_half_fwhm', C.c_double), ('rocking_2sigma2', C.c_double),
                ('half_pi', C.c_double),
                ('mosaic_depth', C.c_int32), ('mosaic_has_cutoff', C.c_int32),
                ('mosaic_cutoff_angle', C.c_double), ('mosaic_A', C.c_double * 4),
                ('pixel_size', C.c_double), ('pixel_xoff', C.c_double), ('pixel_yoff', C.c_double),
                ('pixel_nx', C.c_int32), ('pixel_ny', C.c_int32),
                ('image_offset', C.c_int64), ('mesh', C.POINTER(Mesh)),
                ('apertures', Aperture * XRT_MAX_APERTURES)]


class Scene(C.Structure):
    _fields_ = [('source', Source), ('n_optics', C.c_int32), ('reserved', C.c_int32),
                ('image_bins', C.c_int64), ('optics', Optic * XRT_MAX_OPTICS)]


class RngState(C.Structure):
    """numpy legacy RandomState: MT19937 key, position, cached gaussian."""
    _fields_ = [('key', C.c_uint32 * 624), ('pos', C.c_int32), ('has_gauss', C.c_int32),
                ('gauss', C.c_double)]


class SceneError(NotImplementedError):
    """The scene uses a feature the device path does not implement."""


def _vec(dst, values):
    values = np.asarray(values, dtype=np.float64).ravel()
    if len(values) != len(dst):
        raise ValueError('expected %d values, got %d' % (len(dst), len(values)))
    for i, v in enumerate(values):
        dst[i] = float(v)


def _parse_spread_single(spread):
    spread = np.array([spread]) if np.isscalar(spread) else np.asarray(spread)
    if spread.ndim == 0:
        spread = spread.reshape(1)
    if len(spread) != 1:
        raise Exception('Spread must be a scalar or one element array.')
    return spread


def _parse_spread_xy(spread):
    spread = np.array([spread]) if np.isscalar(spread) else np.asarray(spread)
    if spread.ndim == 0:
        spread = spread.reshape(1)
    if len(spread) == 1:
        return [-spread[0], spread[0], -spread[0], spread[0]]
    if len(spread) == 2:
        return [-spread[0], spread[0], -spread[1], spread[1]]
    if len(spread) == 4:
        return [spread[0], spread[1], spread[2], spread[3]]
    raise Exception('Spread must have 1, 2 or 3 elements. See docstring.')


def voigt_cdf_table(gamma, sigma, gridsize=1000, cutoff=1e-5):
    """Tabulated Voigt CDF on a variable-density grid (xicsrt_voigt.py:30-83); returns (x, cdf)."""
    from scipy.special import wofz
    gridsize_min = 100
    fraction = 0.5
    gauss_hwfm = np.sqrt(2.0 * np.log(1.0 / fraction)) * sigma
    lorentz_hwfm = gamma * np.sqrt(1.0 / fraction - 1.0)
    hwfm_max = np.sqrt(gauss_hwfm ** 2 + lorentz_hwfm ** 2)
    min_spacing = hwfm_max / 5.0
    value = gridsize_min / 2 * min_spacing
    lorentz_cutoff = gamma * np.sqrt(1.0 / cutoff - 1.0)
    gauss_cutoff = np.sqrt(-1 * sigma ** 2 * 2 * np.log(cutoff * sigma * np.sqrt(2 * np.pi)))
    value_cutoff = max(lorentz_cutoff, gauss_cutoff)
    base = np.exp(1 / 10 * np.log(value_cutoff / value))
    bounds = np.linspace(-value, value, gridsize + 1)
    bounds = bounds * base ** np.abs(bounds / value * 10)
    cdf_x = (bounds[:-1] + bounds[1:]) / 2
    z = (cdf_x - 0.0 + 1j * gamma) / np.sqrt(2) / sigma
    cdf_y = wofz(z).real / np.sqrt(2 * np.pi) / sigma * 1.0
    cdf = np.cumsum(cdf_y * (bounds[1:] - bounds[:-1]))
    if np.sum((cdf > 0.25) & (cdf < 0.75)) < 3:
        raise Exception('Voight CDF calculation does not have enough resolution.')
    if np.max(cdf) < 0.99:
        raise Exception('Voight CDF calculation domain too small.')
    return bounds[1:], cdf


PLASMA_GEOMETRY = {'box': 0, 'toroidal': 1}


def _box_distance(obj, point):
    """Smallest distance from `point` to the plasma box (0 inside): bounds the per-bundle spread."""
    p = obj.param
    loc = obj.point_to_local(np.array(point, dtype=np.float64))
    half = np.array([p['xsize'], p['ysize'], p['zsize']], dtype=np.float64) / 2
    return float(np.linalg.norm(np.maximum(np.abs(loc) - half, 0.0)))


def weideman_coefficients(n):
    """
    Coefficients of Weideman's rational approximation of the Faddeeva function for Im z >= 0
    (J.A.C. Weideman, SIAM J. Numer. Anal. 31 (1994) 1497-1518, the `cef` routine of the paper):
    w(z) ~ 2 p(Z) / (L - i z)**2 + (1 / sqrt(pi)) / (L - i z),  Z = (L + i z) / (L - i z),  L = sqrt(n / sqrt(2)),
    p the polynomial with the returned coefficients (highest power first).  With n = 40 the CDF tables of
    tools/xicsrt_voigt.py come out within 3e-16 of the ones scipy.special.wofz gives (checked in tests).
    """
    m = 2 * n
    k = np.arange(-m + 1, m)
    big_l = np.sqrt(n / np.sqrt(2.0))
    t = big_l * np.tan(k * np.pi / m / 2)
    f = np.concatenate(([0.0], np.exp(-t ** 2) * (big_l ** 2 + t ** 2)))
    a = np.real(np.fft.fft(np.fft.fftshift(f))) / (2 * m)
    return big_l, np.ascontiguousarray(a[1:n + 1][::-1], dtype=np.float64)


def plasma_as_source_param(obj, keep):
    """
    The parameters every bundle's XicsrtSourceFocused gets from a plasma object
    (sources/_XicsrtPlasmaGeneric.py:289-336) plus the bundle bookkeeping.

    Returns (q, B, lam, plasma): q the focused-source parameters shared by the bundles, B the
    bundle count, lam the expected rays per bundle when it is one constant, plasma a `Plasma`
    struct when mask / emissivity / temperature / spread depend on the bundle centre (profile
    tables, spread_radius, sightline filters), else None.
    """
    import scipy.constants as const
    p = obj.param
    model = obj.bundle_model()
    ad = 'isotropic' if p['angular_dist'] is None else str(p['angular_dist']).lower()
    if ad != 'isotropic':
        raise Exception(f'Solid angle calculation for "{ad}" is not available.')
    if p['bundle_type'] == 'point':
        voxel_size = 0.0
    elif p['bundle_type'] == 'voxel':
        voxel_size = p['bundle_volume'] ** (1 / 3)
    else:
        raise Exception('bundle_type not understood: {}'.format(p['bundle_type']))
    B = int(p['bundle_count'])
    filters = [f for f in obj.filter_objects if f.filter_kind != 'none']
    if len(filters) > XRT_MAX_BUNDLE_FILTERS:
        raise SceneError('more than %d bundle filters on one plasma' % XRT_MAX_BUNDLE_FILTERS)
    per_bundle = (p['spread_radius'] is not None or len(filters) > 0 or
                  model['emissivity_profile'] is not None or model['temperature_profile'] is not None)

    # setup_bundle_spread (:206-231): one spread for all bundles, or its largest possible value
    if p['spread_radius'] is not None:
        dmin = _box_distance(obj, p['target'])
        spread_b = np.arctan(p['spread_radius'] / dmin) if dmin > 0 else np.pi / 2
    else:
        if p['spread'] is None:
            raise TypeError("unsupported operand type(s) for /: 'NoneType' and 'int'")   # what the reference hits
        spread = np.zeros([B], dtype=np.float64)
        spread[:] = p['spread']
        spread_b = spread[0]
    theta = _parse_spread_single(spread_b)
    solid_angle = np.zeros([1], dtype=np.float64)
    solid_angle[:] = 4 * np.pi * np.sin(theta[0] / 2) ** 2

    # bundle_generate: emissivity / temperature of a bundle (constants, or an upper bound)
    emissivity = np.ones([1], dtype=np.float64)
    temperature = np.ones([1], dtype=np.float64)
    if model['geometry'] == 'toroidal':
        if model['emissivity_profile'] is None:
            emissivity[:] = model['emissivity'] * model['emissivity_scale']
        else:
            emissivity[:] = np.max(model['emissivity_profile'][1]) * model['emissivity_scale']
        if model['temperature_profile'] is None:
            temperature[:] = model['temperature'] * model['temperature_scale']
        if not np.all(np.isfinite(temperature)):
            raise ValueError('No rays generated. Check plasma input parameters')
    else:
        emissivity[:] = model['emissivity']
        temperature[:] = model['temperature']
    intensity = (emissivity[0] * p['time_resolution'] * p['bundle_volume'] * solid_angle[0] / (4 * np.pi))
    intensity *= p['volume'] / (p['bundle_count'] * p['bundle_volume'])
    predicted_rays = int(B * (emissivity[0] * p['time_resolution'] * p['bundle_volume'] * solid_angle[0] / (4 * np.pi)
                              * p['volume'] / (p['bundle_count'] * p['bundle_volume'])))
    if p['max_rays'] and predicted_rays > p['max_rays']:
        if not per_bundle:
            raise ValueError(f"Current settings will produce too many rays ({predicted_rays:0.2e}). "
                             f"Please reduce integration time or adjust other parameters.")
        predicted_rays = int(p['max_rays'])      # an upper bound only; the device flags an overflow
    if p['use_poisson']:
        capacity = int(predicted_rays + 12 * np.sqrt(max(predicted_rays, 1)) + 4096)
    else:
        if intensity < 1 and not per_bundle:
            raise ValueError('intensity of less than one encountered. Turn on poisson statistics.')
        capacity = max(B * int(intensity), 1)
    q = {'xsize': voxel_size, 'ysize': voxel_size, 'zsize': voxel_size, 'spatial_dist': 'uniform',
         'angular_dist': ad, 'spread': spread_b, 'intensity': capacity, 'zaxis': p['zaxis'], 'xaxis': p['xaxis'],
         'wavelength_dist': p['wavelength_dist'], 'wavelength': p['wavelength'],
         'wavelength_range': p['wavelength_range'], 'linewidth': p['linewidth'], 'mass_number': p['mass_number'],
         'temperature': temperature[0], 'velocity': np.asarray(model['velocity'], dtype=np.float64),
         'target': p['target']}
    if not per_bundle:
        return q, B, float(intensity), None

    pl = Plasma()
    pl.geometry = PLASMA_GEOMETRY[model['geometry']]
    pl.has_spread_radius = int(p['spread_radius'] is not None)
    pl.spread_radius = float(p['spread_radius']) if p['spread_radius'] is not None else 0.0
    pl.solid_angle = float(solid_angle[0])
    _vec(pl.torus_origin, model.get('torus_origin', np.zeros(3)))
    pl.major_radius = float(model.get('major_radius', 0.0))
    pl.minor_radius = float(model.get('minor_radius', 0.0))
    pl.emissivity = float(model['emissivity'])
    pl.emissivity_scale = float(model['emissivity_scale'])
    pl.temperature_scale = float(model['temperature_scale'])
    if model['geometry'] == 'toroidal' and model['emissivity_profile'] is None:
        pl.emissivity = float(model['emissivity'])
    for which in ('emissivity', 'temperature'):
        prof = model[which + '_profile']
        n = 0
        if prof is not None:
            x = np.ascontiguousarray(prof[0], dtype=np.float64)
            y = np.ascontiguousarray(prof[1], dtype=np.float64)
            keep.extend([x, y])
            n = len(x)
            setattr(pl, which + '_rho', x.ctypes.data_as(C.POINTER(C.c_double)))
            setattr(pl, which + '_val', y.ctypes.data_as(C.POINTER(C.c_double)))
        setattr(pl, 'n_' + which, n)
    pl.voigt_gamma = 0.0
    pl.n_weideman = 0
    if pl.n_temperature > 0:
        wtype = str.lower(p['wavelength_dist'])
        if wtype == 'voigt' and p['linewidth'] != 0.0:
            # one Voigt profile per bundle (its own temperature, the shared natural width): the device builds
            # each bundle's CDF table itself; gamma as in random_wavelength_voigt (_XicsrtSourceGeneric.py:346)
            c = const.physical_constants['speed of light in vacuum'][0]
            pl.voigt_gamma = float(p['linewidth'] * p['wavelength'] ** 2 / (4 * np.pi * c * 1e10))
            big_l, coeff = weideman_coefficients(40)
            keep.append(coeff)
            pl.weideman_L = float(big_l)
            pl.weideman_a = coeff.ctypes.data_as(C.POINTER(C.c_double))
            pl.n_weideman = len(coeff)
        q['temperature'] = 1.0          # placeholder: selects the Gaussian / Voigt case, sigma comes per bundle
    pl.time_resolution = float(p['time_resolution'])
    pl.bundle_volume = float(p['bundle_volume'])
    pl.four_pi = float(4 * np.pi)
    pl.volume_ratio = float(p['volume'] / (p['bundle_count'] * p['bundle_volume']))
    pl.mass_number = float(p['mass_number'])
    pl.amu_kg = float(const.physical_constants['atomic mass unit-kilogram relationship'][0])
    pl.ev_J = float(const.physical_constants['electron volt-joule relationship'][0])
    pl.c_squared = float(const.physical_constants['speed of light in vacuum'][0] ** 2)
    pl.n_filters = len(filters)
    for i, f in enumerate(filters):
        origin, zaxis, radius = f.sightline()
        _vec(pl.filters[i].origin, origin)
        _vec(pl.filters[i].zaxis, zaxis)
        pl.filters[i].radius = float(radius)
    keep.append(pl)
    return q, B, float(intensity), pl


class ExternalRays:
    """Stands in for the source when a caller's own ray array is traced (TraceObject.trace_global(rays)):
    `rays` [8, n] float64 and `mask` [n] uint8 are device tensors kept alive by this object."""

    cone_axis_rule = 'external'

    def __init__(self, rays, mask):
        self.rays, self.mask = rays, mask
        self.n = int(mask.shape[0])

    @staticmethod
    def address(x):
        return x.data_ptr() if hasattr(x, 'data_ptr') else x.ctypes.data


def flatten_source(obj, out, keep):
    """Fill a Source struct from an initialised XicsrtSource* object; `keep` pins host arrays."""
    out.ext_rays = None
    out.ext_mask = None
    if obj.cone_axis_rule == 'external':
        C.memset(C.byref(out), 0, C.sizeof(out))
        out.kind = SRC_KIND['external']
        out.intensity = obj.n
        _vec(out.orientation, np.eye(3))
        out.two_pi = float(2 * np.pi)
        out.ext_rays = obj.address(obj.rays)
        out.ext_mask = obj.address(obj.mask)
        keep.append(obj)
        return
    p = obj.param
    out.bundle_count = 0
    _vec(out.plasma_size, np.zeros(3))
    out.bundle_intensity = 0.0
    out.use_poisson = 0
    out.plasma = None
    out.n_ray_filters = 0
    if obj.cone_axis_rule != 'plasma':
        # XicsrtSourceGeneric.ray_filter: the attached filters act on the generated rays (their origins)
        filters = [f for f in getattr(obj, 'filter_objects', []) if f.filter_kind != 'none']
        if len(filters) > XRT_MAX_BUNDLE_FILTERS:
            raise SceneError('more than %d filters on one source' % XRT_MAX_BUNDLE_FILTERS)
        out.n_ray_filters = len(filters)
        for i, f in enumerate(filters):
            origin, zaxis, radius = f.sightline()
            _vec(out.ray_filters[i].origin, origin)
            _vec(out.ray_filters[i].zaxis, zaxis)
            out.ray_filters[i].radius = float(radius)
    if obj.cone_axis_rule == 'plasma':
        q, B, lam, pl = plasma_as_source_param(obj, keep)
        if pl is not None:
            out.plasma = C.pointer(pl)
        out.bundle_count = B
        _vec(out.plasma_size, [p['xsize'], p['ysize'], p['zsize']])
        out.bundle_intensity = lam
        out.use_poisson = int(bool(p['use_poisson']))
        p = q
    out.kind = SRC_KIND[obj.cone_axis_rule]
    sd = str(p['spatial_dist'])
    if sd not in SPATIAL:
        raise NotImplementedError(f"spatial_dist: {p['spatial_dist']} not implemented.")
    out.spatial_dist = SPATIAL[sd]
    ad = 'isotropic' if p['angular_dist'] is None else str(p['angular_dist']).lower()
    if ad == 'gaussian':
        # the reference raises NameError here (xicsrt_spread.py:55)
        raise NameError("name 'vector_dist_gaussian' is not defined")
    if ad not in ANGULAR:
        raise Exception(f'Distribution "{ad}" is not known.')
    out.angular_dist = ANGULAR[ad]
    out.intensity = int(p['intensity'])
    _vec(out.origin, obj.origin)
    _vec(out.orientation, obj.orientation)
    _vec(out.size, [p['xsize'], p['ysize'], p['zsize']])
    _vec(out.spatial_A, np.zeros(9))
    if sd == 'gaussian':
        cov = [[p['xsize'] ** 2, 0, 0], [0, p['ysize'] ** 2, 0], [0, 0, p['zsize'] ** 2]]
        sigma_to_fwhm = 2 * np.sqrt(2 * np.log(2))
        cov = np.array(cov) / sigma_to_fwhm ** 2
        # legacy multivariate_normal: x = standard_normal @ (sqrt(s)[:, None] * v)
        (u, s, v) = np.linalg.svd(cov.astype(np.double))
        _vec(out.spatial_A, np.sqrt(s)[:, None] * v)
    axis = p['target'] if obj.cone_axis_rule == 'plasma' else obj.cone_axis()
    if axis is None:
        raise Exception('source cone axis (%s) is not set' % obj.cone_axis_rule)
    _vec(out.axis, axis)
    _vec(out.basis, np.zeros(9))
    if obj.cone_axis_rule not in ('target', 'plasma'):
        # make_normal + random_direction frame for a single (shared) cone axis
        array = np.empty((1, 3))
        array[:] = axis
        normal = array / np.linalg.norm(array, axis=1)[:, np.newaxis]
        o_1 = np.cross(normal, p['xaxis']) + np.cross(normal, p['zaxis'])
        o_1 /= np.linalg.norm(o_1, axis=1)[:, np.newaxis]
        o_2 = np.cross(normal, o_1)
        o_2 /= np.linalg.norm(o_2, axis=1)[:, np.newaxis]
        _vec(out.basis, np.concatenate([o_2[0], o_1[0], normal[0]]))

    spread = p['spread']
    ang = np.zeros(5)
    if ad == 'isotropic':
        theta = _parse_spread_single(spread)
        ang[0] = np.cos(theta)[0]
    elif ad == 'flat':
        theta = _parse_spread_single(spread)
        ang[0] = np.tan(theta)[0]
    elif ad == 'flat_xy':
        ang[0:4] = np.tan(_parse_spread_xy(spread))
    elif ad == 'isotropic_xy':
        theta = _parse_spread_xy(spread)
        theta_xmax = np.max(np.abs(theta[0:2]))
        theta_ymax = np.max(np.abs(theta[2:]))
        theta_max = np.arcsin(np.sqrt(np.sin(theta_xmax) ** 2 + np.sin(theta_ymax) ** 2))
        ang[0] = np.cos(_parse_spread_single(theta_max))[0]
        ang[1:5] = [np.sin(t) for t in theta]
    _vec(out.ang, ang)
    out.two_pi = float(2 * np.pi)

    # wavelength: the reference's case analysis (_XicsrtSourceGeneric.py:295-367)
    import scipy.constants as const
    c = const.physical_constants['speed of light in vacuum'][0]
    amu_kg = const.physical_constants['atomic mass unit-kilogram relationship'][0]
    ev_J = const.physical_constants['electron volt-joule relationship'][0]
    out.wavelength = float(p['wavelength'])
    out.wl_a = out.wl_b = 0.0
    out.voigt_n = 0
    wtype = str.lower(p['wavelength_dist'])
    if wtype == 'monochrome':
        out.wavelength_dist = WL_CONST
    elif wtype == 'uniform':
        out.wavelength_dist = WL_UNIFORM
        low, high = float(p['wavelength_range'][0]), float(p['wavelength_range'][1])
        out.wl_a, out.wl_b = low, high - low
    elif wtype == 'voigt':
        if p['linewidth'] == 0.0 and p['temperature'] == 0.0:
            out.wavelength_dist = WL_CONST
        elif p['linewidth'] == 0.0:
            out.wavelength_dist = WL_NORMAL
            out.wl_a = float(np.sqrt(p['temperature'] / p['mass_number'] / amu_kg / c ** 2 * ev_J)
                             * p['wavelength'])
        else:
            if p['temperature'] == 0.0:
                p['temperature'] += 1.0     # reference quirk (_XicsrtSourceGeneric.py:339)
            gamma = (p['linewidth'] * p['wavelength'] ** 2 / (4 * np.pi * c * 1e10))
            sigma = (np.sqrt(p['temperature'] / p['mass_number'] / amu_kg / c ** 2 * ev_J)
                     * p['wavelength'])
            cdf_x, cdf = voigt_cdf_table(gamma, sigma)
            cdf_x = np.ascontiguousarray(cdf_x, dtype=np.float64)
            cdf = np.ascontiguousarray(cdf, dtype=np.float64)
            keep.extend([cdf_x, cdf])
            out.wavelength_dist = WL_VOIGT
            lo, hi = float(np.min(cdf)), float(np.max(cdf))
            out.wl_a, out.wl_b = lo, hi - lo
            out.voigt_n = len(cdf)
            out.voigt_cdf = cdf.ctypes.data_as(C.POINTER(C.c_double))
            out.voigt_x = cdf_x.ctypes.data_as(C.POINTER(C.c_double))
    else:
        raise Exception(f'Wavelength distribution {wtype} unknown')
    velocity = np.asarray(p['velocity'], dtype=np.float64)
    out.has_velocity = int(not np.all(velocity == 0.0))
    _vec(out.velocity, velocity)
    out.light_speed = float(c)


def _aperture_list(info):
    if info is None:
        return []
    info = np.asarray(info)
    if info.ndim == 0:
        info = info.reshape(1)
    return list(info)


def flatten_mesh(obj, keep):
    """ShapeMesh tables -> Mesh struct (host pointers into arrays pinned in `keep`)."""
    p = obj.param
    fine = p['mesh']
    m = Mesh()

    def ptr(arr, dtype, ctype):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        keep.append(arr)
        return arr.ctypes.data_as(C.POINTER(ctype))

    m.n_points, m.n_faces = len(fine['points']), len(fine['faces'])
    m.interpolate = int(bool(p['mesh_interpolate']))
    m.points = ptr(fine['points'], np.float64, C.c_double)
    for key in ('p0', 'p1', 'p2', 'edge1', 'edge2', 'faces_normal', 'faces_area'):
        setattr(m, key, ptr(fine[key], np.float64, C.c_double))
    m.p_faces_idx = ptr(fine['p_faces_idx'], np.int32, C.c_int32)
    m.p_faces_mask = ptr(fine['p_faces_mask'], np.uint8, C.c_uint8)
    m.n_coarse_faces = 0
    if p['mesh_refine']:
        coarse = p['mesh_coarse']
        m.n_coarse_faces = len(coarse['faces'])
        m.c_p0 = ptr(coarse['p0'], np.float64, C.c_double)
        m.c_edge1 = ptr(coarse['edge1'], np.float64, C.c_double)
        m.c_edge2 = ptr(coarse['edge2'], np.float64, C.c_double)
    m.n_simplices = 0
    if m.interpolate:
        m.n_simplices = len(fine['ct_simplices'])
        m.ct_simplices = ptr(fine['ct_simplices'], np.int32, C.c_int32)
        m.ct_neighbors = ptr(fine['ct_neighbors'], np.int32, C.c_int32)
        m.ct_transform = ptr(fine['ct_transform'], np.float64, C.c_double)
        m.ct_points = ptr(fine['ct_points'], np.float64, C.c_double)
        m.ct_values = ptr(fine['ct_values'], np.float64, C.c_double)
        m.ct_grad = ptr(fine['ct_grad'], np.float64, C.c_double)
        m.ct_vertex_simplex = ptr(fine['ct_vertex_simplex'], np.int32, C.c_int32)
    keep.append(m)
    return m


def flatten_optic(obj, out, image_offset, keep=None):
    """Fill an Optic struct from an initialised XicsrtOptic* object; returns bins used."""
    p = obj.param
    out.mesh = None
    if obj.shape_kind == 'mesh':
        out.mesh = C.pointer(flatten_mesh(obj, keep))
    if obj.shape_kind not in SHAPE or obj.interact_kind not in INTERACT:
        raise SceneError('optic %s is not implemented on the device path' % obj.name)
    out.shape = SHAPE[obj.shape_kind]
    out.interact = INTERACT[obj.interact_kind]
    flags = 0
    if p['check_size']:
        flags |= F_CHECK_SIZE
    if p['check_aperture']:
        flags |= F_CHECK_APERTURE
    half = [0.0, 0.0, 0.0]
    for k, (key, flag) in enumerate((('xsize', F_HAS_XSIZE), ('ysize', F_HAS_YSIZE), ('zsize', F_HAS_ZSIZE))):
        if p[key] is not None:
            flags |= flag
            half[k] = p[key] / 2
    if p['trace_local']:
        flags |= F_TRACE_LOCAL
    _vec(out.origin, obj.origin)
    _vec(out.orientation, obj.orientation)
    _vec(out.half_size, half)

    out.radius = out.radius2 = 0.0
    _vec(out.center, np.zeros(3))
    out.torus_major = out.torus_minor = 0.0
    out.torus_root = 0
    _vec(out.torus_k, np.zeros(5))
    if obj.shape_kind in ('sphere', 'cylinder'):
        out.radius = float(p['radius'])
        out.radius2 = float(p['radius'] ** 2)
        _vec(out.center, p['center'])
        if p['convex']:
            flags |= F_CONVEX
    elif obj.shape_kind == 'torus':
        out.torus_major = float(p['torus_major'])
        out.torus_minor = float(p['torus_minor'])
        out.torus_root = int(p['root_idx'])
        _vec(out.center, p['center'])
        r_major, r_minor = p['torus_major'], p['torus_minor']
        r_sq = r_major ** 2 + r_minor ** 2
        _vec(out.torus_k, [r_sq, 2 * r_sq, 4 * r_major ** 2, 8 * r_major ** 2,
                           (r_major ** 2 - r_minor ** 2) ** 2])

    out.rocking_type = ROCKING_GAUSS
    out.two_d = out.rocking_half_fwhm = out.rocking_2sigma2 = 0.0
    out.reflectivity = 1.0
    out.half_pi = float(np.pi / 2)
    out.mosaic_depth = 0
    out.mosaic_has_cutoff = 0
    out.mosaic_cutoff_angle = 0.0
    _vec(out.mosaic_A, np.zeros(4))
    if obj.interact_kind == 'mosaic':
        out.mosaic_depth = int(p['mosaic_depth'])
        if p['mosaic_cutoff'] is not None:
            out.mosaic_has_cutoff = 1
            angle_sigma = p['mosaic_spread'] / (2 * np.sqrt(2 * np.log(2)))
            out.mosaic_cutoff_angle = float(np.sqrt(-1 * np.log(p['mosaic_cutoff']) * 2 * angle_sigma ** 2))
        # vector_dist_flat_gaussian(mosaic_spread / 2, n) (tools/xicsrt_spread.py:297-339)
        theta = _parse_spread_single(p['mosaic_spread'] / 2.0)
        sigma = theta[0] / (np.sqrt(2 * np.log(2)))
        xsigma = np.sin(sigma)
        ysigma = np.sin(sigma)
        cov = np.array([[xsigma ** 2, 0], [0, ysigma ** 2]])
        (u, sv, v) = np.linalg.svd(cov.astype(np.double))
        _vec(out.mosaic_A, np.sqrt(sv)[:, None] * v)
    if obj.interact_kind in ('crystal', 'mosaic'):
        # `check_bragg is False` is the reference's test (_InteractCrystal.py:120)
        if p['check_bragg'] is not False:
            flags |= F_CHECK_BRAGG
            rt = p['rocking_type']
            if 'step' in rt:
                out.rocking_type = ROCKING_STEP
                out.rocking_half_fwhm = float(p['rocking_fwhm'] / 2)
            elif 'gauss' in rt:
                out.rocking_type = ROCKING_GAUSS
                sigma = p['rocking_fwhm'] / (2 * np.sqrt(2 * np.log(2)))
                out.rocking_2sigma2 = float(2 * sigma ** 2)
            elif 'file' in rt:
                # the reference's reader raises NameError (xicsrt_bragg.py:40,85,87)
                raise NameError("name 'm_log' is not defined")
            else:
                raise Exception('Rocking curve type not understood: {}'.format(rt))
            out.two_d = float(2 * p['crystal_spacing'])
            out.reflectivity = float(p['reflectivity'])

    aps = _aperture_list(p['aperture'])
    if len(aps) > XRT_MAX_APERTURES:
        raise SceneError('more than %d apertures on one optic' % XRT_MAX_APERTURES)
    out.n_apertures = len(aps)
    for k, ap in enumerate(aps):
        a = out.apertures[k]
        shape = (ap.get('shape') or 'none').lower()
        logic = (ap.get('logic') or 'and').lower()
        if shape not in AP_SHAPE:
            raise Exception(f'Aperture shape: "{shape}" is not implemented.')
        if logic not in AP_LOGIC:
            raise Exception(f'Aperture logic "{logic}" is not known.')
        a.shape, a.logic = AP_SHAPE[shape], AP_LOGIC[logic]
        origin = ap.get('origin')
        origin = np.array([0.0, 0.0]) if origin is None else np.asarray(origin, dtype=np.float64).ravel()
        _vec(a.origin, origin[0:2])
        size = np.zeros(2)
        if 'size' in ap:
            s = np.asarray(ap['size'], dtype=np.float64).ravel()
            size[:min(2, len(s))] = s[:2]
        _vec(a.size, size)
        verts = np.zeros(6)
        if shape == 'triangle':
            v = np.asarray(ap['vertices'], dtype=np.float64)
            verts = np.concatenate([v[i, 0:2] + origin[0:2] for i in range(3)])
        _vec(a.vertices, verts)

    bins = 0
    out.pixel_size = out.pixel_xoff = out.pixel_yoff = 0.0
    out.pixel_nx = out.pixel_ny = 0
    out.image_offset = image_offset
    if p['enable_image']:
        flags |= F_IMAGE
        out.pixel_size = float(p['pixel_size'])
        out.pixel_nx, out.pixel_ny = int(p['pixel_xsize']), int(p['pixel_ysize'])
        out.pixel_xoff = float((p['pixel_xsize'] - 1) / 2)
        out.pixel_yoff = float((p['pixel_ysize'] - 1) / 2)
        bins = out.pixel_nx * out.pixel_ny
    out.flags = flags
    return bins


class FlatScene:
    """A Scene struct plus everything that must stay alive with it."""

    def __init__(self, source_obj, optic_objs, names):
        if len(optic_objs) > XRT_MAX_OPTICS:
            raise SceneError('more than %d optics' % XRT_MAX_OPTICS)
        self.struct = Scene()
        self._keep = []
        self.names = list(names)                    # source name then optic names
        self.optic_objs = list(optic_objs)
        flatten_source(source_obj, self.struct.source, self._keep)
        self.struct.n_optics = len(optic_objs)
        offset = 0
        self.image_slices = {}
        for k, (name, obj) in enumerate(zip(self.names[1:], optic_objs)):
            bins = flatten_optic(obj, self.struct.optics[k], offset, self._keep)
            if bins:
                o = self.struct.optics[k]
                self.image_slices[name] = (offset, o.pixel_nx, o.pixel_ny)
            else:
                self.image_slices[name] = None
            offset += bins
        self.struct.image_bins = offset

    @property
    def n_elements(self):
        return self.struct.n_optics + 1

    @property
    def n_rays(self):
        return int(self.struct.source.intensity)

    @property
    def image_bins(self):
        return int(self.struct.image_bins)

    def byref(self):
        return C.byref(self.struct)

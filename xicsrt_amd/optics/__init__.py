"""
Optical elements of the plug-in surface: an element class is the combination of
an *interaction* (none / mirror / Bragg crystal) and a *shape* (plane, sphere,
cylinder, torus); the class name is the config contract (`class_name`).

Host side only holds the parameters and the set-up rules of the reference:

  TraceObject      sizes, pixel grid, apertures, flags
                   (xicsrt/optics/_TraceObject.py:28-133)
  ShapePlane / ShapeSphere / ShapeCylinder / ShapeTorus
                   (xicsrt/optics/_ShapePlane.py:17, _ShapeSphere.py:18-43,
                    _ShapeCylinder.py:18-50, _ShapeTorus.py:18-91)
  InteractNone / InteractMirror / InteractCrystal
                   (xicsrt/optics/_InteractNone.py:15, _InteractMirror.py:16,
                    _InteractCrystal.py:19-94)
  XicsrtOptic*     (Interact, Shape) pairs, e.g. XicsrtOpticSphericalCrystal =
                   (InteractCrystal, ShapeSphere) (xicsrt/optics/_XicsrtOpticSphericalCrystal.py:16)

intersect / check_bounds / interact / make_image are device kernels
(xicsrt_amd/csrc/); the methods below route single-object calls to them.
"""
import numpy as np

from ..objects import GeometryObject


class TraceObject(GeometryObject):
    """Base of all optics: finite extent, pixel grid, apertures."""

    shape_kind = None       # set by Shape* mixins
    interact_kind = None    # set by Interact* mixins

    def default_config(self):
        """
        xsize, ysize, zsize : extent along the local axes (None = unbounded)
        pixel_size          : square pixel size for images (default xsize/100)
        aperture            : dict or list of dicts (shape, size, origin, logic)
        trace_local         : trace in the optic's local frame
        check_size, check_aperture : enable the bounds tests
        filters
        """
        config = super().default_config()
        config['xsize'] = None
        config['ysize'] = None
        config['zsize'] = None
        config['pixel_size'] = None
        config['trace_local'] = False
        config['check_size'] = True
        config['check_aperture'] = True
        config['aperture'] = None
        config['filters'] = []
        return config

    def initialize(self):
        super().initialize()
        p = self.param
        if p['xsize'] and p['ysize']:
            if p['pixel_size'] is None:
                p['pixel_size'] = p['xsize'] / 100
            pixel_xsize = p['xsize'] / p['pixel_size']
            pixel_ysize = p['ysize'] / p['pixel_size']
            if not (abs(pixel_xsize - np.round(pixel_xsize)) < 1.5e-7
                    and abs(pixel_ysize - np.round(pixel_ysize)) < 1.5e-7):
                self.log.warning(
                    f"Optic width ({p['xsize']:0.4f}x{p['ysize']:0.4f})"
                    f"is not a multiple of the pixel_size ({p['pixel_size']:0.4f})."
                    f"May lead to truncation of output image.")
            p['pixel_xsize'] = int(np.round(pixel_xsize))
            p['pixel_ysize'] = int(np.round(pixel_ysize))
            p['enable_image'] = True
        else:
            p['enable_image'] = False

    # ---- per-object tracing API (device-backed) ---------------------------
    def trace_global(self, rays):
        from .. import xicsrt_raytrace as _rt
        return _rt.trace_optic_object(self, rays)

    def trace(self, rays):
        """intersect -> check_bounds -> interact in the frame the rays are given in (optics/_TraceObject.py:157-172);
        meshes and mosaic crystals, whose steps are not separate device calls, go through trace_global()."""
        if self.shape_kind == 'mesh' or getattr(self, 'interact_kind', None) == 'mosaic':
            if self.param['trace_local']:
                raise NotImplementedError('trace() in local coordinates is only available through trace_global().')
            return self.trace_global(rays)
        xloc, norm, mask = self.intersect(rays)
        mask = self.check_bounds(xloc, mask)
        return self.interact(rays, xloc, norm, mask)

    def make_image(self, rays):
        from .. import xicsrt_raytrace as _rt
        return _rt.image_of_optic_object(self, rays)

    # The three steps of trace() as separate device calls (analytic shapes; see include/xicsrt_hip.h).

    def intersect(self, rays):
        """(xloc, norm, mask): intersection points and surface normals (NaN where there is none); like the
        reference's analytic shapes the returned mask IS rays['mask'], updated in place."""
        from .. import xicsrt_raytrace as _rt
        return _rt.optic_intersect(self, rays)

    def check_bounds(self, X, mask):
        """mask &= inside the size limits and the aperture list; updated in place and returned."""
        from .. import xicsrt_raytrace as _rt
        return _rt.optic_check_bounds(self, X, mask)

    def interact(self, rays, xloc, norm, mask=None):
        """Origins <- xloc, reflection / Bragg test of the rays in `mask`; a Bragg test draws
        np.random.uniform(0, 1, n_live) from the global legacy stream, as the reference does."""
        from .. import xicsrt_raytrace as _rt
        return _rt.optic_interact(self, rays, xloc, norm, mask)


# ---- shapes ---------------------------------------------------------------

class ShapeObject(TraceObject):
    pass


class ShapePlane(ShapeObject):
    shape_kind = 'plane'


class _Curved(ShapeObject):
    def default_config(self):
        """radius : radius of curvature;  convex : curvature sign."""
        config = super().default_config()
        config['radius'] = 1.0
        config['convex'] = False
        return config

    def initialize(self):
        super().initialize()
        sign = -1 if self.param['convex'] else 1
        self.param['center'] = sign * self.param['radius'] * self.param['zaxis'] + self.param['origin']


class ShapeSphere(_Curved):
    shape_kind = 'sphere'


class ShapeCylinder(_Curved):
    shape_kind = 'cylinder'


class ShapeTorus(ShapeObject):
    shape_kind = 'torus'

    def default_config(self):
        """radius_major, radius_minor : radii of curvature at the surface; convex : [major, minor]."""
        config = super().default_config()
        config['radius_major'] = 1.0
        config['radius_minor'] = 0.2
        config['convex'] = [False, False]
        return config

    def initialize(self):
        super().initialize()
        p = self.param
        r_minor, r_major = p['radius_minor'], p['radius_major']
        if r_minor >= r_major:
            raise Exception(r'Cannot construct geometry with radius_major <= radius_minor.')
        table = {(False, False): (3, r_major - r_minor, 1),
                 (False, True): (2, r_major + r_minor, 1),
                 (True, False): (1, r_major + r_minor, -1),
                 (True, True): (0, r_major - r_minor, -1)}
        try:
            key = tuple(bool(v) for v in np.asarray(p['convex']).tolist())
            p['root_idx'], p['torus_major'], sign = table[key]
        except Exception:
            raise Exception(f"Cannot be parse convex config option: {p['convex']}")
        p['torus_minor'] = r_minor
        p['center'] = p['origin'] + sign * r_major * p['zaxis']


# ---- interactions ------------------------------------------------------------

class InteractObject(TraceObject):
    interact_kind = 'none'


class InteractNone(InteractObject):
    pass


class InteractMirror(InteractObject):
    interact_kind = 'mirror'


class InteractCrystal(InteractMirror):
    interact_kind = 'crystal'

    def default_config(self):
        """
        crystal_spacing : d spacing [angstrom] (not 2d)
        reflectivity    : scales the reflection probability
        check_bragg     : False = perfect mirror
        rocking_type    : 'step' | 'gaussian' | 'file';  rocking_fwhm [rad]
        rocking_file, rocking_filetype, rocking_mix : tabulated curves
        """
        config = super().default_config()
        config['crystal_spacing'] = 0.0
        config['reflectivity'] = 1.0
        config['check_bragg'] = True
        config['rocking_type'] = 'gaussian'
        config['rocking_fwhm'] = None
        config['rocking_file'] = None
        config['rocking_filetype'] = None
        config['rocking_mix'] = 0.5
        return config

    def initialize(self):
        super().initialize()
        self.param['rocking_type'] = str.lower(self.param['rocking_type'])


class InteractMosaicCrystal(InteractCrystal):
    """Multi-layer mosaic (HOPG-like) crystal (xicsrt/optics/_InteractMosaicCrystal.py:18-139)."""
    interact_kind = 'mosaic'

    def default_config(self):
        """
        mosaic_spread : fwhm of the crystallite-normal distribution [rad]
        mosaic_depth  : number of crystallite layers modelled
        mosaic_cutoff : probability cutoff to skip rays far from the Bragg angle (None = off)
        """
        config = super().default_config()
        config['mosaic_spread'] = 0.0
        config['mosaic_depth'] = 15
        config['mosaic_cutoff'] = None
        return config


# ---- the named element classes ------------------------------------------------

_ELEMENTS = (
    ('XicsrtOpticAperture', InteractNone, ShapePlane),
    ('XicsrtOpticDetector', InteractNone, ShapePlane),
    ('XicsrtOpticPlanarMirror', InteractMirror, ShapePlane),
    ('XicsrtOpticPlanarCrystal', InteractCrystal, ShapePlane),
    ('XicsrtOpticSphericalMirror', InteractMirror, ShapeSphere),
    ('XicsrtOpticSphericalCrystal', InteractCrystal, ShapeSphere),
    ('XicsrtOpticCylindricalMirror', InteractMirror, ShapeCylinder),
    ('XicsrtOpticCylindricalCrystal', InteractCrystal, ShapeCylinder),
    ('XicsrtOpticToroidalCrystal', InteractCrystal, ShapeTorus),
    ('XicsrtOpticPlanarMosaicCrystal', InteractMosaicCrystal, ShapePlane),
    ('XicsrtOpticSphericalMosaicCrystal', InteractMosaicCrystal, ShapeSphere),
)

from .mesh import ShapeMesh, ShapeMeshSphere, ShapeMeshCylinder, ShapeMeshTorus  # noqa: E402

_ELEMENTS = _ELEMENTS + (
    ('XicsrtOpticMeshMirror', InteractMirror, ShapeMesh),
    ('XicsrtOpticMeshCrystal', InteractCrystal, ShapeMesh),
    ('XicsrtOpticMeshSphericalCrystal', InteractCrystal, ShapeMeshSphere),
    ('XicsrtOpticMeshCylindricalCrystal', InteractCrystal, ShapeMeshCylinder),
    ('XicsrtOpticMeshToroidalCrystal', InteractCrystal, ShapeMeshTorus),
    ('XicsrtOpticMeshMosaicCrystal', InteractMosaicCrystal, ShapeMesh),
)

BUILTIN = {}
for _name, _interact, _shape in _ELEMENTS:
    _cls = type(_name, (_interact, _shape), {
        '__doc__': '%s surface with %s interaction.' % (_shape.shape_kind, _interact.interact_kind),
        '__module__': __name__})
    BUILTIN[_name] = _cls
    globals()[_name] = _cls
del _name, _interact, _shape, _cls

# Known reference element classes that the device path does not implement yet.
NOT_IMPLEMENTED = ()

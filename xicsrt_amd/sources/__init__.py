"""
Ray sources of the plug-in surface: class names, config keys and set-up rules
of the reference; the per-ray sampling itself runs on the device.

  XicsrtSourceGeneric   cuboid/gaussian volume, cone about the zaxis
                        (xicsrt/sources/_XicsrtSourceGeneric.py:20-396)
  XicsrtSourceDirected  cone about an explicit `direction`
                        (xicsrt/sources/_XicsrtSourceDirected.py:16-50)
  XicsrtSourceFocused   cone aimed at `target` from every origin
                        (xicsrt/sources/_XicsrtSourceFocused.py:16-44)
  XicsrtPlasmaGeneric / Cubic / Toroidal / ToroidalDatafile
                        plasma volumes emitting from random bundles
                        (xicsrt/sources/_XicsrtPlasma*.py)
"""
import numpy as np

from ..objects import GeometryObject


class XicsrtSourceGeneric(GeometryObject):

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.filter_objects = []

    def default_config(self):
        """
        xsize, ysize, zsize : full widths of the emitting volume (fwhm if gaussian)
        spatial_dist        : 'uniform' | 'gaussian'
        angular_dist        : 'isotropic' | 'isotropic_xy' | 'flat' | 'flat_xy'
        spread              : half-angle(s) of the emission cone [rad]
        intensity           : rays per iteration
        use_poisson         : draw the ray count from Poisson(intensity)
        wavelength_dist     : 'voigt' | 'uniform' | 'monochrome'
        wavelength, wavelength_range, linewidth, mass_number, temperature, velocity
        filters             : names of ray filters
        """
        config = super().default_config()
        config['xsize'] = 0.0
        config['ysize'] = 0.0
        config['zsize'] = 0.0
        config['intensity'] = 0.0
        config['use_poisson'] = False
        config['spatial_dist'] = 'uniform'
        config['angular_dist'] = 'isotropic'
        config['spread'] = np.pi
        config['wavelength_dist'] = 'voigt'
        config['wavelength'] = 1.0
        config['mass_number'] = 1.0
        config['linewidth'] = 0.0
        config['temperature'] = 0.0
        config['velocity'] = np.array([0.0, 0.0, 0.0])
        config['wavelength_range'] = np.array([0.0, 0.0])
        config['filters'] = []
        return config

    def initialize(self):
        super().initialize()
        if self.param['use_poisson']:
            # Consumes the legacy global stream exactly as the reference does
            # (_XicsrtSourceGeneric.py:191-192); the device run loop handles
            # this itself for seeded runs, see xicsrt_amd/xicsrt_raytrace.py.
            self.param['intensity'] = np.random.poisson(self.param['intensity'])
        elif self.param['intensity'] < 1:
            raise ValueError('intensity of less than one encountered. Turn on poisson statistics.')
        self.param['intensity'] = int(self.param['intensity'])

    # which vector the emission cone is built around (see scene.flatten_source)
    cone_axis_rule = 'zaxis'

    def cone_axis(self):
        return self.param['zaxis']

    def generate_rays(self):
        """
        Rays of one iteration drawn from the *current global np.random state*,
        as the reference's method does; runs on the device.
        """
        from .. import xicsrt_raytrace as _rt
        return _rt.generate_rays_from_global_state(self)


class XicsrtSourceDirected(XicsrtSourceGeneric):

    cone_axis_rule = 'direction'

    def default_config(self):
        """direction : axis of the emission cone (defaults to the zaxis)."""
        config = super().default_config()
        config['direction'] = None
        return config

    def initialize(self):
        super().initialize()
        if self.param['direction'] is None:
            self.param['direction'] = self.param['zaxis']

    def cone_axis(self):
        return self.param['direction']


class XicsrtSourceFocused(XicsrtSourceGeneric):

    cone_axis_rule = 'target'

    def default_config(self):
        """target : point every emission cone is aimed at."""
        config = super().default_config()
        config['target'] = None
        return config

    def cone_axis(self):
        return self.param['target']


class XicsrtPlasmaGeneric(GeometryObject):
    """
    A plasma volume emitting from randomly placed bundles; every bundle is an
    XicsrtSourceFocused aimed at `target` whose ray count follows the local
    emissivity (sources/_XicsrtPlasmaGeneric.py:21-393).  Config keys and the
    bundle bookkeeping of the reference; sampling runs on the device.
    """

    cone_axis_rule = 'plasma'

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.filter_objects = []

    def default_config(self):
        """
        xsize, ysize, zsize, target, spread / spread_radius, angular_dist,
        emissivity [photons/s/m^3], temperature [eV], velocity, time_resolution [s],
        bundle_type ('voxel' | 'point'), bundle_volume, bundle_count, max_rays, max_bundles,
        use_poisson, wavelength_dist, wavelength, wavelength_range, mass_number, linewidth, filters
        """
        config = super().default_config()
        config['xsize'] = 0.0
        config['ysize'] = 0.0
        config['zsize'] = 0.0
        config['angular_dist'] = 'isotropic'
        config['spread'] = None
        config['spread_radius'] = None
        config['target'] = None
        config['use_poisson'] = False
        config['wavelength_dist'] = 'voigt'
        config['wavelength'] = 1.0
        config['wavelength_range'] = None
        config['mass_number'] = 1.0
        config['linewidth'] = 0.0
        config['emissivity'] = 0.0
        config['temperature'] = 0.0
        config['velocity'] = 0.0
        config['time_resolution'] = 1e-3
        config['bundle_type'] = 'voxel'
        config['bundle_volume'] = 1e-6
        config['bundle_count'] = None
        config['max_rays'] = int(1e7)
        config['max_bundles'] = int(1e7)
        config['filters'] = []
        return config

    def initialize(self):
        super().initialize()
        p = self.param
        if p['max_rays'] is not None:
            p['max_rays'] = int(p['max_rays'])
        p['volume'] = self.config['xsize'] * self.config['ysize'] * self.config['zsize']
        if p['bundle_count'] is None:
            p['bundle_count'] = p['volume'] / p['bundle_volume']
        p['bundle_count'] = int(np.round(p['bundle_count']))
        if p['bundle_count'] < 1:
            raise Exception(f'Bundle volume is larger than the plasma volume.')
        if p['bundle_count'] > p['max_bundles']:
            raise ValueError(
                f"Current settings will produce too many bundles ({p['bundle_count']:0.2e}). "
                f"Increase the bundle_volume, explicitly set bundle_count or increase max_bundles.")

    def cone_axis(self):
        return self.param['target']

    def generate_rays(self):
        from .. import xicsrt_raytrace as _rt
        return _rt.generate_rays_from_global_state(self)

    def bundle_model(self):
        """
        What bundle_generate() of the class assigns to every unmasked bundle, as data for the
        device: geometry of the flux coordinate ('box': none), emissivity / temperature either
        constants or (rho, value) profile tables with a scale, and the bundle velocity.
        The base class leaves the set-up defaults: emissivity 1, temperature 1, velocity 0
        (sources/_XicsrtPlasmaGeneric.py:186-190, :238-240).
        """
        return {'geometry': 'box', 'emissivity': 1.0, 'temperature': 1.0, 'velocity': np.zeros(3),
                'emissivity_scale': 1.0, 'temperature_scale': 1.0,
                'emissivity_profile': None, 'temperature_profile': None}


class XicsrtPlasmaCubic(XicsrtPlasmaGeneric):
    """A cuboid plasma with constant temperature and emissivity (sources/_XicsrtPlasmaCubic.py:16-35)."""

    def bundle_model(self):
        m = super().bundle_model()
        m['temperature'] = self.param['temperature']
        m['emissivity'] = self.param['emissivity']
        return m


class XicsrtPlasmaCylindrical(XicsrtPlasmaGeneric):
    """
    Present for the class-name surface only: the reference's class is marked broken and fails in
    bundle_generate with AttributeError('emissivity') (sources/_XicsrtPlasmaCylindrical.py:18-55,
    it reads self.emissivity, which no object defines); the same error is raised here.
    """

    def bundle_model(self):
        raise AttributeError('emissivity')


class XicsrtPlasmaToroidal(XicsrtPlasmaGeneric):
    """
    Toroidal geometry with a circular cross-section (sources/_XicsrtPlasmaToroidal.py:19-78): every
    bundle gets emissivity, temperature and velocity from its normalised radius
    rho = sqrt(r_minor^2 / minor_radius), times the *_scale factors; bundles with a non-finite
    temperature are dropped.
    """

    def default_config(self):
        """major_radius, minor_radius, torus_origin, emissivity_scale, temperature_scale, velocity_scale"""
        config = super().default_config()
        config['major_radius'] = 0.0
        config['minor_radius'] = 0.0
        config['torus_origin'] = np.array([0.0, 0.0, 0.0])
        config['emissivity_scale'] = 1.0
        config['temperature_scale'] = 1.0
        config['velocity_scale'] = 1.0
        return config

    def profile(self, which):
        """(rho, value) table of get_<which>(rho), or None when it is the constant param[which]."""
        return None

    def bundle_model(self):
        p = self.param
        m = super().bundle_model()
        m['geometry'] = 'toroidal'
        m['torus_origin'] = np.asarray(p['torus_origin'], dtype=np.float64)
        m['major_radius'] = p['major_radius']
        m['minor_radius'] = p['minor_radius']
        m['emissivity'] = p['emissivity']
        m['temperature'] = p['temperature']
        m['emissivity_scale'] = p['emissivity_scale']
        m['temperature_scale'] = p['temperature_scale']
        m['emissivity_profile'] = self.profile('emissivity')
        m['temperature_profile'] = self.profile('temperature')
        # get_velocity(rho) * velocity_scale broadcast into the (n, 3) bundle array (:72)
        vel = np.zeros((1, 3), dtype=np.float64)
        vel[:] = p['velocity'] * p['velocity_scale']
        m['velocity'] = vel[0]
        return m


class XicsrtPlasmaToroidalDatafile(XicsrtPlasmaToroidal):
    """
    Toroidal plasma whose emissivity and temperature profiles are read from two-column text files
    (rho, value), np.interp with zero outside the table (sources/_XicsrtPlasmaToroidalDatafile.py:21-45).
    `velocity_file` is a config key of the reference that it never reads.
    """

    def default_config(self):
        """emissivity_file, temperature_file, velocity_file"""
        config = super().default_config()
        config['emissivity_file'] = None
        config['temperature_file'] = None
        config['velocity_file'] = None
        return config

    def profile(self, which):
        if which not in ('emissivity', 'temperature'):
            return None
        data = np.loadtxt(self.param[which + '_file'], dtype=np.float64)
        return (np.ascontiguousarray(data[:, 0]), np.ascontiguousarray(data[:, 1]))


BUILTIN = {cls.__name__: cls for cls in (XicsrtSourceGeneric, XicsrtSourceDirected, XicsrtSourceFocused,
                                         XicsrtPlasmaGeneric, XicsrtPlasmaCubic, XicsrtPlasmaCylindrical, XicsrtPlasmaToroidal,
                                         XicsrtPlasmaToroidalDatafile)}

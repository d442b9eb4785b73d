"""
Element base classes of the plug-in surface (host side).

These mirror the reference's object lifecycle so that existing config dicts and
user code that pokes at element objects keep working:

  ConfigObject    default_config -> update_config(strict) -> check_config ->
                  param = deepcopy(config) with lists as ndarrays -> setup ->
                  check_param -> initialize      (xicsrt/objects/_ConfigObject.py:24-84)
  GeometryObject  origin + orientation rows [xaxis, zaxis x xaxis, zaxis], default
                  xaxis rule, local <-> external transforms
                                                  (xicsrt/objects/_GeometryObject.py:76-168)
  RayArray        dict of ndarrays with O/D/W/M shortcuts
                                                  (xicsrt/objects/_RayArray.py:12-96)

The objects only hold parameters; all per-ray arithmetic happens on the device
(see xicsrt_amd/scene.py for the flattening into the C ABI structs).
"""
import copy as _copy
import logging

import numpy as np

from . import config as _config


class ConfigObject:
    """Base class of everything that is built from a config dictionary."""

    def __init__(self, config=None, strict=None, initialize=None):
        if initialize is None:
            initialize = True
        self.name = self.__class__.__name__
        self.log = logging.getLogger('xicsrt').getChild(self.name)

        self.config = self.default_config()
        self.update_config(config, strict=strict)
        self.check_config()

        self.param = _copy.deepcopy(self.config)
        self.param = _config.convert_to_numpy(self.param, inplace=True)

        if initialize:
            self.setup()
            self.check_param()
            self.initialize()

    def default_config(self):
        config = dict()
        config['class_name'] = self.__class__.__name__
        # Present in every reference object config; kept so that configs saved by
        # the reference pass the strict key check here.
        config['yo_mama'] = 'Is a beautiful person and she loves you.'
        return config

    def get_config(self):
        return self.config

    def check_config(self):
        pass

    def setup(self):
        pass

    def check_param(self):
        pass

    def initialize(self):
        pass

    def update_config(self, config_new, **kwargs):
        _config.update_config(self.config, config_new, **kwargs)


# ---------------------------------------------------------------------------
# frames
# ---------------------------------------------------------------------------
# The numbers below feed the device (scene.py copies `origin` and `orientation` into the C structs), so the
# operations are the ones the reference's objects perform (xicsrt/objects/_GeometryObject.py:76-111): the
# frame's rows are x, z cross x, z; nothing is re-orthonormalised and zaxis is taken as given.

_UP = (0.0, 0.0, 1.0)


def default_xaxis(zaxis):
    """x axis of an element that names none: the horizontal direction perpendicular to zaxis."""
    sideways = np.cross(np.array(_UP), zaxis)
    if np.any(sideways != 0.0):
        return sideways / np.linalg.norm(sideways)
    return np.array([1.0, 0.0, 0.0])            # zaxis is vertical: any horizontal axis will do


def frame_rows(zaxis, xaxis):
    return np.array([xaxis, np.cross(zaxis, xaxis), zaxis])


# einsum signatures of the two rotations, for one vector and for an (n, 3) array of them
_TO_EXTERNAL = {1: 'ij,i->j', 2: 'ij,ki->kj'}
_TO_LOCAL = {1: 'ji,i->j', 2: 'ji,ki->kj'}
_AXIS_ROW = {'xaxis': 0, 'yaxis': 1, 'zaxis': 2}


class GeometryObject(ConfigObject):
    """An object with a position and an orientation in 3D."""

    def __getattr__(self, key):
        # xaxis / yaxis / zaxis are views of the frame, available once setup() has built it
        if key in _AXIS_ROW and 'orientation' in self.__dict__:
            return self.__dict__['orientation'][_AXIS_ROW[key], :]
        raise AttributeError(key)

    def default_config(self):
        config = super().default_config()
        config['origin'] = np.array([0.0, 0.0, 0.0])
        config['zaxis'] = np.array([0.0, 0.0, 1.0])
        config['xaxis'] = None
        return config

    def check_config(self):
        given = self.config['xaxis']
        if given is not None and not np.isclose(np.dot(np.array(self.config['zaxis']), np.array(given)), 0.0):
            raise ValueError('zaxis and xaxis are not orthogonal.')

    def setup(self):
        super().setup()
        p = self.param
        for key in ('origin', 'zaxis'):
            p[key] = np.array(p[key])
        p['xaxis'] = default_xaxis(p['zaxis']) if p['xaxis'] is None else np.array(p['xaxis'])
        self.origin = p['origin']
        self.orientation = frame_rows(p['zaxis'], p['xaxis'])

    def set_orientation(self, zaxis, xaxis=None):
        self.orientation = frame_rows(zaxis, default_xaxis(zaxis) if xaxis is None else xaxis)

    def get_default_xaxis(self, zaxis):
        return default_xaxis(zaxis)

    # -- transforms (host-side conveniences; they act in place unless copy=True, like the reference's) --

    @staticmethod
    def to_ndarray(vector_in):
        return vector_in if isinstance(vector_in, np.ndarray) else np.array(vector_in, dtype=np.float64)

    def to_vector_array(self, vector_in):
        v = self.to_ndarray(vector_in)
        return v if v.ndim >= 2 else v[None, :]

    def _rotated(self, vector, signatures, copy):
        v = self.to_ndarray(_copy.copy(vector) if copy else vector)
        if v.ndim not in signatures:
            raise Exception('vector.ndim must be 1 or 2')
        v[:] = np.einsum(signatures[v.ndim], self.orientation, v)
        return v

    def vector_to_external(self, vector, copy=False):
        return self._rotated(vector, _TO_EXTERNAL, copy)

    def vector_to_local(self, vector, copy=False):
        return self._rotated(vector, _TO_LOCAL, copy)

    def point_to_external(self, point_local, copy=False):
        return self.vector_to_external(point_local, copy=copy) + self.origin

    def point_to_local(self, point_external, copy=False):
        return self.vector_to_local(point_external - self.origin, copy=copy)

    def _ray_moved(self, ray, point_fn, vector_fn, copy):
        ray = _copy.deepcopy(ray) if copy else ray
        ray['origin'] = point_fn(ray['origin'])
        ray['direction'] = vector_fn(ray['direction'])
        return ray

    def ray_to_external(self, ray_local, copy=False):
        return self._ray_moved(ray_local, self.point_to_external, self.vector_to_external, copy)

    def ray_to_local(self, ray_external, copy=False):
        return self._ray_moved(ray_external, self.point_to_local, self.vector_to_local, copy)

    def aim_to_point(self, aim_point, xaxis=None):
        zaxis = aim_point - self.origin
        zaxis /= np.linalg.norm(zaxis)
        return {'zaxis': zaxis, 'xaxis': default_xaxis(zaxis) if xaxis is None else xaxis}


# ---------------------------------------------------------------------------
# ray arrays
# ---------------------------------------------------------------------------

# field -> (shape behind the ray axis, dtype, value of a missing field); `weight` rides along when present
# but is not one of the four fields the reference's RayArray knows about (xicsrt/objects/_RayArray.py:12-96)
_RAY_FIELDS = {'origin': ((3,), np.float64, None), 'direction': ((3,), np.float64, None),
               'mask': ((), bool, True), 'wavelength': ((), np.float64, 0.0)}
_RAY_ALIAS = {'O': 'origin', 'D': 'direction', 'W': 'wavelength', 'M': 'mask'}


class RayArray(dict):
    """Dictionary of per-ray ndarrays: origin (N,3), direction (N,3), wavelength (N,), mask (N,),
    also reachable as attributes (`rays.origin`, `rays.O`, ...)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if 'origin' in self and 'direction' in self:
            self.initialize()

    def initialize(self):
        """Fill in what is missing (mask: all rays on, wavelength: 0) and make every field an ndarray."""
        if 'origin' not in self or 'direction' not in self:
            raise Exception('Cannot initialize, origin and direction must be present.')
        count = np.shape(self['origin'])[0]
        for name, (tail, dtype, fill) in _RAY_FIELDS.items():
            if name not in self:
                self[name] = np.full((count,) + tail, fill, dtype=dtype)
            elif not isinstance(self[name], np.ndarray):
                self[name] = np.array(self[name])

    def __getattribute__(self, key):
        name = _RAY_ALIAS.get(key, key)
        if name in _RAY_FIELDS:
            return self[name]
        return super().__getattribute__(key)

    def __setattr__(self, key, value):
        name = _RAY_ALIAS.get(key, key)
        if name in _RAY_FIELDS:
            self[name] = value
        else:
            super().__setattr__(key, value)

    def zeros(self, num):
        """`num` rays, all fields zero (mask off)."""
        for name, (tail, dtype, _) in _RAY_FIELDS.items():
            self[name] = np.zeros((num,) + tail, dtype=dtype)

    def copy(self):
        return RayArray({name: values.copy() for name, values in self.items()})

    def extend(self, ray_in):
        """Append the rays of `ray_in` (fields this array does not have are ignored)."""
        for name in list(self):
            self[name] = np.concatenate((self[name], ray_in[name]))

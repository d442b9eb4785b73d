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


class GeometryObject(ConfigObject):
    """An object with a position and an orientation in 3D."""

    def __getattr__(self, key):
        if key in ('xaxis', 'yaxis', 'zaxis'):
            try:
                orientation = self.__dict__['orientation']
            except KeyError:
                raise AttributeError(key)
            return orientation['xyz'.index(key[0]), :]
        raise AttributeError(key)

    def default_config(self):
        config = super().default_config()
        config['origin'] = np.array([0.0, 0.0, 0.0])
        config['zaxis'] = np.array([0.0, 0.0, 1.0])
        config['xaxis'] = None
        return config

    def check_config(self):
        if self.config['xaxis'] is not None:
            zaxis = np.array(self.config['zaxis'])
            xaxis = np.array(self.config['xaxis'])
            if not np.isclose(np.dot(zaxis, xaxis), 0.0):
                raise ValueError('zaxis and xaxis are not orthogonal.')

    def setup(self):
        super().setup()
        self.param['origin'] = np.array(self.param['origin'])
        self.param['zaxis'] = np.array(self.param['zaxis'])
        if self.param['xaxis'] is None:
            self.param['xaxis'] = self.get_default_xaxis(self.param['zaxis'])
        else:
            self.param['xaxis'] = np.array(self.param['xaxis'])
        self.origin = self.param['origin']
        self.set_orientation(self.param['zaxis'], self.param['xaxis'])

    def set_orientation(self, zaxis, xaxis=None):
        if xaxis is None:
            xaxis = self.get_default_xaxis(zaxis)
        self.orientation = np.array([xaxis, np.cross(zaxis, xaxis), zaxis])

    def get_default_xaxis(self, zaxis):
        xaxis = np.cross(np.array([0.0, 0.0, 1.0]), zaxis)
        if not np.all(xaxis == 0.0):
            xaxis /= np.linalg.norm(xaxis)
        else:
            xaxis = np.array([1.0, 0.0, 0.0])
        return xaxis

    # -- transforms (host-side conveniences; act in place like the reference) --

    @staticmethod
    def to_ndarray(vector_in):
        if not isinstance(vector_in, np.ndarray):
            vector_in = np.array(vector_in, dtype=np.float64)
        return vector_in

    def to_vector_array(self, vector_in):
        vector_in = self.to_ndarray(vector_in)
        return vector_in[None, :] if vector_in.ndim < 2 else vector_in

    def _rotate(self, vector, spec2, spec1, copy):
        if copy:
            vector = _copy.copy(vector)
        vector = self.to_ndarray(vector)
        if vector.ndim == 2:
            vector[:] = np.einsum(spec2, self.orientation, vector)
        elif vector.ndim == 1:
            vector[:] = np.einsum(spec1, self.orientation, vector)
        else:
            raise Exception('vector.ndim must be 1 or 2')
        return vector

    def vector_to_external(self, vector, copy=False):
        return self._rotate(vector, 'ij,ki->kj', 'ij,i->j', copy)

    def vector_to_local(self, vector, copy=False):
        return self._rotate(vector, 'ji,ki->kj', 'ji,i->j', copy)

    def point_to_external(self, point_local, copy=False):
        return self.vector_to_external(point_local, copy=copy) + self.origin

    def point_to_local(self, point_external, copy=False):
        return self.vector_to_local(point_external - self.origin, copy=copy)

    def ray_to_external(self, ray_local, copy=False):
        ray = _copy.deepcopy(ray_local) if copy else ray_local
        ray['origin'] = self.point_to_external(ray['origin'])
        ray['direction'] = self.vector_to_external(ray['direction'])
        return ray

    def ray_to_local(self, ray_external, copy=False):
        ray = _copy.deepcopy(ray_external) if copy else ray_external
        ray['origin'] = self.point_to_local(ray['origin'])
        ray['direction'] = self.vector_to_local(ray['direction'])
        return ray

    def aim_to_point(self, aim_point, xaxis=None):
        zaxis = aim_point - self.origin
        zaxis /= np.linalg.norm(zaxis)
        if xaxis is None:
            xaxis = self.get_default_xaxis(zaxis)
        return {'zaxis': zaxis, 'xaxis': xaxis}


_SHORT = {'O': 'origin', 'D': 'direction', 'W': 'wavelength', 'M': 'mask'}


class RayArray(dict):
    """Dictionary of per-ray ndarrays: origin (N,3), direction (N,3), wavelength (N,), mask (N,)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if 'origin' in self and 'direction' in self:
            self.initialize()

    def initialize(self):
        if not (('origin' in self) and ('direction' in self)):
            raise Exception('Cannot initialize, origin and direction must be present.')
        for key in ('origin', 'direction'):
            if not isinstance(self[key], np.ndarray):
                self[key] = np.array(self[key])
        num = self['origin'].shape[0]
        if 'mask' not in self:
            self['mask'] = np.ones(num, dtype=bool)
        if 'wavelength' not in self:
            self['wavelength'] = np.zeros(num)
        for key in ('mask', 'wavelength'):
            if not isinstance(self[key], np.ndarray):
                self[key] = np.array(self[key])

    def __getattribute__(self, key):
        full = _SHORT.get(key, key)
        if full in ('origin', 'direction', 'wavelength', 'mask'):
            return self[full]
        return super().__getattribute__(key)

    def __setattr__(self, key, value):
        full = _SHORT.get(key, key)
        if full in ('origin', 'direction', 'wavelength', 'mask'):
            self[full] = value
        else:
            super().__setattr__(key, value)

    def zeros(self, num):
        self['origin'] = np.zeros((num, 3))
        self['direction'] = np.zeros((num, 3))
        self['mask'] = np.zeros((num), dtype=bool)
        self['wavelength'] = np.zeros((num))

    def copy(self):
        new = RayArray()
        for key in self:
            new[key] = self[key].copy()
        return new

    def extend(self, ray_in):
        for key in self:
            self[key] = np.concatenate((self[key], ray_in[key]))

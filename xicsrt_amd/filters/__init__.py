"""
Bundle filters of the plug-in surface (reference: xicsrt/filters/).

  XicsrtBundleFilter           base class, keeps every bundle
                               (filters/_XicsrtBundleFilter.py:14-29)
  XicsrtBundleFilterSightline  keeps the plasma bundles within `radius` of the
                               line through `origin` along `zaxis`
                               (filters/_XicsrtBundleFilterSightline.py:13-56)

The objects only carry configuration; the per-bundle test itself runs on the
device inside the plasma source stage (see scene.flatten_plasma).
"""
import numpy as np

from ..objects import GeometryObject


class XicsrtBundleFilter(GeometryObject):
    """A filter that keeps every bundle."""

    filter_kind = 'none'

    def default_config(self):
        config = super().default_config()
        return config


class XicsrtBundleFilterSightline(XicsrtBundleFilter):
    """Keep bundles whose centre lies inside a cylinder of `radius` about the sightline."""

    filter_kind = 'sightline'

    def default_config(self):
        """radius : radius of the cylindrical sightline [m]"""
        config = super().default_config()
        config['radius'] = None
        return config

    def sightline(self):
        """(origin, zaxis, radius) exactly as the reference's test reads them: from `config`, un-normalised."""
        return (np.asarray(self.config['origin'], dtype=np.float64),
                np.asarray(self.config['zaxis'], dtype=np.float64), self.config['radius'])


BUILTIN = {cls.__name__: cls for cls in (XicsrtBundleFilter, XicsrtBundleFilterSightline)}

"""
xicsrt_amd -- MI355X-native photon propagation behind the XICSRT API.

    import xicsrt_amd as xicsrt
    results = xicsrt.raytrace(config)

Same entry points, config schema, element class names and result dictionary
as PrincetonUniversity/xicsrt 0.8 (xicsrt/__init__.py:8-16); the per-photon
work runs in hand-written HIP kernels for gfx950 (xicsrt_amd/csrc) reached
through the C ABI of include/xicsrt_hip.h.
"""
import logging as _logging

from .config import __version__
from .xicsrt_raytrace import raytrace, raytrace_mp, raytrace_single, combine_raytrace  # noqa: F401
from . import config as xicsrt_config  # noqa: F401
from . import xicsrt_io  # noqa: F401


def get_element(config_user, name, section=None, initialize=True):
    """Build one element object for interactive use (xicsrt/xicsrt_public.py:13-28)."""
    from . import xicsrt_raytrace as _rt
    config = xicsrt_config.get_config(config_user)
    if section is None:
        found = [s for s in ('optics', 'sources', 'filters') if name in config[s]]
        if len(found) == 0:
            raise Exception(f'Could not find element: {name} in any section.')
        if len(found) > 1:
            raise Warning(f'Element name: {name} was found in more than one section.'
                          f' Please provide an explicit section name.')
        section = found[0]
    general = config['general']
    cls = _rt.find_class(config[section][name]['class_name'], section, general.get('pathlist', []))
    obj = cls(config[section][name], initialize=False, strict=general['strict_config_check'])
    if initialize:
        obj.setup()
        obj.check_param()
        obj.initialize()
    return obj


def warn_version(v_string):
    """Warn when a config was written for another major.minor (xicsrt/util/version.py:15-23)."""
    from packaging import version
    log = _logging.getLogger('xicsrt')
    v_in, v_cur = version.parse(v_string), version.parse(__version__)
    if (v_in.major, v_in.minor) < (v_cur.major, v_cur.minor):
        log.warning('This config is for an older version of xicsrt. Some options may have changed.')
    elif (v_in.major, v_in.minor) > (v_cur.major, v_cur.minor):
        log.warning('This config is for a newer version of xicsrt. Please upgrade.')

"""
Configuration dictionaries: defaults of the ``general`` section, recursive
merge with strict key checking, list -> ndarray conversion.

Host-side mirror of the reference's config semantics (behaviour, not code):
  * sections and ``general`` keys/defaults      xicsrt/xicsrt_config.py:169-205
  * get_config = defaults updated non-strictly, unknown keys kept   :208-211
  * recursive merge / strict unknown-key error   :294-364
  * numeric lists become ndarrays (string and empty lists do not)
                                                 xicsrt/tools/xicsrt_misc.py:18-51
"""
import os
import numpy as np

__version__ = '0.8.13'      # API/config level this package is a drop-in for

_GENERAL_DEFAULTS = (
    ('version', __version__),
    ('number_of_iter', 1),
    ('number_of_runs', 1),
    ('random_seed', None),
    ('pathlist', None),             # filled per call (fresh lists)
    ('pathlist_default', None),
    ('strict_config_check', True),
    ('output_path', None),
    ('output_prefix', 'xicsrt'),
    ('output_suffix', None),
    ('output_run_suffix', None),
    ('image_ext', '.tif'),
    ('results_ext', '.hdf5'),
    ('config_ext', '.json'),
    ('make_directories', False),
    ('keep_meta', True),
    ('keep_images', True),
    ('keep_history', True),
    ('history_max_lost', 10000),
    ('save_config', False),
    ('save_images', False),
    ('save_results', False),
    ('print_results', True),
)


def get_pathlist_default():
    """Directories holding the built-in element classes (filters, sources, optics)."""
    here = os.path.dirname(os.path.abspath(__file__))
    return [os.path.join(here, name) for name in ('filters', 'sources', 'optics')]


def default_config():
    general = dict(_GENERAL_DEFAULTS)
    general['pathlist'] = []
    general['pathlist_default'] = get_pathlist_default()
    return {'general': general, 'sources': {}, 'optics': {}, 'filters': {}, 'scenario': {}}


def update_config(config, config_new, strict=None, update=None, ignore_none=None):
    """
    Overwrite entries of `config` with those of `config_new`, descending into
    nested dicts.  strict (default True): unknown keys raise; update: unknown
    keys are added (only meaningful when not strict); ignore_none: None values
    in `config_new` do not overwrite.
    """
    strict = True if strict is None else strict
    update = False if update is None else update
    ignore_none = False if ignore_none is None else ignore_none
    if config_new is None:
        return config
    for key, value in config_new.items():
        if key not in config:
            if strict:
                raise Exception("User option not recognized: {}".format(key))
            if update:
                config[key] = value
        elif isinstance(config[key], dict) and isinstance(value, dict):
            update_config(config[key], value, strict=strict, update=update, ignore_none=ignore_none)
        elif not (ignore_none and value is None):
            config[key] = value
    return config


def get_config(config_user=None):
    config = default_config()
    update_config(config, config_user, strict=False, update=True)
    return config


def convert_to_numpy(obj, inplace=False):
    """Numeric lists inside a dict/list become ndarrays; unicode and empty lists stay."""
    if not inplace:
        obj = obj.copy()
    if isinstance(obj, dict):
        keys = list(obj.keys())
    elif isinstance(obj, list):
        keys = range(len(obj))
    else:
        raise TypeError('Object must be either a dict or a list.')
    for key in keys:
        value = obj[key]
        if isinstance(value, list):
            if value:
                arr = np.array(value)
                if arr.dtype.char == 'U':
                    pass
                elif arr.dtype.char == 'O':
                    obj[key] = convert_to_numpy(value)
                else:
                    obj[key] = arr
        elif isinstance(value, dict):
            obj[key] = convert_to_numpy(value)
    return obj


def convert_from_numpy(obj, inplace=False):
    if not inplace:
        obj = obj.copy()
    keys = list(obj.keys()) if isinstance(obj, dict) else range(len(obj))
    for key in keys:
        value = obj[key]
        if isinstance(value, np.ndarray):
            obj[key] = value.tolist()
        elif isinstance(value, (dict, list)):
            obj[key] = convert_from_numpy(value)
    return obj


def config_to_numpy(obj):
    # As in the reference (xicsrt_config.py:284-286) the conversion result is
    # discarded: element objects convert their own `param` copy instead.
    convert_to_numpy(obj)
    return obj


def config_from_numpy(obj):
    convert_from_numpy(obj, inplace=True)
    return obj

"""
Files of the drop-in boundary: config / results dictionaries and detector images
(reference: xicsrt/xicsrt_io.py:27-222, tools/xicsrt_misc.py:18-90).

  load_config / save_config      json (numeric lists <-> ndarrays) or pickle
  save_results / load_results    the raytrace() output dictionary, json or pickle
  save_images                    one image file per optic that made an image (PIL, rot90 as the reference)
  generate_filename              '<prefix>_<name>_<suffix>_<run_suffix><ext>' under output_path

hdf5 (the reference's default `results_ext`) is written and read in the on-disk layout of the
reference's util/mirhdf5 (xicsrt/util/mirhdf5.py:58-330: one group per dict with its key order as an
attribute, lists as groups of '0000', '0001', ..., None and str marked by attributes) through h5py,
which is imported on demand.  Without h5py, hdf5 targets are refused up front by `require_writable`
(called from check_config before anything is traced), not after the work is done.
"""
import copy
import json
import logging
import os
import pathlib
import pickle

import numpy as np

from . import config as xconfig

log = logging.getLogger(__name__)


def lists_to_numpy(obj):
    """Numeric lists of a (nested) dict/list become ndarrays; empty and string lists stay lists."""
    if isinstance(obj, dict):
        keys = list(obj.keys())
    elif isinstance(obj, list):
        keys = range(len(obj))
    else:
        raise TypeError('Object must be either a dict or a list.')
    out = obj.copy()
    for k in keys:
        v = out[k]
        if isinstance(v, dict):
            out[k] = lists_to_numpy(v)
        elif isinstance(v, list) and v:
            arr = np.array(v)
            if arr.dtype.char == 'U':
                continue
            out[k] = lists_to_numpy(v) if arr.dtype.char == 'O' else arr
    return out


def numpy_to_lists(obj):
    """Inverse of lists_to_numpy for json: ndarrays -> lists, numpy scalars -> python scalars."""
    if isinstance(obj, dict):
        return {k: numpy_to_lists(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [numpy_to_lists(v) for v in obj]
    if isinstance(obj, np.ndarray):
        return numpy_to_lists(obj.tolist())
    if isinstance(obj, np.generic):
        return obj.item()
    return obj


def _suffix(filename):
    return pathlib.Path(filename).suffix


def _make_parent(filename):
    path = pathlib.Path(filename).expanduser()
    if path.suffix:
        path = path.parent
    if not path.exists():
        path.mkdir(parents=True, exist_ok=True)
        log.info(f'Made directory: {path}')


# ---------------------------------------------------------------------------
# hdf5, mirhdf5 layout
# ---------------------------------------------------------------------------

_H5_TYPE = '_mirhdf5 python object type'
_H5_ORDER = '_mirhdf5 dictionary order'
_H5_NONE = '_mirhdf5 python None'
_H5_STR = '_mirhdf5 python str'


def _h5py():
    try:
        import h5py
    except ImportError:
        raise ImportError("hdf5 files need the h5py package, which is not installed; set general.results_ext "
                          "(or config_ext) to '.json' or '.pickle'") from None
    return h5py


def is_hdf5(ext):
    return 'hdf5' in ext or 'h5' in ext


def require_writable(ext):
    """Raise now (before any tracing) when files with extension `ext` cannot be written here."""
    if is_hdf5(ext):
        _h5py()
    elif not ('pickle' in ext or 'pkl' in ext or 'json' in ext):
        raise NotImplementedError(f'filetype: {ext} not currently supported.')


def _h5_put(group, key, item):
    if isinstance(item, dict):
        sub = group.create_group(key)
        sub.attrs[_H5_TYPE] = b'dict'
        try:
            sub.attrs[_H5_ORDER] = [k.encode() for k in item]
        except (TypeError, AttributeError):
            log.error('Could not save dictionary key order. keys of unsupported data type.')
        for k, v in item.items():
            _h5_put(sub, k, v)
    elif isinstance(item, list):
        sub = group.create_group(key)
        sub.attrs[_H5_TYPE] = b'list'
        for i, v in enumerate(item):
            _h5_put(sub, '{:04d}'.format(i), v)
    elif item is None:
        group[key] = False
        group[key].attrs[_H5_NONE] = True
    else:
        try:
            group.create_dataset(key, data=item)
            if isinstance(item, str):
                group[key].attrs[_H5_STR] = True
        except TypeError:
            log.exception('Could not add key "{}" of type {} to hdf5 file.'.format(key, type(item)))


def dict_to_hdf5(data, filename):
    """Write a (nested) dict with string keys as an hdf5 file the reference's mirhdf5.hdf5ToDict reads back."""
    h5py = _h5py()
    if not isinstance(data, dict):
        raise Exception('Incorrect input type. Dictionary expected.')
    with h5py.File(filename, 'w') as ff:
        ff.attrs[_H5_TYPE] = b'dict'
        ff.attrs[_H5_ORDER] = [k.encode() for k in data]
        for k, v in data.items():
            _h5_put(ff, k, v)


def _h5_get(node, h5py):
    attrs = node.attrs
    if isinstance(node, h5py.Group):
        kind = attrs.get(_H5_TYPE, b'dict')
        kind = kind.decode() if hasattr(kind, 'decode') else kind
        if kind == 'dict':
            keys = attrs[_H5_ORDER] if _H5_ORDER in attrs else list(node.keys())
            out = {}
            for k in keys:
                k = k.decode() if hasattr(k, 'decode') else k
                out[k] = _h5_get(node[k], h5py)
            return out
        if kind == 'list':
            return [_h5_get(node[k], h5py) for k in node.keys()]
        raise Exception('Unknown group type: {}'.format(kind))
    if _H5_NONE in attrs:
        return None
    value = node[()]
    if _H5_STR in attrs:
        value = value.decode()
    return value


def hdf5_to_dict(filename):
    """Read an hdf5 file written by dict_to_hdf5 or by the reference's mirhdf5.dictToHdf5."""
    h5py = _h5py()
    with h5py.File(filename, 'r') as ff:
        return _h5_get(ff, h5py)


def read_dict(filename):
    filename = pathlib.Path(filename).expanduser()
    ext = filename.suffix
    if 'pickle' in ext or 'pkl' in ext:
        with open(filename, 'rb') as ff:
            return pickle.load(ff)
    if 'json' in ext:
        with open(filename, 'r') as ff:
            return lists_to_numpy(json.load(ff))
    if 'hdf5' in ext or 'h5' in ext:
        return hdf5_to_dict(filename)
    raise NotImplementedError(f'filetype: {ext} not currently supported.')


def write_dict(data, filename, mkdir=False, overwrite=False):
    if mkdir:
        _make_parent(filename)
    filename = pathlib.Path(filename).expanduser()
    if not overwrite and filename.exists():
        raise FileExistsError("File exists. Use overwrite=True to overwrite.")
    ext = filename.suffix
    if 'pickle' in ext or 'pkl' in ext:
        with open(filename, 'wb') as ff:
            pickle.dump(data, ff)
    elif 'json' in ext:
        with open(filename, 'w') as ff:
            json.dump(numpy_to_lists(copy.deepcopy(data)), ff, indent=2)
    elif 'hdf5' in ext or 'h5' in ext:
        dict_to_hdf5(data, filename)
    else:
        raise NotImplementedError(f'filetype: {ext} not currently supported.')


def generate_filename(config, kind=None, name=None, path=None):
    config = xconfig.get_config(config)
    general = config['general']
    if path is None:
        path = general['output_path']
    if kind is None:
        ext = ''
    elif kind in ('image', 'results', 'config'):
        ext = general[kind + '_ext']
    else:
        raise Exception(f'Data kind {kind} unknown.')
    if name is None:
        name = kind
    parts = (general['output_prefix'], name, general['output_suffix'], general['output_run_suffix'])
    return os.path.join(path, '_'.join(filter(None, parts)) + ext)


def load_config(filename):
    return read_dict(filename)


def _target(config, kind, filename, path):
    if filename is None:
        return generate_filename(config, kind=kind, path=path)
    return os.path.join(path, filename) if path is not None else filename


def save_config(config, filename=None, path=None, mkdir=None, overwrite=None):
    filename = _target(config, 'config', filename, path)
    if mkdir is None:
        mkdir = config['general'].get('make_directories', False) if 'general' in config else False
    write_dict(config, filename, mkdir=mkdir, overwrite=overwrite)
    log.info('Config saved to {}'.format(pathlib.Path(filename).expanduser().resolve()))


def save_results(output, filename=None, path=None, mkdir=None, overwrite=None):
    config = output['config']
    filename = _target(config, 'results', filename, path)
    if mkdir is None:
        mkdir = config['general'].get('make_directories', False)
    write_dict(output, filename, mkdir=mkdir, overwrite=overwrite)
    log.info('History saved to {}'.format(pathlib.Path(filename).expanduser().resolve()))


def load_results(filename=None, path=None, config=None):
    return read_dict(_target(config, 'results', filename, path))


def save_images(output, rotate=True, path=None, mkdir=None):
    """One image file per optic with an image, named generate_filename(config, 'image', optic)."""
    from PIL import Image
    config = output['config']
    if mkdir is None:
        mkdir = config['general'].get('make_directories', False)
    for key in config['optics']:
        image = output['total']['image'].get(key)
        if image is None:
            continue
        filename = generate_filename(config, 'image', key, path=path)
        if mkdir:
            _make_parent(filename)
        if rotate:
            image = np.rot90(image)
        Image.fromarray(image).save(filename)
        log.info('Saved image: {}'.format(pathlib.Path(filename).expanduser().resolve()))


def path_exists(path):
    return pathlib.Path(path).expanduser().exists()

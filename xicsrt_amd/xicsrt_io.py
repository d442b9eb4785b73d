"""
Files of the drop-in boundary: config / results dictionaries and detector images
(reference: xicsrt/xicsrt_io.py:27-222, tools/xicsrt_misc.py:18-90).

  load_config / save_config      json (numeric lists <-> ndarrays) or pickle
  save_results / load_results    the raytrace() output dictionary, json or pickle
  save_images                    one image file per optic that made an image (PIL, rot90 as the reference)
  generate_filename              '<prefix>_<name>_<suffix>_<run_suffix><ext>' under output_path

hdf5 (the reference's default `results_ext`, through its own util/mirhdf5 on h5py) is only
available when h5py is importable; this image has no h5py, so '.hdf5' raises with that message.
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
        raise NotImplementedError('hdf5 files need h5py, which is not installed here; use .json or .pickle')
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
        raise NotImplementedError('hdf5 files need h5py, which is not installed here; set results_ext to '
                                  "'.json' or '.pickle'")
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

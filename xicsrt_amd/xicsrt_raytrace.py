"""
Run driver: `raytrace(config)` and friends, on the device.

Mirrors the reference's driver semantics (behaviour, not code):

  raytrace            run loop, per-run seed, merge, reset of per-run options
                      (xicsrt/xicsrt_raytrace.py:28-84)
  raytrace_single     one run = one MT19937 stream; filters -> sources -> optics
                      are built in that order; iteration loop
                      (xicsrt/xicsrt_raytrace.py:87-175)
  _sort_raytrace      found / lost split, shuffled truncation of lost rays
                      (xicsrt/xicsrt_raytrace.py:229-278)
  combine_raytrace    sums of meta and images, concatenated histories
                      (xicsrt/xicsrt_raytrace.py:281-393)
  raytrace_mp         same result as raytrace; the `multiprocessing.Pool` over
                      runs (xicsrt/xicsrt_multiprocessing.py:12-81) is replaced by
                      runs in flight on the GPU, and by one process per GPU
                      (torch.distributed over RCCL) with a single all-reduce of
                      the integer histogram and counters.

Seed schedule: run i uses seed + i(i+1)/2 (cumulative `random_seed += ii`,
xicsrt_raytrace.py:60-63).

All per-ray arithmetic is done by libxicsrt_hip.so through the C ABI
(include/xicsrt_hip.h).  PyTorch only supplies device buffers, the stream and
the process group.
"""
import copy
import ctypes as C
import importlib.util
import glob
import logging
import os

import numpy as np

from . import config as xconfig
from . import scene as xscene
from .objects import RayArray
from . import sources as _sources
from . import optics as _optics
from . import filters as _filters
from . import xicsrt_io

m_log = logging.getLogger('xicsrt')



# ---------------------------------------------------------------------------
# element construction (Dispatcher.instantiate/setup/check_param/initialize,
# xicsrt/objects/_Dispatcher.py:44-140)
# ---------------------------------------------------------------------------

def _builtin_registry(section):
    if section == 'sources':
        return _sources.BUILTIN
    if section == 'optics':
        return _optics.BUILTIN
    if section == 'filters':
        return _filters.BUILTIN
    return {}


def find_class(class_name, section, pathlist):
    """
    Resolve `class_name`.  Built-in names map to this package's classes.  A file
    `_<class_name>.py` on the user `pathlist` is loaded the way the reference
    loads plug-ins (_Dispatcher.py:63-113); it is accepted when the class is a
    re-parametrisation of built-in shapes/interactions (the device cannot run
    arbitrary NumPy methods), otherwise NotImplementedError.
    """
    for path in pathlist or []:
        for filepath in glob.glob(os.path.join(path, '_%s.py' % class_name)):
            spec = importlib.util.spec_from_file_location(class_name, filepath)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            cls = getattr(mod, class_name)
            base = {'sources': (_sources.XicsrtSourceGeneric, _sources.XicsrtPlasmaGeneric),
                    'filters': _filters.XicsrtBundleFilter}.get(section, _optics.TraceObject)
            if not issubclass(cls, base):
                raise NotImplementedError(
                    'User plug-in %s (%s) is not built from xicsrt_amd element classes; arbitrary NumPy '
                    'plug-ins cannot run on the device path and there is no CPU fallback.' % (class_name, filepath))
            return cls
    registry = _builtin_registry(section)
    if class_name in registry:
        return registry[class_name]
    if class_name in getattr(_optics, 'NOT_IMPLEMENTED', ()):
        raise NotImplementedError('%s is not implemented on the device path yet.' % class_name)
    raise Exception('Could not find {} in available objects.'.format(class_name))


class Elements:
    """The initialised objects of one run: one source and the optics in config order."""

    def __init__(self, config):
        general = config['general']
        strict = general['strict_config_check']
        pathlist = list(general.get('pathlist', []) or [])
        sources = config['sources']
        if len(sources) == 0:
            raise Exception('No ray sources defined.')
        if len(sources) != 1:
            raise NotImplementedError('Multiple ray sources are not currently supported.')
        self.objects = {}
        self.source_name = list(sources.keys())[0]
        built = {}
        for section in ('filters', 'sources', 'optics'):
            objs = {}
            for key, sub in (config.get(section) or {}).items():
                cls = find_class(sub['class_name'], section, pathlist)
                objs[key] = cls(sub, initialize=False, strict=strict)
            if section != 'filters':
                # Dispatcher.apply_filters (objects/_Dispatcher.py:198-211): filters named in an
                # element's `filters` list are attached in the order of the filters section
                for obj in objs.values():
                    wanted = obj.config.get('filters') if hasattr(obj.config, 'get') else None
                    if wanted is None:
                        continue
                    for fname, fobj in built['filters'].items():
                        if fname in wanted:
                            obj.filter_objects.append(fobj)
            for method in ('setup', 'check_param', 'initialize'):
                if section == 'filters' and method == 'check_param':
                    continue                                     # (xicsrt_raytrace.py:125-128)
                for obj in objs.values():
                    getattr(obj, method)()
            built[section] = objs
        self.filters = built['filters']
        self.source = built['sources'][self.source_name]
        self.optic_names = list(built['optics'].keys())
        self.optics = [built['optics'][k] for k in self.optic_names]
        self.names = [self.source_name] + self.optic_names

    def get_config(self, section):
        if section == 'sources':
            return {self.source_name: self.source.get_config()}
        return {k: o.get_config() for k, o in zip(self.optic_names, self.optics)}

    def flatten(self):
        return xscene.FlatScene(self.source, self.optics, self.names)


def run_seeds(random_seed, num_runs):
    """Per-run seeds: cumulative increment, i.e. seed + i(i+1)/2 (xicsrt_raytrace.py:60-63).  A schedule that
    leaves np.random.seed's range raises what np.random.seed raises in the reference (:111)."""
    if random_seed is None:
        return [int.from_bytes(os.urandom(4), 'little') for _ in range(num_runs)]
    seeds = []
    s = random_seed
    for ii in range(num_runs):
        s += ii
        seeds.append(_check_seed(int(s)))
    return seeds


def _check_seed(seed):
    if seed < 0 or seed > 2 ** 32 - 1:
        raise ValueError('Seed must be between 0 and 2**32 - 1')
    return seed


def check_config(config):
    """Output directory must exist when anything is to be saved (xicsrt_raytrace.py:396-411)."""
    general = config['general']
    do_save = any(general[key] for key in general if 'save' in key)
    if do_save:
        path = general['output_path']
        if path is None or not os.path.exists(path):
            if not general['make_directories']:
                raise Exception('Output directory does not exist. Create directory or set make_directories to True.')
    # a file type that cannot be written here (hdf5, the default results_ext, without h5py) is refused before
    # the trace, not after it
    if general.get('save_results'):
        xicsrt_io.require_writable(general['results_ext'])
    if general.get('save_config'):
        xicsrt_io.require_writable(general['config_ext'])


# ---------------------------------------------------------------------------
# device plumbing
# ---------------------------------------------------------------------------

def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError('xicsrt_amd needs a HIP device (torch.cuda.is_available() is False); '
                           'there is no CPU fallback.')
    return torch


def _dist():
    """(rank, world) of the default process group, (0, 1) when not distributed."""
    try:
        import torch.distributed as dist
    except Exception:
        return None, 0, 1
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def shard_runs(num_runs, rank, world):
    """Run indices of `rank`: i = rank, rank + world, ... (runs are independent units)."""
    return list(range(rank, num_runs, world))


def pack_counts(num_out, images):
    """One integer vector [num_out | image bins] so that the exchange step is a single all-reduce."""
    import torch
    return torch.cat([num_out.reshape(-1), images.reshape(-1)])


def unpack_counts(packed, n_elements):
    return packed[:n_elements], packed[n_elements:]


def rng_state_from_seed(seed):
    """init_genrand state as np.random.seed(seed) leaves it (key[624], pos=624, no cached gauss)."""
    st = np.random.RandomState(_check_seed(int(seed))).get_state()
    return st[1].astype(np.uint32), int(st[2]), int(st[3]), float(st[4])


def raise_device_status(code, message):
    """Turn a device status (DeviceTrace.status) into the exception the reference raises for it."""
    if code == 0:
        return
    if code in (-6, -7, -10):
        # conditions the reference reports with ValueError while it builds the bundle sources
        # (_XicsrtSourceGeneric.py:193-194, _XicsrtPlasmaGeneric.py:277-281, :368-369)
        raise ValueError(message)
    from .capi import DeviceLibraryError
    raise DeviceLibraryError('xrt_check failed (%d): %s' % (code, message))


class DeviceTrace:
    """Owns the device buffers of one flattened scene and issues the C ABI calls."""

    def __init__(self, flat):
        from . import capi
        self.torch = _torch()
        self.lib = capi.lib()
        self.capi = capi
        self.flat = flat
        capi.check(self.lib.xrt_scene_check(flat.byref()), 'xrt_scene_check')
        t = self.torch
        self.dev = t.device('cuda', t.cuda.current_device())
        # counters and pixel bins are two views of one buffer: the results come over in one copy
        self._acc = t.zeros(flat.n_elements + max(flat.image_bins, 1), dtype=t.int64, device=self.dev)
        self.num_out = self._acc[:flat.n_elements]
        self.images = self._acc[flat.n_elements:]
        self._acc_host = None
        self._ws = None

    def _workspace(self, n_runs):
        # (the size depends on the scene, the run count and the library's route switches; above 256 runs the library
        #  sweeps the layouts of every possible second pass, a millisecond of host time: asked once per case)
        key = (n_runs,) + tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith('XICSRT_')))
        if not hasattr(self, '_ws_need'):
            self._ws_need = {}
        need = self._ws_need.get(key)
        if need is None:
            need = self._ws_need[key] = int(self.lib.xrt_workspace_bytes(self.flat.byref(), n_runs))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            try:
                self._ws = self.torch.empty(max(need, 16), dtype=self.torch.uint8, device=self.dev)
            except self.torch.OutOfMemoryError:
                # The big regions of a workspace are sized to budgets meant for a 288 GB part (include/xicsrt_hip.h,
                # xrt_set_workspace_budget); on a card that is shared, or smaller, the library is told what there is and
                # asked again: it then takes the runs in batches or a route that parks less.
                self.torch.cuda.empty_cache()
                free, _ = self.torch.cuda.mem_get_info(self.dev)
                self.lib.xrt_set_workspace_budget(max(int(free * 0.6), 1 << 20))
                self._ws_need.clear()
                need = self._ws_need[key] = int(self.lib.xrt_workspace_bytes(self.flat.byref(), n_runs))
                self._ws = self.torch.empty(max(need, 16), dtype=self.torch.uint8, device=self.dev)
        return self._ws, need

    def trace(self, seeds, n_iter, keep_images=True):
        """Launch all runs (asynchronous); results accumulate into num_out / images."""
        if len(seeds) == 0:
            return
        seeds_arr = (C.c_uint32 * len(seeds))(*[_check_seed(int(s)) for s in seeds])
        ws, need = self._workspace(len(seeds))
        stream = self.torch.cuda.current_stream().cuda_stream
        st = self.lib.xrt_trace(self.flat.byref(), seeds_arr, len(seeds), int(n_iter),
                                self.num_out.data_ptr(),
                                self.images.data_ptr() if keep_images else None,
                                ws.data_ptr(), need, stream)
        self.capi.check(st, 'xrt_trace')

    def trace_history(self, state, keep_images=True, all_rays=False, on_device=False):
        """
        One iteration from an explicit MT19937 state (key, pos, has_gauss, gauss).
        Returns (rays[n_el, 8, N] float64, mask[n_el, N] bool, state_out) on the host.
        `all_rays`: keep rays whose source mask is off (caller-supplied ray arrays).
        `on_device`: return the device tensors (rays [n_el, 8, capacity], mask uint8) without the big copy.
        """
        t = self.torch
        n, ne = self.flat.n_rays, self.flat.n_elements
        rays = t.full((ne, xscene.XRT_HIST_COMPONENTS, max(n, 1)), float('nan'), dtype=t.float64, device=self.dev)
        mask = t.zeros((ne, max(n, 1)), dtype=t.uint8, device=self.dev)
        st_in = xscene.RngState()
        key, pos, has_gauss, gauss = state
        C.memmove(st_in.key, np.ascontiguousarray(key, dtype=np.uint32).ctypes.data, 624 * 4)
        st_in.pos, st_in.has_gauss, st_in.gauss = int(pos), int(has_gauss), float(gauss)
        st_out = t.zeros(C.sizeof(xscene.RngState), dtype=t.uint8, device=self.dev)
        ws, need = self._workspace(1)
        stream = t.cuda.current_stream().cuda_stream
        status = self.lib.xrt_trace_history(self.flat.byref(), C.byref(st_in),
                                            self.num_out.data_ptr(),
                                            self.images.data_ptr() if keep_images else None,
                                            rays.data_ptr(), mask.data_ptr(), st_out.data_ptr(),
                                            ws.data_ptr(), need, stream)
        self.capi.check(status, 'xrt_trace_history')
        # every xrt_trace_history call starts with a clean status word: read it now, or a capacity overflow /
        # "intensity of less than one" of this iteration would be overwritten by the next one
        self.raise_status()
        t.cuda.current_stream().synchronize()
        out = xscene.RngState.from_buffer_copy(st_out.cpu().numpy().tobytes())
        state_out = (np.ctypeslib.as_array(out.key).copy(), int(out.pos), int(out.has_gauss), float(out.gauss))
        if on_device:
            return rays, mask, state_out
        mask_h = mask.cpu().numpy()[:, :n].astype(bool)
        if not all_rays and self.flat.struct.source.kind == xscene.SRC_KIND['plasma']:
            n = int(mask_h[0].sum())    # plasma sources: the ray count is drawn, n_rays is the capacity
        return rays.cpu().numpy()[:, :, :n], mask_h[:, :n], state_out

    def status(self):
        """(code, message) of the device status word of the calls issued so far (synchronises the stream);
        code 0 = fine, -6 plasma ray capacity exceeded, -7 'intensity of less than one', -10 'No rays generated', -8 Gaussian
        wavelength candidates exhausted (include/xicsrt_hip.h, xrt_check)."""
        if self._ws is None:
            return 0, ''
        code = int(self.lib.xrt_check(self._ws.data_ptr(), self.torch.cuda.current_stream().cuda_stream))
        if code == 0:
            return 0, ''
        msg = self.lib.xrt_last_error()
        return code, (msg.decode() if msg else '?')

    def raise_status(self):
        raise_device_status(*self.status())

    def results(self):
        """Host copies (synchronises the stream): num_out list, {optic: image or None}."""
        t = self.torch
        if self._ws is None:
            t.cuda.current_stream().synchronize()
            host = self._acc.cpu().numpy()
            return self.unpack(host[:self.flat.n_elements], host[self.flat.n_elements:])
        # the status word and the results are queued behind the launches as two copies into pinned memory and waited for
        # once (three blocking copies cost three round trips: a quarter of a small call)
        if self._acc_host is None:
            self._acc_host = t.empty(self._acc.numel(), dtype=t.int64, pin_memory=True)
            self._flags_host = t.empty(4, dtype=t.uint8, pin_memory=True)
        self._flags_host.copy_(self._ws[xscene.XRT_WS_STATUS_BYTE:xscene.XRT_WS_STATUS_BYTE + 4], non_blocking=True)
        self._acc_host.copy_(self._acc, non_blocking=True)
        t.cuda.current_stream().synchronize()
        if int(self._flags_host.view(t.int32)[0]) != 0:
            self.raise_status()             # (the library words the condition: xrt_check)
        host = self._acc_host.numpy().copy()
        return self.unpack(host[:self.flat.n_elements], host[self.flat.n_elements:])

    def unpack(self, num_out, images):
        meta = {}
        for k, name in enumerate(self.flat.names):
            meta[name] = {'num_out': np.int64(num_out[k])}
        image = {}
        for name in self.flat.names[1:]:
            sl = self.flat.image_slices[name]
            if sl is None:
                image[name] = None
            else:
                off, nx, ny = sl
                image[name] = images[off:off + nx * ny].reshape(nx, ny).astype(np.float64)
        return meta, image


# ---------------------------------------------------------------------------
# result dictionaries
# ---------------------------------------------------------------------------

def _empty_output(config):
    return {'config': config,
            'total': {'meta': {}, 'image': {}},
            'found': {'meta': {}, 'history': {}},
            'lost': {'meta': {}, 'history': {}}}


def _history_from_device(names, rays, mask, optics=None):
    """
    Per-element ray dictionaries in original ray order from the device snapshot `rays` [element, 8, ray],
    `mask` [element, ray] (see _history_from_parts).
    """
    # (copies: with a single ray the transposed view is already contiguous and would alias `rays`)
    origin = np.array(rays[:, 0:3, :].transpose(0, 2, 1), order='C', copy=True)
    direction = np.array(rays[:, 3:6, :].transpose(0, 2, 1), order='C', copy=True)
    return _history_from_parts(names, origin, direction, rays[:, 6, :].copy(), rays[0, 7, :].copy(),
                               np.array(mask, copy=True), optics)


def _history_from_parts(names, origin, direction, wavelength, weight, mask, optics=None):
    """
    Per-element ray dictionaries from origin / direction [element, ray, 3], wavelength / mask [element, ray] and
    the source's weight [ray]; the arrays are taken over (updated in place, the dictionaries hold views of them).
    Rays that died at an element carry the intersection point they died at
    (or NaN), later elements see NaN origins and the unchanged direction, as
    the reference's masked NumPy updates leave them (optics/_ShapeObject.py:76-79,
    _InteractObject.py:36-39).  An optic traced in its local frame (`optics[e-1]`
    with trace_local) rotates EVERY ray there and back, dead ones included
    (optics/_TraceObject.py:146-154), which perturbs their directions when the
    orientation matrix is not exactly orthonormal; that round trip is replayed here.
    """
    history = {}
    for e, name in enumerate(names):
        if e > 0:
            dead_before = ~mask[e - 1]
            if dead_before.any():
                origin[e][dead_before] = np.nan
                direction[e][dead_before] = direction[e - 1][dead_before]
                wavelength[e][dead_before] = wavelength[e - 1][dead_before]
                obj = optics[e - 1] if optics is not None and e - 1 < len(optics) else None
                if obj is not None and obj.param.get('trace_local'):
                    d = np.ascontiguousarray(direction[e][dead_before])
                    direction[e][dead_before] = obj.vector_to_external(obj.vector_to_local(d))
        history[name] = RayArray({'origin': origin[e], 'direction': direction[e],
                                  'wavelength': wavelength[e], 'mask': mask[e]})
        if e == 0:
            history[name]['weight'] = weight
    return history


def _sort_history(history, rng, max_lost):
    """found/lost split with the reference's shuffled truncation (xicsrt_raytrace.py:253-274)."""
    found, lost = {}, {}
    keys = list(history.keys())
    last = history[keys[-1]]['mask']
    w_found = np.flatnonzero(last)
    w_lost = np.flatnonzero(np.invert(last))
    max_lost = min(max_lost, len(w_lost))
    index_lost = np.arange(len(w_lost))
    rng.shuffle(index_lost)
    w_lost = w_lost[index_lost[:max_lost]]
    for key in keys:
        found[key] = {k: v[w_found] for k, v in history[key].items()}
        lost[key] = {k: v[w_lost] for k, v in history[key].items()}
    return found, lost


def _shuffled_head(rng, n, m, library_from=500000):
    """np.arange(n) shuffled by `rng` (numpy's legacy RandomState), first m entries; the generator ends where
    rng.shuffle leaves it.  The reference shuffles the indices of all lost rays to keep history_max_lost of them
    (xicsrt_raytrace.py:264-266); the draws are sequential by contract, and for millions of rays numpy's shuffle is the
    longest step of the call (1e7 indices: 0.24 s).  The library makes the same draws and follows only the m wanted cells."""
    if n < library_from:
        index = np.arange(n)
        rng.shuffle(index)
        return index[:m]
    from . import capi
    st = rng.get_state()
    state = xscene.RngState()
    C.memmove(state.key, np.ascontiguousarray(st[1], dtype=np.uint32).ctypes.data, 624 * 4)
    state.pos, state.has_gauss, state.gauss = int(st[2]), int(st[3]), float(st[4])
    out = np.empty(m, dtype=np.int64)
    capi.check(capi.lib().xrt_legacy_shuffle_head(C.byref(state), n, m, out.ctypes.data_as(C.POINTER(C.c_int64))),
               'xrt_legacy_shuffle_head')
    rng.set_state(('MT19937', np.ctypeslib.as_array(state.key).copy(), int(state.pos), int(st[3]), float(st[4])))
    return out


def _sorted_history_from_device(elements, device, d_rays, d_mask, rng, max_lost):
    """
    _sort_raytrace (xicsrt_raytrace.py:229-278) without moving every ray to the host: the masks come
    over (one byte per ray and element), the found rays and the reference's shuffled sample of at most
    `max_lost` lost rays are chosen exactly as _sort_history does (same draws from `rng`), and only those
    rays are gathered on the device and copied.  Per-ray results are independent, so this equals
    _history_from_device + _sort_history on the full arrays.
    """
    t = device.torch
    n = d_mask.shape[1]
    if device.flat.struct.source.kind == xscene.SRC_KIND['plasma']:
        n = int((d_mask[0] != 0).sum())   # plasma sources: the ray count is drawn, the arrays hold the capacity
    last = d_mask[-1, :n] != 0
    d_found = t.nonzero(last).flatten()                   # ascending, as np.flatnonzero
    d_lost = t.nonzero(~last).flatten()
    max_lost = min(max_lost, d_lost.numel())
    head = _shuffled_head(rng, d_lost.numel(), max_lost)
    d_sel = t.cat([d_found, d_lost[t.from_numpy(np.ascontiguousarray(head, dtype=np.int64)).to(d_lost.device)]])
    rays_sel = d_rays.index_select(2, d_sel)
    # laid out on the device as the dictionaries want them, so the host arrays are views of what comes over
    history = _history_from_parts(elements.names,
                                  rays_sel[:, 0:3, :].transpose(1, 2).contiguous().cpu().numpy(),
                                  rays_sel[:, 3:6, :].transpose(1, 2).contiguous().cpu().numpy(),
                                  rays_sel[:, 6, :].contiguous().cpu().numpy(),
                                  rays_sel[0, 7, :].contiguous().cpu().numpy(),
                                  (d_mask.index_select(1, d_sel) != 0).cpu().numpy(), elements.optics)
    nf = d_found.numel()
    found = {key: {k: v[:nf] for k, v in history[key].items()} for key in history}
    lost = {key: {k: v[nf:] for k, v in history[key].items()} for key in history}
    return found, lost


def _summed_meta(results, names):
    """total/meta of the merged result: per element, every counter of the first result summed over all results."""
    merged = {}
    for name in names:
        counters = results[0]['total']['meta'][name]
        merged[name] = {counter: sum(res['total']['meta'][name][counter] for res in results) for counter in counters}
    return merged


def _summed_images(results, names):
    """total/image of the merged result.  An element the first result has no image entry for gets none; an element
    without pixel grid stays None; pixel grids of different shapes cannot be added: warning and None."""
    merged = {}
    for name in names:
        if name not in results[0]['total']['image']:
            continue
        stack = [res['total']['image'][name] for res in results]
        if stack[0] is None:
            merged[name] = None
        elif len({np.shape(img) for img in stack}) == 1:
            merged[name] = np.add.reduce(np.asarray(stack, dtype=np.float64), axis=0)
        else:
            m_log.warning('Image dimensions do not match. Cannot combine images.')
            merged[name] = None
    return merged


def _joined_history(results, group, names):
    """found/history or lost/history of the merged result: per element the four ray fields of a RayArray (a source's
    `weight` is not one of them, App. C.7) of all results one after the other, in the order of `results`.  How many
    rays a result contributes is read off its last element (every element of one result holds the same rays)."""
    counts = [len(res[group]['history'][names[-1]]['mask']) for res in results]
    blank = RayArray()
    blank.zeros(0)
    merged = {}
    for name in names:
        rays = RayArray()
        for field, empty in blank.items():
            parts = [np.asarray(res[group]['history'][name][field])[:cnt] for res, cnt in zip(results, counts)]
            if len(parts) == 1 and parts[0].dtype == empty.dtype:
                rays[field] = np.ascontiguousarray(parts[0])        # one result: its arrays are the answer
            else:
                rays[field] = np.concatenate([empty] + parts).astype(empty.dtype, copy=False)
        merged[name] = rays
    return merged


def combine_raytrace(input_list, keep_images=True, components=None):
    """
    One result dictionary (SURVEY 8b "Result schema") out of several: `config` of the first, `total/meta` summed,
    `total/image` summed (keep_images), `found` / `lost` histories joined when the results carry any.  `components`
    restricts the merge to the named elements.  (Contract: xicsrt/xicsrt_raytrace.py:281-393.)
    """
    results = list(input_list)
    names = list(results[0]['total']['meta']) if components is None else list(components)
    output = _empty_output(results[0]['config'])
    output['total']['meta'] = _summed_meta(results, names)
    if keep_images:
        output['total']['image'] = _summed_images(results, names)
    if len(results[0]['found']['history']) > 0:
        for group in ('found', 'lost'):
            output[group]['history'] = _joined_history(results, group, names)
    return output


def print_raytrace(results):
    keys = list(results['total']['meta'].keys())
    num_source = results['total']['meta'][keys[0]]['num_out']
    num_detector = results['total']['meta'][keys[-1]]['num_out']
    print('')
    print('Rays Generated: {:6.3e}'.format(num_source))
    print('Rays Detected:  {:6.3e}'.format(num_detector))
    print('Efficiency:     {:6.3e} ± {:3.1e} ({:7.5f}%)'.format(
        num_detector / num_source, np.sqrt(num_detector) / num_source, num_detector / num_source * 100))
    print('')


# ---------------------------------------------------------------------------
# public drivers
# ---------------------------------------------------------------------------

def _prepare(config):
    config = xconfig.get_config(config)
    check_config(config)
    elements = Elements(config)
    config['sources'] = elements.get_config('sources')
    config['optics'] = elements.get_config('optics')
    return config, elements


def _advance(rng, state_out):
    """Bring a numpy RandomState to the MT state the device reported."""
    key, pos, has_gauss, gauss = state_out
    rng.set_state(('MT19937', key, pos, has_gauss, gauss))


def _run_with_history(config, elements, device, seed, max_lost_iter):
    """One run with keep_history: iterations are issued one by one so that the lost-ray shuffle
    (xicsrt_raytrace.py:265) consumes the run's stream between them exactly as in the reference."""
    general = config['general']
    rng = np.random.RandomState(_check_seed(int(seed)))
    outputs = []
    for _ in range(general['number_of_iter']):
        st = rng.get_state()
        d_rays, d_mask, state_out = device.trace_history((st[1], st[2], st[3], st[4]), general['keep_images'],
                                                         on_device=True)
        _advance(rng, state_out)
        found, lost = _sorted_history_from_device(elements, device, d_rays, d_mask, rng, max_lost_iter)
        single = _empty_output(config)
        # totals are accumulated on the device and filled in by the caller
        single['total']['meta'] = {name: {'num_out': 0} for name in elements.names}
        single['found']['history'] = found
        single['lost']['history'] = lost
        outputs.append(single)
    return combine_raytrace(outputs)


def _max_lost_iter(general, internal):
    num_iter = general['number_of_iter']
    max_lost = int(general['history_max_lost'] / num_iter)
    if internal:
        max_lost = max_lost // general['number_of_runs']
    return max(int(max_lost), 1)


class _RunImageWriter:
    """
    The reference writes every run's own images when `save_images` is set (raytrace_single,
    xicsrt_raytrace.py:169-170, file names carry the run suffix).  The device accumulates over
    runs, so the image of one run is the difference of the totals before and after it.
    """

    def __init__(self, config, device):
        self.config, self.device = config, device
        self.previous = None

    def __call__(self, run_index):
        meta, image = self.device.results()
        delta = {}
        for name, img in image.items():
            if img is None:
                delta[name] = None
            elif self.previous is None or self.previous.get(name) is None:
                delta[name] = img.copy()
            else:
                delta[name] = img - self.previous[name]
        self.previous = image
        cfg = copy.deepcopy(self.config)
        cfg['general']['output_run_suffix'] = '{:04d}'.format(run_index)
        xicsrt_io.save_images({'config': cfg, 'total': {'image': delta}})


def _raytrace_runs(config, run_indices, seeds, internal, per_run_images=False):
    """Trace the given runs on this process' device; returns one combined result dict."""
    config, elements = _prepare(config)
    general = config['general']
    flat = elements.flatten()
    device = DeviceTrace(flat)
    my_seeds = [seeds[i] for i in run_indices]
    after_run = _RunImageWriter(config, device) if per_run_images else None
    # evaluated whether or not histories are kept, as in the reference (raytrace_single :113-114):
    # number_of_iter = 0 is a ZeroDivisionError there, and here
    max_lost = _max_lost_iter(general, internal)
    if general['keep_history']:
        outputs = []
        for i, s in zip(run_indices, my_seeds):
            outputs.append((i, _run_with_history(config, elements, device, s, max_lost)))
            if after_run:
                after_run(i)
        return outputs, device, config
    if after_run:
        for i, s in zip(run_indices, my_seeds):
            device.trace([s], general['number_of_iter'], general['keep_images'])
            after_run(i)
    else:
        device.trace(my_seeds, general['number_of_iter'], general['keep_images'])
    return None, device, config


def _finish(output, config_user_general):
    """Saving and printing at the end of raytrace() / raytrace_single() (xicsrt_raytrace.py:74-81, :163-170)."""
    general = output['config']['general']
    if general['save_config']:
        xicsrt_io.save_config(output['config'])
    if general['save_images']:
        xicsrt_io.save_images(output)
    if general['save_results']:
        xicsrt_io.save_results(output)
    if general['print_results']:
        print_raytrace(output)
    return output


def _history_payload(single):
    """What travels between ranks of one run's history: the found / lost ray dictionaries only."""
    return {group: {name: {k: np.asarray(v) for k, v in rays.items()}
                    for name, rays in single[group]['history'].items()} for group in ('found', 'lost')}


def raytrace(config):
    """
    Perform `number_of_runs` ray-tracing runs of `number_of_iter` iterations each
    and return the combined results dictionary.

    When a torch.distributed process group is initialised (one process per
    GPU) the runs are sharded over the ranks (run i -> rank i mod world).  The
    exchange step: one all-reduce (RCCL) of the integer vector [num_out | image
    bins], preceded by a one-word all-reduce of the device status so that every
    rank raises when any rank's device reported a condition the reference
    raises for; histories, if kept, are gathered and concatenated in run order
    (xicsrt_raytrace.py:359-390), so every rank returns what a single process
    returns.  Files are written and the summary is printed by rank 0 only.
    """
    config_in = xconfig.get_config(config)
    general = config_in['general']
    num_runs = general['number_of_runs']
    seeds = run_seeds(general['random_seed'], num_runs)
    dist, rank, world = _dist()
    # (a group of ONE rank takes the same branch: the collectives are then RCCL calls that move nothing, and the code that
    #  runs on eight GPUs is the code that runs on one)
    distributed = dist is not None
    if distributed and general['random_seed'] is None:
        # every rank must use the same (random) seeds: rank 0's
        box = [seeds]
        dist.broadcast_object_list(box, src=0)
        seeds = box[0]
    indices = shard_runs(num_runs, rank, world)

    # per-run image files carry the run number in their names: each rank writes those of its own runs
    failure = None
    run_outputs = device = cfg = None
    try:
        run_outputs, device, cfg = _raytrace_runs(copy.deepcopy(config_in), indices, seeds, internal=True,
                                                  per_run_images=bool(general['save_images']))
    except Exception as exc:        # noqa: BLE001
        # Under a process group nothing may be raised on one rank alone: the others would sit in the next collective
        # until the watchdog fires.  (With keep_history / save_images the device status is read run by run, in front
        # of any collective: 'intensity of less than one', a plasma over its capacity ...)  The exception travels to the
        # status agreement below and is raised there, on every rank.
        if not distributed:
            raise
        failure = exc

    if distributed:
        import torch as t
        code, message = (0, '')
        if failure is None:
            code, message = device.status()
        if device is not None:
            where = device.num_out.device
        else:
            where = t.device('cuda', t.cuda.current_device()) if dist.get_backend() == 'nccl' else t.device('cpu')
        flag = t.tensor([1 if (code != 0 or failure is not None) else 0], dtype=t.int64, device=where)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) != 0:
            if failure is not None:
                raise failure
            raise_device_status(code, message)
            raise RuntimeError('another rank reported a device error; results are not valid')
        packed = pack_counts(device.num_out, device.images)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        num_out, images = unpack_counts(packed, device.num_out.numel())
        meta, image = device.unpack(num_out.cpu().numpy(), images.cpu().numpy())
        if general['keep_history']:
            mine = [(i, _history_payload(single)) for i, single in run_outputs]
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            run_outputs = []
            for i, payload in sorted((item for part in gathered for item in part), key=lambda item: item[0]):
                single = _empty_output(cfg)
                single['total']['meta'] = {name: {'num_out': 0} for name in meta}
                for group in ('found', 'lost'):
                    single[group]['history'] = payload[group]
                run_outputs.append((i, single))
    else:
        t = device.torch
        meta, image = device.results()

    output = _empty_output(cfg)
    if run_outputs:
        output = combine_raytrace([single for _, single in run_outputs])
        output['config'] = cfg
    output['total']['meta'] = meta if general['keep_meta'] else {}
    output['total']['image'] = image if general['keep_images'] else {}
    output['config']['general']['output_run_suffix'] = general['output_run_suffix']
    output['config']['general']['random_seed'] = general['random_seed']
    if distributed and rank != 0:
        return output
    return _finish(output, general)


def visible_devices():
    """HIP devices this process can launch on (torch.cuda.device_count() does not initialise the GPU)."""
    import torch
    return int(torch.cuda.device_count())


def device_scope(index):
    """Context that makes device `index` the calling thread's current device."""
    import torch
    return torch.cuda.device(index)


def _trace_on_device(config_in, index, width, seeds, per_run_images):
    """What one pool worker of the reference does for its runs (xicsrt_multiprocessing.py:40-53), on one device:
    runs index, index + width, ... through the C ABI on that device's stream; the host copies of its sums."""
    with device_scope(index):
        run_outputs, device, cfg = _raytrace_runs(copy.deepcopy(config_in), shard_runs(len(seeds), index, width), seeds,
                                                  internal=True, per_run_images=per_run_images)
        meta, image = device.results()           # (raises what the device reported)
    return run_outputs, meta, image, cfg


def _raytrace_devices(config, width):
    """
    `raytrace` fanned out over `width` devices of this process: one host thread per device (the C ABI is handle-free
    and takes the stream per call; ctypes releases the interpreter lock for the duration of a call), run i on device
    i mod width with its own seed, the per-device sums of [num_out | image bins] (120 KB for the bench scene) added
    on the host, histories put together in run order.  Every partition of the runs gives the same sums.
    """
    from concurrent.futures import ThreadPoolExecutor
    config_in = xconfig.get_config(config)
    general = config_in['general']
    seeds = run_seeds(general['random_seed'], general['number_of_runs'])
    per_run_images = bool(general['save_images'])
    with ThreadPoolExecutor(max_workers=width) as pool:
        jobs = [pool.submit(_trace_on_device, config_in, d, width, seeds, per_run_images) for d in range(width)]
        parts = []
        failure = None
        for job in jobs:                         # every device finishes before anything is raised
            try:
                parts.append(job.result())
            except Exception as exc:             # noqa: BLE001 - re-raised below, the first device's first
                failure = failure or exc
        if failure is not None:
            raise failure
    cfg = parts[0][3]
    meta = {name: {'num_out': sum(part[1][name]['num_out'] for part in parts)} for name in parts[0][1]}
    image = {}
    for name, first in parts[0][2].items():
        image[name] = None if first is None else np.add.reduce([part[2][name] for part in parts], axis=0)
    output = _empty_output(cfg)
    singles = sorted((item for part in parts for item in (part[0] or [])), key=lambda item: item[0])
    if singles:
        output = combine_raytrace([single for _, single in singles])
        output['config'] = cfg
    output['total']['meta'] = meta if general['keep_meta'] else {}
    output['total']['image'] = image if general['keep_images'] else {}
    output['config']['general']['output_run_suffix'] = general['output_run_suffix']
    output['config']['general']['random_seed'] = general['random_seed']
    return _finish(output, general)


def raytrace_mp(config, processes=None):
    """
    Drop-in for xicsrt.raytrace_mp (xicsrt_multiprocessing.py:12-81): identical results to `raytrace`.  The reference
    fans the runs out over a pool of `processes` workers (None: all cores); here over the GPUs of this process --
    `processes=None`: every visible device, `processes=k`: at most k of them -- one host thread per device, no
    process group needed.  Under an initialised torch.distributed group (one process per GPU) the runs are already
    sharded over the ranks and this is `raytrace`.
    """
    dist, rank, world = _dist()
    width = 1
    if dist is None:
        width = visible_devices()
        if processes is not None:
            width = min(width, int(processes))
        width = max(1, min(width, xconfig.get_config(config)['general']['number_of_runs']))
    output = raytrace(config) if width <= 1 else _raytrace_devices(config, width)
    # reference quirk kept: raytrace_mp stores output_run_suffix into random_seed
    # (xicsrt_multiprocessing.py:69)
    output['config']['general']['random_seed'] = output['config']['general']['output_run_suffix']
    return output


def raytrace_single(config, _internal=False):
    """One run (`general.random_seed` seeds it) of `number_of_iter` iterations."""
    config_in = xconfig.get_config(config)
    general = config_in['general']
    seed = general['random_seed']
    if seed is None:
        seed = int.from_bytes(os.urandom(4), 'little')
    saved_runs = general['number_of_runs']
    run_outputs, device, cfg = _raytrace_runs(copy.deepcopy(config_in), [0], [seed], internal=_internal)
    meta, image = device.results()
    output = combine_raytrace([single for _, single in run_outputs]) if run_outputs else _empty_output(cfg)
    output['config'] = cfg
    output['total']['meta'] = meta if general['keep_meta'] else {}
    output['total']['image'] = image if general['keep_images'] else {}
    assert saved_runs == general['number_of_runs']
    if _internal:
        return output
    return _finish(output, general)


# ---------------------------------------------------------------------------
# single-object API (plug-in surface): global np.random state in -> out
# ---------------------------------------------------------------------------

def _global_state():
    st = np.random.get_state()
    return st[1], st[2], st[3], st[4]


def _set_global_state(state_out):
    key, pos, has_gauss, gauss = state_out
    np.random.set_state(('MT19937', key, pos, has_gauss, gauss))


def generate_rays_from_global_state(source_obj):
    """XicsrtSource*.generate_rays(): consumes the global legacy stream like the reference."""
    flat = xscene.FlatScene(source_obj, [], ['source'])
    device = DeviceTrace(flat)
    rays, mask, state_out = device.trace_history(_global_state(), keep_images=False)
    _set_global_state(state_out)
    out = RayArray({'origin': np.array(rays[0, 0:3, :].T, order='C', copy=True),
                    'direction': np.array(rays[0, 3:6, :].T, order='C', copy=True),
                    'wavelength': rays[0, 6, :].copy(),
                    'mask': mask[0].copy()})
    out['weight'] = rays[0, 7, :].copy()
    return out


def _rays_to_device(rays):
    """RayArray / dict of (n,3),(n,3),(n,),(n,) arrays -> device tensors [8, n] float64 and [n] uint8."""
    t = _torch()
    n = len(rays['mask'])
    host = np.empty((xscene.XRT_HIST_COMPONENTS, n), dtype=np.float64)
    host[0:3] = np.asarray(rays['origin'], dtype=np.float64).T
    host[3:6] = np.asarray(rays['direction'], dtype=np.float64).T
    host[6] = np.asarray(rays['wavelength'], dtype=np.float64)
    host[7] = np.asarray(rays['weight'], dtype=np.float64) if 'weight' in rays else 1.0
    dev = t.device('cuda', t.cuda.current_device())
    d_rays = t.from_numpy(host).to(dev)
    d_mask = t.from_numpy(np.ascontiguousarray(np.asarray(rays['mask']), dtype=np.uint8)).to(dev)
    return d_rays, d_mask


def trace_optic_object(optic_obj, rays):
    """
    XicsrtOptic*.trace_global(rays) on a caller's ray array (optics/_TraceObject.py:135-178): the rays
    are updated in place and returned, like the reference's masked NumPy updates leave them (dead rays
    get a NaN origin or the point they died at; directions change only for reflected rays).  A Bragg
    test draws from the global legacy np.random stream, which is advanced accordingly.
    """
    d_rays, d_mask = _rays_to_device(rays)
    flat = xscene.FlatScene(xscene.ExternalRays(d_rays, d_mask), [optic_obj], ['rays', 'optic'])
    device = DeviceTrace(flat)
    out, mask, state_out = device.trace_history(_global_state(), keep_images=False, all_rays=True)
    _set_global_state(state_out)
    history = _history_from_device(['rays', 'optic'], out, mask, [optic_obj])
    after = history['optic']
    rays['origin'][:] = after['origin']
    rays['direction'][:] = after['direction']
    rays['mask'][:] = after['mask']
    return rays


def image_of_optic_object(optic_obj, rays):
    """XicsrtOptic*.make_image(rays) (optics/_TraceObject.py:234-293): float64 (nx, ny) counts or None."""
    if not optic_obj.param['enable_image']:
        return None
    t = _torch()
    from . import capi
    d_rays, d_mask = _rays_to_device(rays)
    flat = xscene.FlatScene(xscene.ExternalRays(d_rays, d_mask), [optic_obj], ['rays', 'optic'])
    o = flat.struct.optics[0]
    images = t.zeros(max(flat.image_bins, 1), dtype=t.int64, device=d_rays.device)
    capi.check(capi.lib().xrt_make_image(C.byref(o), len(rays['mask']), d_rays.data_ptr(), d_mask.data_ptr(),
                                         images.data_ptr(), t.cuda.current_stream().cuda_stream), 'xrt_make_image')
    t.cuda.current_stream().synchronize()
    return images.cpu().numpy()[:o.pixel_nx * o.pixel_ny].reshape(o.pixel_nx, o.pixel_ny).astype(np.float64)


# ---------------------------------------------------------------------------
# TraceObject.intersect / check_bounds / interact as separate device calls
# ---------------------------------------------------------------------------

def _one_optic(optic_obj, n):
    """The C struct of one optic (through a one-optic scene, so that all derived constants are the scene's)."""
    t = _torch()
    dev = t.device('cuda', t.cuda.current_device())
    dummy = xscene.ExternalRays(t.zeros((xscene.XRT_HIST_COMPONENTS, max(n, 1)), dtype=t.float64, device=dev),
                                t.zeros(max(n, 1), dtype=t.uint8, device=dev))
    flat = xscene.FlatScene(dummy, [optic_obj], ['rays', 'optic'])
    return flat, flat.struct.optics[0]


def _vec_to_device(t, a):
    """(n, 3) host array -> [3][n] device tensor."""
    return t.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).T)).to(t.device('cuda', t.cuda.current_device()))


def optic_intersect(optic_obj, rays):
    from . import capi
    t = _torch()
    n = len(rays['mask'])
    flat, o = _one_optic(optic_obj, n)
    d_rays, d_mask = _rays_to_device(rays)
    xloc = t.empty((3, max(n, 1)), dtype=t.float64, device=d_rays.device)
    norm = t.empty((3, max(n, 1)), dtype=t.float64, device=d_rays.device)
    m_out = t.empty(max(n, 1), dtype=t.uint8, device=d_rays.device)
    capi.check(capi.lib().xrt_optic_intersect(C.byref(o), n, d_rays.data_ptr(), d_mask.data_ptr(), xloc.data_ptr(),
                                              norm.data_ptr(), m_out.data_ptr(), t.cuda.current_stream().cuda_stream),
               'xrt_optic_intersect')
    t.cuda.current_stream().synchronize()
    rays['mask'][:] = m_out.cpu().numpy()[:n].astype(bool)
    return (np.ascontiguousarray(xloc.cpu().numpy()[:, :n].T), np.ascontiguousarray(norm.cpu().numpy()[:, :n].T),
            rays['mask'])


def optic_check_bounds(optic_obj, X, mask):
    from . import capi
    t = _torch()
    n = len(mask)
    flat, o = _one_optic(optic_obj, n)
    d_x = _vec_to_device(t, X)
    d_m = t.from_numpy(np.ascontiguousarray(np.asarray(mask), dtype=np.uint8)).to(d_x.device)
    capi.check(capi.lib().xrt_optic_check_bounds(C.byref(o), n, d_x.data_ptr(), d_m.data_ptr(),
                                                 t.cuda.current_stream().cuda_stream), 'xrt_optic_check_bounds')
    t.cuda.current_stream().synchronize()
    mask[:] = d_m.cpu().numpy().astype(bool)
    return mask


def optic_interact(optic_obj, rays, xloc, norm, mask=None):
    from . import capi
    t = _torch()
    if mask is None:
        mask = rays['mask']
    n = len(mask)
    flat, o = _one_optic(optic_obj, n)
    d_rays, _ = _rays_to_device(rays)
    d_x, d_n = _vec_to_device(t, xloc), _vec_to_device(t, norm)
    d_m = t.from_numpy(np.ascontiguousarray(np.asarray(mask), dtype=np.uint8)).to(d_rays.device)
    d_test = None
    if o.interact == xscene.INTERACT['crystal'] and (o.flags & xscene.F_CHECK_BRAGG):
        # rocking_curve_filter (optics/_InteractCrystal.py:189): one uniform deviate per ray still alive, in ray order
        live = np.flatnonzero(np.asarray(mask))
        test = np.zeros(n, dtype=np.float64)
        test[live] = np.random.uniform(0.0, 1.0, len(live))
        d_test = t.from_numpy(test).to(d_rays.device)
    capi.check(capi.lib().xrt_optic_interact(C.byref(o), n, d_rays.data_ptr(), d_x.data_ptr(), d_n.data_ptr(), d_m.data_ptr(),
                                             d_test.data_ptr() if d_test is not None else None,
                                             t.cuda.current_stream().cuda_stream), 'xrt_optic_interact')
    t.cuda.current_stream().synchronize()
    out = d_rays.cpu().numpy()
    rays['origin'][:] = out[0:3, :n].T
    rays['direction'][:] = out[3:6, :n].T
    new_mask = d_m.cpu().numpy().astype(bool)
    # The reference's interact() works on the caller's mask array in place: InteractCrystal.angle_check thins it out
    # (`m[m] &= ...`, _InteractCrystal.py:128) and interact() then makes it rays['mask'] (:93); InteractNone / Mirror
    # copy it into rays['mask'] (_InteractObject.py:32-33, _InteractMirror.py:25-26).
    if isinstance(mask, np.ndarray) and mask is not rays['mask'] and mask.shape == new_mask.shape:
        mask[:] = new_mask
        if o.interact == xscene.INTERACT['crystal']:
            rays['mask'] = mask
            return rays
    if isinstance(rays['mask'], np.ndarray) and rays['mask'].shape == new_mask.shape:
        rays['mask'][:] = new_mask
    else:
        rays['mask'] = new_mask
    return rays

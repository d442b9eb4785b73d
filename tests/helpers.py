"""
Test helpers: golden fixtures, the CPU oracle binding, scene building.

The oracle (oracle/libxrt_oracle.so) is test infrastructure; it is loaded only
from here, from __graft_entry__.smoke() and from bench.py's cpu_baseline leg.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
ORACLE_DIR = os.path.join(ROOT, 'oracle')
ORACLE_LIB = os.path.join(ORACLE_DIR, 'libxrt_oracle.so')

from xicsrt_amd import scene as xscene          # noqa: E402
from xicsrt_amd import xicsrt_raytrace as xrt         # noqa: E402
from xicsrt_amd import config as xconfig        # noqa: E402

_oracle = None


def load_oracle():
    global _oracle
    if _oracle is not None:
        return _oracle
    src = os.path.join(ORACLE_DIR, 'xrt_oracle.c')
    hdr = os.path.join(ROOT, 'include', 'xicsrt_hip.h')
    stale = (not os.path.exists(ORACLE_LIB)
             or os.path.getmtime(ORACLE_LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if stale:
        subprocess.check_call(['make', '-C', ORACLE_DIR, '-s'])
    L = C.CDLL(ORACLE_LIB)
    P = C.POINTER
    L.xrt_oracle_sizeof_scene.restype = C.c_size_t
    assert L.xrt_oracle_sizeof_scene() == C.sizeof(xscene.Scene), 'scene layout mismatch'
    L.xrt_oracle_trace.restype = C.c_int
    L.xrt_oracle_trace.argtypes = [P(xscene.Scene), P(C.c_uint32), C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_int32]
    L.xrt_oracle_trace_history.restype = C.c_int
    L.xrt_oracle_trace_history.argtypes = [P(xscene.Scene), P(xscene.RngState), C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, P(xscene.RngState)]
    for name, ct in (('xrt_oracle_mt_u32', C.c_uint32), ('xrt_oracle_mt_double', C.c_double),
                     ('xrt_oracle_mt_gauss', C.c_double)):
        getattr(L, name).restype = None
        getattr(L, name).argtypes = [C.c_uint32, C.c_void_p, C.c_int64]
    _oracle = L
    return L


def golden_names(kind=None):
    index = json.load(open(os.path.join(GOLDEN, 'INDEX.json')))
    return sorted(k for k, v in index.items() if kind is None or v['kind'] == kind)


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + '.npz'))
    cfg = json.loads(str(d['config_json']))
    cfg.pop('caller_mask', None)       # object-level cases: a note for make_golden.py, the mask itself is in the fixture
    # data fixtures (profile files) are named relative to the repo root
    for section in ('sources', 'optics'):
        for sub in cfg.get(section, {}).values():
            for k, v in sub.items():
                if k.endswith('_file') and isinstance(v, str) and not os.path.isabs(v):
                    sub[k] = os.path.join(ROOT, v)
    return cfg, d


def build(cfg):
    """config -> (merged config, Elements, FlatScene) through the product's host code."""
    config = xconfig.get_config(cfg)
    elements = xrt.Elements(config)
    return config, elements, elements.flatten()


def seed_state(seed):
    key, pos, has_gauss, gauss = xrt.rng_state_from_seed(seed)
    st = xscene.RngState()
    C.memmove(st.key, key.ctypes.data, 624 * 4)
    st.pos, st.has_gauss, st.gauss = pos, has_gauss, gauss
    return st


def state_next_double(st):
    """Next double numpy's legacy generator yields from this state."""
    rs = np.random.RandomState(0)
    rs.set_state(('MT19937', np.ctypeslib.as_array(st.key).copy(), int(st.pos), int(st.has_gauss), float(st.gauss)))
    return rs.random_sample()


def oracle_counts(flat, seeds, n_iter, threads=1):
    L = load_oracle()
    num_out = np.zeros(flat.n_elements, dtype=np.uint64)
    images = np.zeros(max(flat.image_bins, 1), dtype=np.uint64)
    arr = (C.c_uint32 * len(seeds))(*seeds)
    st = L.xrt_oracle_trace(flat.byref(), arr, len(seeds), n_iter,
                            num_out.ctypes.data, images.ctypes.data, threads)
    assert st == 0
    return num_out.astype(np.int64), images.astype(np.int64)


def oracle_history(flat, state, all_rays=False):
    L = load_oracle()
    n, ne = flat.n_rays, flat.n_elements
    num_out = np.zeros(ne, dtype=np.uint64)
    images = np.zeros(max(flat.image_bins, 1), dtype=np.uint64)
    rays = np.full((ne, xscene.XRT_HIST_COMPONENTS, max(n, 1)), np.nan)
    mask = np.zeros((ne, max(n, 1)), dtype=np.uint8)
    out = xscene.RngState()
    st = L.xrt_oracle_trace_history(flat.byref(), C.byref(state), num_out.ctypes.data, images.ctypes.data,
                                    rays.ctypes.data, mask.ctypes.data, C.byref(out))
    assert st == 0
    if not all_rays and flat.struct.source.kind == xscene.SRC_KIND['plasma']:
        n = int(mask[0].sum())  # plasma sources: drawn ray count, n_rays is the capacity
    return num_out.astype(np.int64), images.astype(np.int64), rays[:, :, :n], mask[:, :n].astype(bool), out


def split_images(flat, images):
    out = {}
    for name in flat.names[1:]:
        sl = flat.image_slices[name]
        if sl is not None:
            off, nx, ny = sl
            out[name] = images[off:off + nx * ny].reshape(nx, ny)
    return out


# Float tolerance of the parity tests: 1e-12 relative (the task's bound is 1e-6).  One fixture needs more:
# the reference's near-flat torus of testing/integrated_test_02 (radii 1e5 / 0.5e5 m) makes the quartic
# ill-conditioned; its complex Ferrari/Cardano evaluation and the real-arithmetic restatement agree to
# 1.6e-11 there (masks and counts are identical).
RTOL_DEFAULT = 1e-12
RTOL_CASE = {'I2_ToroidalCrystal_trace': 1e-9}
# The Y_fuzz cases (found by tests/fuzz_parity.py): the position of ONE LOST ray each -- a ray that grazes the next plane
# and is recorded 24 km away, a ray on a double root of the torus quartic.  The oracle (libm) agrees with the reference to
# 7e-13 there; the device (its own square roots, divisions and last-ulp transcendental differences, amplified by the
# ill-conditioned point) to 1.7e-9 and 1.3e-8.  Masks, counters and pixels are equal; the task's bound is 1e-6.
RTOL_CASE_DEVICE = {'Y_fuzz_9011073_trace': 1e-7, 'Y_fuzz_9015074_trace': 1e-7, 'Y_fuzz_14039276_trace': 1e-7}
# ... and how many rays of the scene may need it (every other ray is held to the tight tolerance): measured on the device,
# three of the 40 000 rays of Y_fuzz_9011073 (30456, 31178, 38131) and of Y_fuzz_14039276 (12085, 23654, 36193) -- rays on
# or next to a double root of the torus quartic / grazing the next plane
OUTLIER_RAYS = {'Y_fuzz_9011073_trace': 4, 'Y_fuzz_9015074_trace': 4, 'Y_fuzz_14039276_trace': 4}


def rtol_for(name, device=False):
    """The tolerance every ray of the case is held to.  (The device's looser bound for the Y_fuzz cases applies to ONE ray of
    the scene only: see outlier_rtol_for.)"""
    return RTOL_CASE.get(name, RTOL_DEFAULT)


def outlier_rtol_for(name, device=False):
    """The bound for the single ill-conditioned ray of a Y_fuzz case on the device (None: the case has no such ray)."""
    return RTOL_CASE_DEVICE.get(name) if device else None


def assert_history_matches_golden(flat, rays, mask, gold, rtol=1e-12, outlier_rtol=None, max_outliers=1):
    """
    Compare a device/oracle history snapshot with a golden 'trace' fixture:
    masks bit-exact for every ray at every element; origin, direction and
    wavelength of rays alive after the element within `rtol` (relative to the
    vector's magnitude); rays that died at an element carry the point they
    died at, NaN pattern included.  `outlier_rtol`: `max_outliers` rays of the
    scene (the same rays at every element) may be off by up to that instead.
    """
    hist = xrt._history_from_device(flat.names, rays, mask, flat.optic_objs)
    outliers = set()
    for name in flat.names:
        gm = gold['mask/' + name]
        assert np.array_equal(hist[name]['mask'], gm), 'mask mismatch at %s' % name
        for key in ('origin', 'direction', 'wavelength'):
            g = gold[key + '/' + name]
            h = hist[name][key]
            assert np.array_equal(np.isnan(h), np.isnan(g)), 'NaN pattern of %s at %s' % (key, name)
            ok = ~np.isnan(g)
            scale = max(np.max(np.abs(g[ok])) if ok.any() else 1.0, 1e-300)
            diff = np.where(ok, np.abs(h - g), 0.0)
            per_ray = diff.reshape(diff.shape[0], -1).max(axis=1) if diff.size else np.zeros(0)
            err = per_ray.max() if per_ray.size else 0.0
            if outlier_rtol is None:
                assert err <= rtol * scale, '%s at %s: err %.3e (scale %.3e)' % (key, name, err, scale)
            else:
                outliers |= set(np.flatnonzero(per_ray > rtol * scale).tolist())
                assert err <= outlier_rtol * scale, '%s at %s: err %.3e (scale %.3e)' % (key, name, err, scale)
    assert len(outliers) <= max_outliers, 'more than %d rays beyond %.1e: %s' % (max_outliers, rtol, sorted(outliers))


class OracleDeviceTrace:
    """
    Stand-in for xicsrt_raytrace.DeviceTrace on machines without a GPU: same interface, CPU tensors, the
    oracle as the per-rank compute.  Lets the CPU suite drive the product's own host code paths
    (raytrace()'s process-group branch, history gathering, saving) end to end; never used by the product.
    `fail_code`: what status() reports (to test the error agreement between ranks).
    """
    fail_code = 0

    def __init__(self, flat):
        import torch
        self.torch = torch
        self.flat = flat
        self.num_out = torch.zeros(flat.n_elements, dtype=torch.int64)
        self.images = torch.zeros(max(flat.image_bins, 1), dtype=torch.int64)

    def trace(self, seeds, n_iter, keep_images=True):
        if len(seeds) == 0:
            return
        n, img = oracle_counts(self.flat, [int(s) for s in seeds], int(n_iter))
        self.num_out += self.torch.from_numpy(n)
        if keep_images:
            self.images += self.torch.from_numpy(img)

    def trace_history(self, state, keep_images=True, all_rays=False, on_device=False):
        self.raise_status()         # (DeviceTrace.trace_history reads the device status of every call on the spot)
        key, pos, has_gauss, gauss = state
        st = xscene.RngState()
        C.memmove(st.key, np.ascontiguousarray(key, dtype=np.uint32).ctypes.data, 624 * 4)
        st.pos, st.has_gauss, st.gauss = int(pos), int(has_gauss), float(gauss)
        n, img, rays, mask, out = oracle_history(self.flat, st, all_rays=True)
        self.num_out += self.torch.from_numpy(n)
        if keep_images:
            self.images += self.torch.from_numpy(img)
        state_out = (np.ctypeslib.as_array(out.key).copy(), int(out.pos), int(out.has_gauss), float(out.gauss))
        if on_device:
            return self.torch.from_numpy(rays.copy()), self.torch.from_numpy(mask.astype(np.uint8)), state_out
        nn = int(mask[0].sum()) if (not all_rays and self.flat.struct.source.kind == xscene.SRC_KIND['plasma']) else rays.shape[2]
        return rays[:, :, :nn], mask[:, :nn], state_out

    def status(self):
        return (self.fail_code, 'intensity of less than one encountered. Turn on poisson statistics.') \
            if self.fail_code else (0, '')

    def raise_status(self):
        xrt.raise_device_status(*self.status())

    def results(self):
        self.raise_status()
        return self.unpack(self.num_out.numpy(), self.images.numpy())

    unpack = xrt.DeviceTrace.unpack       # (bound at import: tests replace xrt.DeviceTrace by this class)

"""
ctypes binding of libxicsrt_hip.so (include/xicsrt_hip.h).

The device library is the product: if it is missing or cannot be loaded the
import of this module's `lib()` fails loudly.  There is no CPU fallback.
"""
import ctypes as C
import os

from . import scene as _scene

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('XICSRT_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libxicsrt_hip.so')

EXPORTS = (
    'xrt_abi_version', 'xrt_last_error', 'xrt_sizeof_scene', 'xrt_scene_check',
    'xrt_device_count', 'xrt_workspace_bytes', 'xrt_trace', 'xrt_trace_history',
    'xrt_timing_begin', 'xrt_timing_end', 'xrt_mt_jump_poly', 'xrt_check', 'xrt_make_image', 'xrt_last_path',
    'xrt_optic_intersect', 'xrt_optic_check_bounds', 'xrt_optic_interact', 'xrt_selftest_div3', 'xrt_legacy_shuffle_head',
    'xrt_status_offset', 'xrt_set_workspace_budget', 'xrt_mt_jump_polys',
)
PATH_FUSED, PATH_STAGED, PATH_STAGE_SPLIT, PATH_JUMP, PATH_SEEK, PATH_SEGMENTED, PATH_GAUSS_PREPARED = 1, 2, 4, 8, 16, 32, 64
PATH_PLASMA_SCOUT = 128
PATH_LDS_BINS = 256
PATH_ONE_PASS = 512
PATH_MESH_SPLIT = 1024
PATH_MESH_FANS = 2048
PATH_MOSAIC_FUSED = 4096

_lib = None


class DeviceLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the HIP library with prototypes set."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DeviceLibraryError(
            'HIP extension not built: %s is missing. Run `python -c "import __graft_entry__ as g; g.build()"` '
            '(or xicsrt_amd/csrc/build.sh). There is no CPU fallback.' % LIB_PATH)
    try:
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise DeviceLibraryError('cannot load %s: %s' % (LIB_PATH, e))
    for name in EXPORTS:
        if not hasattr(L, name):
            raise DeviceLibraryError('%s does not export %s' % (LIB_PATH, name))
    P = C.POINTER
    L.xrt_abi_version.restype = C.c_int
    L.xrt_last_error.restype = C.c_char_p
    L.xrt_sizeof_scene.restype = C.c_size_t
    L.xrt_scene_check.restype = C.c_int
    L.xrt_scene_check.argtypes = [P(_scene.Scene)]
    L.xrt_device_count.restype = C.c_int
    L.xrt_device_count.argtypes = [P(C.c_int)]
    L.xrt_workspace_bytes.restype = C.c_size_t
    L.xrt_workspace_bytes.argtypes = [P(_scene.Scene), C.c_int32]
    L.xrt_trace.restype = C.c_int
    L.xrt_trace.argtypes = [P(_scene.Scene), P(C.c_uint32), C.c_int32, C.c_int32,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.xrt_trace_history.restype = C.c_int
    L.xrt_trace_history.argtypes = [P(_scene.Scene), P(_scene.RngState), C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_size_t, C.c_void_p]
    L.xrt_mt_jump_poly.restype = C.c_int
    L.xrt_mt_jump_poly.argtypes = [C.c_uint64, P(C.c_uint32)]
    L.xrt_mt_jump_polys.restype = C.c_int
    L.xrt_mt_jump_polys.argtypes = [P(C.c_uint64), C.c_int32, P(C.c_uint32)]
    L.xrt_check.restype = C.c_int
    L.xrt_check.argtypes = [C.c_void_p, C.c_void_p]
    L.xrt_make_image.restype = C.c_int
    L.xrt_make_image.argtypes = [P(_scene.Optic), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.xrt_optic_intersect.restype = C.c_int
    L.xrt_optic_intersect.argtypes = [P(_scene.Optic), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
    L.xrt_optic_check_bounds.restype = C.c_int
    L.xrt_optic_check_bounds.argtypes = [P(_scene.Optic), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.xrt_optic_interact.restype = C.c_int
    L.xrt_optic_interact.argtypes = [P(_scene.Optic), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
    L.xrt_selftest_div3.restype = C.c_int
    L.xrt_selftest_div3.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.xrt_last_path.restype = C.c_uint32
    L.xrt_legacy_shuffle_head.restype = C.c_int
    L.xrt_legacy_shuffle_head.argtypes = [P(_scene.RngState), C.c_int64, C.c_int64, P(C.c_int64)]
    L.xrt_last_path.argtypes = [C.c_int32]
    L.xrt_status_offset.restype = C.c_size_t
    L.xrt_set_workspace_budget.restype = None
    L.xrt_set_workspace_budget.argtypes = [C.c_size_t]
    L.xrt_timing_begin.restype = C.c_int
    L.xrt_timing_end.restype = C.c_int
    L.xrt_timing_end.argtypes = [P(C.c_double), P(C.c_int64)]
    if L.xrt_abi_version() != _scene.XRT_ABI_VERSION:
        raise DeviceLibraryError('ABI version mismatch: library %d, binding %d'
                                 % (L.xrt_abi_version(), _scene.XRT_ABI_VERSION))
    if L.xrt_sizeof_scene() != C.sizeof(_scene.Scene):
        raise DeviceLibraryError('struct xrt_scene layout mismatch: library %d bytes, binding %d'
                                 % (L.xrt_sizeof_scene(), C.sizeof(_scene.Scene)))
    if L.xrt_status_offset() != _scene.XRT_WS_STATUS_BYTE:
        raise DeviceLibraryError('workspace status word: library keeps it at byte %d, binding reads byte %d'
                                 % (L.xrt_status_offset(), _scene.XRT_WS_STATUS_BYTE))
    _lib = L
    return L


def check(status, what):
    if status != 0:
        msg = lib().xrt_last_error()
        raise DeviceLibraryError('%s failed (%d): %s' % (what, status, msg.decode() if msg else '?'))

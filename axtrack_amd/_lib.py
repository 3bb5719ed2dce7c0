"""ctypes binding of libaxtrack_hip.so (include/axtrack_hip.h). No CPU fallback: if the HIP
library is missing or a call fails, this module raises."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# AXT_LIB_PATH: a diagnostic build of the same library (profiles/wino_stamps.py); there is still no fallback
LIB_PATH = os.environ.get('AXT_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libaxtrack_hip.so')

c_void_p, c_int, c_float, c_double, c_size_t, c_int64 = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_size_t, ctypes.c_int64)

# name -> (restype, argtypes); mirrors include/axtrack_hip.h one to one
SIGNATURES = {
    'axt_last_error': (ctypes.c_char_p, []),
    'axt_abi_version': (c_int, []),
    'axt_detector_create': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, ctypes.POINTER(c_void_p)]),
    'axt_detector_destroy': (None, [c_void_p]),
    'axt_detector_set_arith': (c_int, [c_void_p, c_int]),
    'axt_detector_set_fused_front': (c_int, [c_void_p, c_int]),
    'axt_detector_device_bytes': (c_size_t, [c_void_p]),
    'axt_cnn_forward': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'axt_cnn_forward_frames': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                       c_void_p, c_void_p]),
    'axt_cnn_front_frames': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    'axt_cnn_back': (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    'axt_cnn_flops_per_tile': (c_double, []),
    'axt_detector_set_profiling': (c_int, [c_void_p, c_int]),
    'axt_detector_read_profile': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    'axt_cnn_kernel_flops_per_tile': (c_double, [c_int]),
    'axt_preprocess_u16': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_int, c_float, c_void_p,
                                   c_void_p]),
    'axt_tile_occupancy': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'axt_decode_stitch_nms': (c_int, [c_void_p, c_int, c_int, c_void_p, c_float, c_int, c_int, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p]),
    'axt_grid_create': (c_int, [c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    'axt_grid_destroy': (None, [c_void_p]),
    'axt_obs_costs': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_void_p]),
    'axt_path_cost': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                              c_int, c_void_p, c_void_p]),
    'axt_path_cells': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                               c_int, c_void_p, c_void_p, c_void_p]),
    'axt_build_arcs': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                               c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, ctypes.POINTER(c_int64), c_void_p]),
    'axt_build_arcs_rows': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                    c_int, c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_int64), c_void_p,
                                    c_void_p]),
    'axt_build_arcs_from_lengths': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            ctypes.POINTER(c_int64), c_void_p]),
    'axt_box_histograms': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p]),
    'axt_build_arcs_vis': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                   c_int, c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_int64), c_void_p]),
    'axt_mcf_solve': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                              c_void_p, c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int64)]),
    'axt_mcf_solve_duals': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                    c_void_p, c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int64),
                                    c_void_p, c_void_p, ctypes.POINTER(c_int64)]),
    'axt_mcf_shard_begin': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                    ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64)]),
    'axt_mcf_shard_export': (c_int, [c_void_p, c_void_p]),
    'axt_mcf_shard_finish': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, ctypes.POINTER(c_int),
                                     ctypes.POINTER(c_int64)]),
    'axt_mcf_shard_finish_duals': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, ctypes.POINTER(c_int),
                                           ctypes.POINTER(c_int64), c_void_p, c_void_p, ctypes.POINTER(c_int64)]),
    'axt_mcf_shard_free': (None, [c_void_p]),
    'axt_hungarian_assoc': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'axt_hungarian_pairs': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'axt_hungarian_pairs_grid': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'axt_hungarian_pairs_costs': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'axt_chain_tracks': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'axt_ided_table': (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'axt_detection_confusion': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'axt_arc_cost_int': (c_int64, [c_double, c_int, c_int64, c_int64]),
}

_lib = None


class AxtError(RuntimeError):
    pass


def load():
    """Load the HIP library; raise (never fall back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AxtError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           f'or `make -C axtrack_amd/csrc`. axtrack_amd has no CPU fallback.')
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc < 0:
        raise AxtError(f'{what} failed ({rc}): {load().axt_last_error().decode()}')
    return rc


def dptr(t):
    """Device (or host) address of a torch tensor / numpy array; None -> NULL."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        return t.ctypes.data
    return t.data_ptr()

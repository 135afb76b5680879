"""ctypes binding of the C ABI in include/drp.h.

The shared library `libdrp.so` is built in-tree by `__graft_entry__.build()` (hipcc,
gfx950).  There is no CPU fallback: if the library is missing or no GPU is present,
creating an engine raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DRP_LIB: alternative build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get('DRP_LIB') or os.path.join(_HERE, 'libdrp.so')

DRP_K = 10
DRP_F = 64
DRP_N_WEIGHTS = 38403
ENGINE_VALU = 0
ENGINE_MFMA = 1
ENGINE_SPLIT = 2
ENGINE_FUSED = 3
TRAIN_MODES = {'eval': 0, 'grad': 1, 'update': 2}
DIST_TRANSFORMS = {'cv5': 0, 'exact': 1}
NOISE_TYPES = {'normal': 0, 'uniform': 1, 'total_rand': 2}
ENGINES = {'valu': ENGINE_VALU, 'mfma': ENGINE_MFMA, 'split': ENGINE_SPLIT, 'fused': ENGINE_FUSED}

c_float_p = ctypes.POINTER(ctypes.c_float)
c_double_p = ctypes.POINTER(ctypes.c_double)
c_int16_p = ctypes.POINTER(ctypes.c_int16)
c_uint8_p = ctypes.POINTER(ctypes.c_uint8)


class MpcParams(ctypes.Structure):
    """struct drp_mpc_params (include/drp.h)."""
    _fields_ = [('n_batch', ctypes.c_int), ('n_particles', ctypes.c_int),
                ('n_sample', ctypes.c_int), ('n_look_ahead', ctypes.c_int),
                ('sigma', ctypes.c_double), ('beta_filter', ctypes.c_double),
                ('reward_weight', ctypes.c_double),
                ('act_lo', ctypes.c_float * 4), ('act_hi', ctypes.c_float * 4),
                ('seed', ctypes.c_uint64), ('sample_offset', ctypes.c_uint64),
                ('noise_type', ctypes.c_int), ('reserved', ctypes.c_int)]


# name -> (restype, argtypes); exactly the symbols include/drp.h declares
SIGNATURES = {
    'drp_create': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    'drp_destroy': (None, [ctypes.c_void_p]),
    'drp_last_error': (ctypes.c_char_p, [ctypes.c_void_p]),
    'drp_sync': (ctypes.c_int, [ctypes.c_void_p]),
    'drp_set_engine': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'drp_device_info': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t,
                                       ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_size_t)]),
    'drp_load_weights': (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_size_t, ctypes.c_float]),
    'drp_set_camera': (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_float, c_float_p]),
    'drp_set_goal': (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_int, ctypes.c_int,
                                    c_float_p, ctypes.c_int]),
    'drp_distance_transform': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, c_float_p]),
    'drp_set_goal_image': (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, c_float_p, c_float_p,
                                          ctypes.POINTER(ctypes.c_int)]),
    'drp_gen_s_delta': (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_float_p, ctypes.c_int,
                                       ctypes.c_int, c_float_p]),
    'drp_build_graph': (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_float_p, ctypes.c_int,
                                       ctypes.c_int, c_int16_p, c_uint8_p]),
    'drp_step': (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                ctypes.c_int, ctypes.c_int, c_float_p]),
    'drp_forward': (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                   c_int16_p, c_uint8_p, ctypes.c_int, ctypes.c_int, c_float_p]),
    'drp_rollout': (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, ctypes.c_int,
                                   ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_int, c_float_p,
                                   c_float_p]),
    'drp_reward': (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, c_float_p]),
    'drp_mpc_begin': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(MpcParams), c_float_p,
                                     c_float_p, c_float_p, c_double_p]),
    'drp_mpc_sample': (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_uint64]),
    'drp_mpc_set_actions': (ctypes.c_int, [ctypes.c_void_p, c_float_p]),
    'drp_mpc_rollout': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'drp_mpc_partials': (ctypes.c_int, [ctypes.c_void_p, c_double_p]),
    'drp_mpc_update': (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_int, c_double_p]),
    'drp_mpc_update_device': (ctypes.c_int, [ctypes.c_void_p]),
    'drp_mpc_elite': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p]),
    'drp_mpc_update_elite': (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_int, ctypes.c_int, c_double_p]),
    'drp_mpc_update_elite_device': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'drp_mpc_get': (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                   c_double_p]),
    'drp_mpc_fetch_async': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'drp_mpc_wait': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_float_p, c_float_p]),
    'drp_fps': (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.POINTER(ctypes.c_int32), c_float_p]),
    'drp_train_begin': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double]),
    'drp_train_step': (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, ctypes.POINTER(ctypes.c_int32),
                                      c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, c_float_p]),
    'drp_train_set_lr': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double]),
    'drp_get_weights': (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_size_t]),
    'drp_depth2fgpcd': (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_int,
                                       ctypes.c_int, c_double_p, c_double_p, ctypes.c_int,
                                       ctypes.POINTER(ctypes.c_int)]),
    'drp_downsample_pcd': (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_int, ctypes.c_double, c_double_p,
                                          ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    'drp_fps_pcd': (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_int32), ctypes.c_uint64, c_float_p, c_double_p]),
    'drp_fps_rad': (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int)]),
    'drp_recenter': (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_int,
                                    c_double_p, c_float_p]),
    'drp_obs2ptcl': (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                    c_double_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int32),
                                    ctypes.c_uint64, c_double_p, c_double_p, ctypes.POINTER(ctypes.c_int),
                                    ctypes.POINTER(ctypes.c_int)]),
    'drp_gd_begin': (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int,
                                    c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_float_p, c_float_p]),
    'drp_gd_grad': (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p]),
    'drp_gd_step': (ctypes.c_int, [ctypes.c_void_p, c_float_p]),
    'drp_gd_get': (ctypes.c_int, [ctypes.c_void_p, c_float_p]),
    'drp_gd_step_async': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'drp_gd_wait': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_float_p, c_float_p]),
    'drp_comm_unique_id': (ctypes.c_int, [ctypes.c_char_p]),
    'drp_comm_init': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]),
    'drp_comm_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'drp_comm_info': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                     ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.c_size_t]),
    'drp_debug_stall': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'drp_comm_allgather': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'drp_probe_begin': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p]),
    'drp_probe_read': (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.POINTER(ctypes.c_long)]),
    'drp_probe_work': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]),
    'drp_dispatch_reset': (ctypes.c_int, [ctypes.c_void_p]),
    'drp_last_dispatch': (ctypes.c_long, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]),
    'drp_dispatch_variants': (ctypes.c_long, [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]),
    'drp_range_info': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), c_double_p, c_double_p,
                                      ctypes.POINTER(ctypes.c_int)]),
    'drp_debug_fetch': (ctypes.c_long, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p,
                                        ctypes.c_size_t]),
}

_lib = None


DRP_ERANGE = -6


class DrpError(RuntimeError):
    pass


class DrpRangeError(DrpError):
    """DRP_ERANGE: weights or inputs outside the range the split-fp16 relation encoder is scaled for (include/drp.h)."""


def load():
    """Load libdrp.so and bind every declared symbol.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DrpError('HIP extension %s is missing: run `python -c "import __graft_entry__ as g; '
                       'g.build()"` (there is no CPU fallback)' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib

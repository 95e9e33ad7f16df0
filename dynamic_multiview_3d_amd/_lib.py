"""ctypes binding of libmv3d_hip.so (include/mv3d_hip.h).

The HIP library IS the product path: there is no CPU or torch fallback.  `lib()` raises if the
shared object is missing or cannot be loaded.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MV3D_LIB") or os.path.join(HERE, "libmv3d_hip.so")      # MV3D_LIB: an experiment build (tools/build_variant.py)

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3
ACT_BY_NAME = {None: ACT_NONE, 'none': ACT_NONE, 'lrelu': ACT_LRELU, 'relu': ACT_RELU, 'tanh': ACT_TANH}
F32 = 0


class ConvGeom(C.Structure):
    """mv3d_conv_geom"""
    _fields_ = [(n, C.c_int32) for n in
                ("N", "H", "W", "C", "Ho", "Wo", "K", "kh", "kw", "sh", "sw", "img_ld", "feat_ld", "dtype")]


class Epilogue(C.Structure):
    """mv3d_epilogue"""
    _fields_ = [("bias", C.c_void_p), ("act", C.c_int32), ("leak", C.c_float), ("gmask_act", C.c_int32),
                ("gmask_leak", C.c_float), ("gmask_ref", C.c_void_p), ("gmask_ld", C.c_int32)]


class FcChainLayer(C.Structure):
    """mv3d_fc_chain_layer"""
    _fields_ = [("M", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("y_ld", C.c_int32), ("out", C.c_int32),
                ("act", C.c_int32), ("leak", C.c_float)]


class FcChain(C.Structure):
    """mv3d_fc_chain"""
    _fields_ = [("B", C.c_int32), ("nlayers", C.c_int32), ("in_", C.c_int32), ("x_ld", C.c_int32), ("x", C.c_void_p),
                ("l", FcChainLayer * 4)]


class Mv3dError(RuntimeError):
    pass


_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
_G, _E = C.POINTER(ConvGeom), C.POINTER(Epilogue)

# name -> argtypes; every one of these returns int status (see include/mv3d_hip.h)
STATUS_FUNCS = {
    "mv3d_conv2d_fwd": [_G, _vp, _vp, _vp, _E, _vp, _sz, _vp],
    "mv3d_conv2d_dgrad": [_G, _vp, _vp, _vp, _E, _vp, _sz, _vp],
    "mv3d_conv2d_wgrad": [_G, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mv3d_deconv2d_fwd": [_G, _vp, _vp, _vp, _E, _vp, _sz, _vp],
    "mv3d_deconv2d_dgrad": [_G, _vp, _vp, _vp, _E, _vp, _sz, _vp],
    "mv3d_deconv2d_wgrad": [_G, _vp, _vp, _vp, _vp, _sz, _vp],
    "mv3d_fc_fwd": [_i, _i, _i, _vp, _i, _vp, _vp, _i, _E, _vp, _sz, _vp],
    "mv3d_fc_dgrad": [_i, _i, _i, _vp, _i, _vp, _vp, _i, _E, _vp, _sz, _vp],
    "mv3d_fc_wgrad": [_i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp],
    "mv3d_fc_wgrad_dgrad": [_i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _E, _vp, _sz, _vp],
    "mv3d_fc_chain_fwd": [C.POINTER(FcChain), _vp],
    "mv3d_act_fwd": [_i64, _i, _vp, _i, _vp, _i, _i, _f, _vp],
    "mv3d_act_bwd": [_i64, _i, _vp, _i, _vp, _i, _vp, _i, _i, _f, _vp],
    "mv3d_copy2d": [_i64, _i, _vp, _i64, _i64, _vp, _i64, _i, _vp],
    "mv3d_group_sum": [_i64, _i, _i, _vp, _i64, _vp, _i64, _vp],
    "mv3d_warp_resample_fwd": [_i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp],
    "mv3d_warp_resample_bwd": [_i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp],
    "mv3d_warp_resample_loss": [_i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _i, _f, _vp, _vp, _vp, _i, _vp, _vp],
    "mv3d_pixel_loss": [_i64, _i, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp],
    "mv3d_loss_overwrite_next": [],
    "mv3d_pixel_loss_strided": [_i64, _i, _vp, _i, _vp, _i, _f, _vp, _i, _i, _f, _vp, _vp, _i, _vp],
    "mv3d_tfrecord_open": [C.c_char_p, _i, C.POINTER(_vp)],
    "mv3d_tfrecord_read": [_vp, _i, _i, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(_vp), C.POINTER(C.c_int)],
    "mv3d_u8_to_unit_f32": [_i64, _vp, _vp, _vp],
    "mv3d_fill": [_vp, _i64, _f, _vp],
    "mv3d_adam_step": [_i64, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _f, _f, _vp],
    "mv3d_adam_step_dev": [_i64, _vp, _vp, _vp, _vp, _vp, _i, C.POINTER(_i64), C.POINTER(_i64), _vp],
    "mv3d_adam_advance": [_vp, _vp],
    "mv3d_fc_wgrad_adam": [_i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "mv3d_grad_finalize_begin": [],
    "mv3d_grad_finalize_add": [_vp, _i64],
    "mv3d_grad_finalize_commit": [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp],
    "mv3d_grad_finalize_abort": [],
    "mv3d_comm_unique_id": [_vp],
    "mv3d_comm_init": [C.POINTER(_vp), _i, _i, _vp],
    "mv3d_comm_destroy": [_vp],
    "mv3d_comm_allreduce_sum": [_vp, _vp, _i64, _vp],
    "mv3d_comm_reduce_scatter_sum": [_vp, _vp, _vp, _i64, _vp],
    "mv3d_comm_allgather": [_vp, _vp, _vp, _i64, _vp],
    "mv3d_ipc_export": [_vp, _vp, C.POINTER(_i64)],
    "mv3d_ipc_open": [_vp, C.POINTER(_vp)],
    "mv3d_ipc_close": [_vp],
    "mv3d_mesh_reduce_sum": [C.POINTER(_vp), _i, _vp, _i64, _vp],
    "mv3d_mesh_copy": [_vp, _vp, _i64, _vp],
    "mv3d_filter_cache_bind": [_G, _i, _vp, _vp, _sz],
    "mv3d_filter_cache_commit": [_vp, _sz, _vp],
    "mv3d_filter_cache_refresh": [_vp],
    "mv3d_filter_cache_clear": [],
    "mv3d_plan_begin": [_vp],
    "mv3d_plan_end": [],
    "mv3d_plan_run": [_vp, _vp],
    "mv3d_plan_run_range": [_vp, _i, _i, _vp],
    "mv3d_plan_run_range2": [_vp, _i, _i, _vp, _vp],
    "mv3d_plan_run_range_multi": [_vp, _i, _i, _vp, C.POINTER(_vp), _i, _i],
    "mv3d_plan_run_side": [_vp, _i, _vp],
    "mv3d_plan_side": [_i],
    "mv3d_plan_profile": [_vp, _i],
    "mv3d_plan_profile_collect": [_vp],
    "mv3d_plan_profile_select": [_vp, C.c_char_p],
    "mv3d_plan_profile_reset": [_vp],
    "mv3d_plan_op_info": [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_double),
                          C.POINTER(C.c_double), C.POINTER(C.c_int)],
}
OTHER_FUNCS = {
    "mv3d_version": (C.c_char_p, []),
    "mv3d_last_error": (C.c_char_p, []),
    "mv3d_conv_workspace_bytes": (_sz, [_G]),
    "mv3d_conv_wgrad_workspace_bytes": (_sz, [_G]),
    "mv3d_fc_workspace_bytes": (_sz, [_i, _i, _i]),
    "mv3d_crc32c": (C.c_uint32, [_vp, _sz]),
    "mv3d_set_diagnostics": (C.c_int, [_i]),
    "mv3d_set_wgrad_cus": (C.c_int, [_i]),
    "mv3d_tfrecord_close": (None, [_vp]),
    "mv3d_filter_prepared_bytes": (_sz, [_G, _i]),
    "mv3d_filter_cache_table_bytes": (_sz, []),
    "mv3d_grad_finalize_table_bytes": (_sz, []),
    "mv3d_plan_create": (_vp, []),
    "mv3d_plan_destroy": (None, [_vp]),
    "mv3d_plan_size": (_i, [_vp]),
    "mv3d_comm_available": (_i, []),
    "mv3d_fc_wgrad_adam_supported": (_i, [_i, _i, _i, _i, _i]),
    "mv3d_debug_cconv_stamps": (_i, [_vp, _sz]),
    "mv3d_debug_cwgrad_stamps": (_i, [_vp, _sz]),
    "mv3d_debug_band_stamps": (_i, [_vp, _sz]),
}
EXPORTS = sorted(list(STATUS_FUNCS) + list(OTHER_FUNCS))

_lib = None


class _Checked:
    """Callable that raises Mv3dError with mv3d_last_error() on a negative status."""

    def __init__(self, name, fn, dll):
        self.name, self.fn, self.dll = name, fn, dll

    def __call__(self, *args):
        rc = self.fn(*args)
        if rc != 0:
            raise Mv3dError("%s failed (%d): %s" % (self.name, rc, self.dll.mv3d_last_error().decode()))
        return rc


class _Lib:
    def __init__(self, path):
        if not os.path.exists(path):
            raise Mv3dError("%s not found: build it with `python -m dynamic_multiview_3d_amd.build` "
                            "(hipcc --offload-arch=gfx950); there is no fallback path" % path)
        # torch must be imported BEFORE the dlopen: it ships its own libamdhip64, and the library has to bind
        # to that already-loaded HIP runtime (two runtimes in one process = "no ROCm-capable device").
        import torch  # noqa: F401
        self.dll = C.CDLL(path)
        for name, (res, args) in OTHER_FUNCS.items():
            fn = getattr(self.dll, name)
            fn.restype, fn.argtypes = res, args
            if name[5:] != 'last_error':
                setattr(self, name[5:], fn)
        for name, args in STATUS_FUNCS.items():
            fn = getattr(self.dll, name)
            fn.restype, fn.argtypes = C.c_int, args
            setattr(self, name[5:], _Checked(name, fn, self.dll))
            setattr(self, "raw_" + name[5:], fn)

    def last_error(self):
        return self.dll.mv3d_last_error().decode()


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib(LIB_PATH)
    return _lib


def plan_ops(plan):
    """[(kernel label, algorithmic flops, algorithmic bytes, total ms, timed runs)] of a plan."""
    l = lib()
    out = []
    for i in range(l.plan_size(plan)):
        name, fl, by, ms, runs = C.c_char_p(), C.c_double(), C.c_double(), C.c_double(), C.c_int()
        l.plan_op_info(plan, i, C.byref(name), C.byref(fl), C.byref(by), C.byref(ms), C.byref(runs))
        out.append((name.value.decode(), fl.value, by.value, ms.value, runs.value))
    return out


def epilogue(bias=None, act=ACT_NONE, leak=0.2, gmask_act=ACT_NONE, gmask_leak=0.2, gmask_ref=None, gmask_ld=0):
    return Epilogue(bias, act, leak, gmask_act, gmask_leak, gmask_ref, gmask_ld)


def conv_geom(N, H, W, Cc, K, kh, kw, sh, sw, img_ld=None, feat_ld=None):
    Ho, Wo = -(-H // sh), -(-W // sw)
    return ConvGeom(N, H, W, Cc, Ho, Wo, K, kh, kw, sh, sw, Cc if img_ld is None else img_ld,
                    K if feat_ld is None else feat_ld, F32)

"""ctypes binding of libpdecnn_hip.so (C ABI in include/pdecnn.h).

There is no CPU fallback and no routing through ``oracle/``: if the HIP library is
missing or a call fails, this raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# PDECNN_LIB: developer override (an alternative build of the same library, e.g. make WAVES=4)
LIB_PATH = os.environ.get("PDECNN_LIB") or os.path.join(HERE, "lib", "libpdecnn_hip.so")

PDE_MAX_SWEEPS = 96
PDE_MAX_N = 32
PDE_IO_F32, PDE_IO_BF16 = 0, 1
PDE_AXIS_X, PDE_AXIS_Y = 0, 1

ERRORS = {
    -1: "PDE_E_BADARG (null pointer, bad dimension or enum)",
    -2: "PDE_E_UNSUPPORTED_N (line length outside [2, 128], or a per-step / one-launch entry point at a line length "
        "without fused kernels: those exist for multiples of 4 in [8, 32])",
    -3: "PDE_E_TOO_MANY_SWEEPS",
    -4: "PDE_E_LAUNCH (HIP launch failed)",
    -5: "PDE_E_WORKSPACE (workspace too small or misaligned)",
}


class PdeSweep(C.Structure):
    _fields_ = [("axis", C.c_int32), ("delta", C.c_float), ("h2", C.c_float), ("t", C.c_float)]


class PdeAdiDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("C", C.c_int32), ("N", C.c_int32), ("io_dtype", C.c_int32),
                ("num_sweeps", C.c_int32), ("smooth3", C.c_int32), ("has_clamp_max", C.c_int32),
                ("clamp_max", C.c_float), ("eps", C.c_float), ("sweep", PdeSweep * PDE_MAX_SWEEPS)]


class PdeSmallLayer(C.Structure):
    """One of the layers that share an input in pde_adi_multi_* (include/pdecnn.h)."""
    _fields_ = [("desc", C.POINTER(PdeAdiDesc)), ("sweeps_per_step", C.c_int32), ("mode", C.c_int32),
                ("M", C.c_void_p), ("skip_weight", C.c_void_p),
                ("alpha_base", C.c_void_p), ("beta_base", C.c_void_p), ("alpha_slope", C.c_void_p), ("beta_slope", C.c_void_p),
                ("weight", C.c_float), ("weight_ptr", C.c_void_p), ("states", C.c_void_p), ("plane_sums", C.c_void_p),
                ("steps_workspace", C.c_void_p), ("steps_workspace_bytes", C.c_size_t),
                ("kappa_max", C.c_void_p), ("kappa_max_host", C.c_void_p),
                ("gys", C.c_void_p), ("g_plane_sums", C.c_void_p), ("ckpt_mask", C.POINTER(C.c_uint64)),
                ("g_alpha_base", C.c_void_p), ("g_beta_base", C.c_void_p), ("g_alpha_slope", C.c_void_p),
                ("g_beta_slope", C.c_void_p), ("gM", C.c_void_p), ("g_skip_weight", C.c_void_p), ("g_weight", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class PdeError(RuntimeError):
    pass


_vp, _fp, _sz, _i32, _f32 = C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_float
_D = C.POINTER(PdeAdiDesc)

# name -> (restype, argtypes): must list every symbol include/pdecnn.h declares
SIGNATURES = {
    "pde_adi_line_length_path": (C.c_int, [_i32]),
    "pde_adi_backward_kernel": (C.c_int, [C.POINTER(PdeAdiDesc), _i32]),
    "pde_adi_forward_workspace_bytes": (_sz, [_D]),
    "pde_adi_backward_workspace_bytes": (_sz, [_D, _i32]),
    "pde_adi_forward": (C.c_int, [_D, _vp, _vp, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _vp, _sz, _vp]),
    "pde_adi_backward": (C.c_int, [_D, _vp, _vp, _vp, C.POINTER(C.c_uint64), _vp, _fp, _fp, _fp, _fp,
                                   _fp, _fp, _fp, _fp, _vp, _vp, _sz, _vp]),
    "pde_adi_kappa_max": (C.c_int, [_D, _fp, _fp, _fp, _fp, _fp, _vp]),
    "pde_adi_steps_workspace_bytes": (_sz, [_D, _i32]),
    "pde_adi_factor_steps": (C.c_int, [_D, _i32, _fp, _fp, _fp, _fp, _fp, _vp, _sz, _vp]),
    "pde_adi_forward_step": (C.c_int, [_D, _i32, _i32, _vp, _vp, _vp, _vp]),
    "pde_adi_backward_step_workspace_bytes": (_sz, [_D, _i32, _i32]),
    "pde_adi_backward_step": (C.c_int, [_D, _i32, _i32, _vp, _vp, _vp, C.POINTER(C.c_uint64), _vp, _vp, _vp, _sz,
                                        _i32, _vp]),
    "pde_adi_param_grads": (C.c_int, [_D, _i32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _vp, _vp]),
    "pde_adi_mixed_forward": (C.c_int, [_D, _i32, _i32, _vp, _vp, _vp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _vp, _sz, _vp]),
    "pde_adi_mixed_backward_workspace_bytes": (_sz, [_D, _i32, _i32]),
    "pde_adi_mixed_backward": (C.c_int, [_D, _i32, _i32, _vp, _vp, _vp, _vp, _fp, C.POINTER(C.c_uint64), _vp, _fp, _fp, _fp, _fp,
                                         _fp, _fp, _fp, _fp, _fp, _vp, _vp, _sz, _vp]),
    "pde_adi_small_supported": (C.c_int, [_D, _i32]),
    "pde_adi_mixed_one_launch": (C.c_int, [_D, _i32]),
    "pde_adi_small_forward": (C.c_int, [_D, _i32, _i32, _vp, _vp, _vp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _vp, _sz,
                                        _vp]),
    "pde_adi_small_backward_workspace_bytes": (_sz, [_D, _i32, _i32]),
    "pde_adi_small_backward": (C.c_int, [_D, _i32, _i32, _vp, _vp, _vp, _fp, _fp, C.POINTER(C.c_uint64), _vp,
                                         _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _vp, _sz, _vp]),
    "pde_adi_multi_forward": (C.c_int, [_i32, C.POINTER(PdeSmallLayer), _vp, _vp, _vp, _vp]),
    "pde_adi_multi_backward": (C.c_int, [_i32, C.POINTER(PdeSmallLayer), _vp, _vp, _vp, _vp]),
    "pde_gate_combine_forward": (C.c_int, [_i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _fp,
                                           _vp, _vp]),
    "pde_gate_combine_backward": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                            _fp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _vp]),
    "pde_bn_pool_workspace_bytes": (_sz, [_i32, _i32]),
    "pde_bn_pool_forward": (C.c_int, [_i32, _i32, _i32, _fp, _fp, _fp, _f32, _i32, _f32, _fp, _fp, _fp, _fp, _fp, _vp, _vp, _sz,
                                      _vp]),
    "pde_bn_pool_backward": (C.c_int, [_i32, _i32, _i32, _fp, _fp, _fp, _fp, _vp, _fp, _i32, _fp, _fp, _fp, _vp, _sz, _vp]),
    "pde_channel_mix_forward": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _fp, _vp, _vp]),
    "pde_channel_mix_backward_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "pde_channel_mix_backward": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _fp, _vp, _fp, _vp, _sz, _vp]),
    "pde_channel_mix_backward_steps": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _fp, _vp, _fp, _vp, _sz, _i32, _i32,
                                                 _vp]),
    "pde_skip_blend_forward": (C.c_int, [C.c_int64, _i32, _vp, _vp, _fp, _vp, _vp]),
    "pde_skip_blend_backward_workspace_bytes": (_sz, [C.c_int64]),
    "pde_skip_blend_backward": (C.c_int, [C.c_int64, _i32, _vp, _vp, _vp, _fp, _vp, _vp, _fp, _vp, _sz, _vp]),
    "pde_explicit5_forward": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _fp, _fp, _f32, _f32, _f32, _f32, _i32, _vp,
                                        _vp, _vp]),
    "pde_explicit5_backward_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "pde_explicit5_backward": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _fp, _fp, _f32, _f32, _f32, _f32,
                                         _i32, _vp, _fp, _fp, _vp, _sz, _vp]),
    "pde_jacobi_forward": (C.c_int, [_i32, _i32, _i32, _i32, _fp, _fp, _fp, _fp, _vp]),
    "pde_jacobi_backward_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "pde_jacobi_backward": (C.c_int, [_i32, _i32, _i32, _i32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _sz, _vp]),
    "pde_sym_layer_supported": (C.c_int, [_i32, _i32]),
    "pde_sym_layer_workspace_bytes": (_sz, [_i32, _i32]),
    "pde_sym_layer_forward": (C.c_int, [_i32, _i32, _i32, _i32, _fp, _fp, _fp, _fp, _fp, _fp, _f32, _f32, _fp, _f32,
                                        _fp, _fp, _fp, _fp, _fp, _vp, _sz, _vp]),
    "pde_sym_layer_backward": (C.c_int, [_i32, _i32, _i32, _i32, _fp, _f32, _fp, _fp, _fp, _fp, _fp, _fp, _fp,
                                         _fp, _fp, _fp, _fp, _fp, _vp, _sz, _vp]),
    "pde_timing_enable": (C.c_int, [_i32]),
    "pde_timing_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                  C.POINTER(C.c_int64)]),
    "pde_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load the shared library once; raise if it is not there (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise PdeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C cnn-with-pde_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)         # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


#: the native host path of the launch-bound layer calls (csrc/host_ext.cpp: torch C++ autograd nodes over the same C ABI).
#: Host glue only — the same launches as the ctypes path, without the interpreter between them.  PDE_HOST_EXT=0 keeps
#: every call on the ctypes path; a developer override of the library (PDECNN_LIB) does too, since the extension is linked
#: against the in-tree build.
HOST_EXT_PATH = os.path.join(HERE, "lib", "_pdecnn_host.so")
_host = None


def host_ext():
    """The extension module, or None when it is switched off.  Raises when it should be there and is not."""
    global _host
    if _host is None:
        if os.environ.get("PDE_HOST_EXT", "1") == "0" or os.environ.get("PDECNN_LIB"):
            _host = False
        else:
            if not os.path.isfile(HOST_EXT_PATH):
                raise PdeError(f"{HOST_EXT_PATH} not found: build it with `make -C cnn-with-pde_amd/csrc` "
                               "(or set PDE_HOST_EXT=0 to run every call through the ctypes path)")
            load()                                  # libpdecnn_hip.so first: the extension resolves its symbols from it
            import importlib.util
            import torch  # noqa: F401  (libtorch must be loaded before the extension)
            spec = importlib.util.spec_from_file_location("_pdecnn_host", HOST_EXT_PATH)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mod.set_error_class(PdeError)
            _host = mod
    return _host or None


def check(rc, what):
    if rc != 0:
        raise PdeError(f"{what} failed: {ERRORS.get(rc, rc)}")

"""ctypes binding of libraygun_hip.so (the C-ABI declared in include/raygun_hip.h).

The product path has no fallback: if the library is missing or a call fails, a RuntimeError
carrying rg_last_error_string() is raised (mirrors the RuntimeErrors the reference raises,
networks/UNet.py:85,233,575).
"""
import ctypes as C
import os

from . import _debug

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libraygun_hip.so")
LIB_PATHS = {"bf16": LIB_PATH, "f16": os.path.join(_HERE, "lib", "libraygun_hip_f16.so")}  # same sources, 16-bit storage type differs

RG_BF16, RG_F32, RG_U8, RG_I32, RG_U64, RG_F64, RG_F16 = range(7)
ACT = {"none": 0, "relu": 1, "leaky": 2, "sigmoid": 3, "tanh": 4}


class rg_tensor(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("dtype", C.c_int32), ("rank", C.c_int32),
                ("sizes", C.c_int64 * 6), ("strides", C.c_int64 * 6)]


class rg_pack_params(C.Structure):
    _fields_ = [("dim0", C.c_int32), ("dim1", C.c_int32), ("K", C.c_int32 * 3), ("swap", C.c_int32),
                ("k0", C.c_int32 * 3), ("kstep", C.c_int32 * 3), ("ntap", C.c_int32 * 3), ("red_align", C.c_int32), ("src_channels_last", C.c_int32)]


class rg_conv_params(C.Structure):
    _fields_ = [("R", C.c_int32 * 3), ("T", C.c_int32 * 3), ("rstride", C.c_int32 * 3),
                ("tbase", C.c_int32 * 3), ("tstep", C.c_int32 * 3), ("ostride", C.c_int32 * 3),
                ("ooffset", C.c_int32 * 3), ("cout", C.c_int32), ("act", C.c_int32), ("slope", C.c_float)]


class rg_mws_params(C.Structure):
    _fields_ = [("shape", C.c_int32 * 3), ("n_offsets", C.c_int32), ("n_attractive", C.c_int32),
                ("strides", C.c_int32 * 3), ("randomize_strides", C.c_int32), ("invert_repulsive_weights", C.c_int32),
                ("bias_cut", C.c_float), ("origin", C.c_int32 * 3)]


_P = C.POINTER
_T = _P(rg_tensor)
_I3 = C.c_int32 * 3
_SIGS = {
    "rg_version": (C.c_int, []),
    "rg_last_error_string": (C.c_char_p, []),
    "rg_last_kernel_name": (C.c_char_p, []),
    "rg_hip_status": (C.c_int, []),
    "rg_set_option": (C.c_int, [C.c_char_p, C.c_int32]),
    "rg_get_option": (C.c_int32, [C.c_char_p]),
    "rg_copy_convert": (C.c_int, [_T, _T, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "rg_copy_convert_store": (C.c_int, [_T, _T, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "rg_nonfinite_check": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "rg_conv3d_kalign": (C.c_int, [C.c_int, C.c_int]),
    "rg_pack_weights_bytes": (C.c_int64, [_P(rg_pack_params)]),
    "rg_pack_weights": (C.c_int, [C.c_void_p, C.c_void_p, _P(rg_pack_params), C.c_void_p]),
    "rg_conv3d_igemm": (C.c_int, [_T, C.c_void_p, C.c_void_p, _T, _P(rg_conv_params), C.c_void_p]),
    "rg_conv3d_igemm_d2s": (C.c_int, [_T, C.c_void_p, C.c_void_p, _T, _P(rg_conv_params), C.c_int32, C.c_void_p]),
    "rg_conv3d_igemm_workspace_bytes": (C.c_int64, [_T, _T, _P(rg_conv_params)]),
    "rg_conv3d_igemm2": (C.c_int, [_T, C.c_void_p, C.c_void_p, _T, _P(rg_conv_params), C.c_void_p, C.c_size_t, C.c_void_p]),
    "rg_wino_weights_rows": (C.c_int64, [C.c_int32]),
    "rg_wino_weights_red": (C.c_int64, [C.c_int32]),
    "rg_wino_weights": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rg_wino_input": (C.c_int, [_T, C.c_void_p, C.c_int64, _I3, _I3, C.c_void_p]),
    "rg_wino_output": (C.c_int, [C.c_void_p, C.c_int64, _I3, _T, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "rg_gemm_batched_tile_n": (C.c_int32, [C.c_int32]),
    "rg_gemm_batched": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                  C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "rg_conv3d_wgrad_workspace_bytes": (C.c_int64, [_T, _T, _P(rg_conv_params), _P(C.c_int32)]),
    "rg_conv3d_wgrad2": (C.c_int, [_T, _T, C.c_void_p, _P(rg_conv_params), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rg_conv3d_wgrad": (C.c_int, [_T, _T, C.c_void_p, _P(rg_conv_params), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rg_act_bwd_bias": (C.c_int, [_T, _T, _T, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rg_act_bwd_bias_workspace_bytes": (C.c_int64, [_T]),
    "rg_maxpool3d_fwd": (C.c_int, [_T, _T, _I3, C.c_void_p]),
    "rg_maxpool3d_bwd": (C.c_int, [_T, _T, _T, _I3, C.c_void_p]),
    "rg_upcat_fwd": (C.c_int, [_T, _T, _T, _I3, _I3, _I3, C.c_void_p]),
    "rg_upcat_bwd": (C.c_int, [_T, _T, _T, _I3, _I3, _I3, C.c_void_p]),
    "rg_instnorm_workspace_bytes": (C.c_int64, [_T]),
    "rg_instnorm_stats": (C.c_int, [_T, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rg_instnorm_apply": (C.c_int, [_T, C.c_void_p, _T, _T, C.c_int, C.c_float, C.c_void_p]),
    "rg_instnorm_bwd": (C.c_int, [_T, C.c_void_p, _T, _T, _T, C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rg_norm_apply_affine": (C.c_int, [_T, C.c_void_p, C.c_void_p, _T, _T, C.c_int, C.c_float, C.c_void_p]),
    "rg_norm_bwd_affine": (C.c_int, [_T, C.c_void_p, C.c_void_p, C.c_int, _T, _T, _T, C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rg_pad3d_fwd": (C.c_int, [_T, _T, _I3, C.c_int, C.c_void_p]),
    "rg_pad3d_bwd": (C.c_int, [_T, _T, _I3, C.c_int, C.c_void_p]),
    "rg_add": (C.c_int, [_T, _T, _T, C.c_void_p]),
    "rg_diag_spread": (C.c_int, [_T, _T, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rg_d2s_add_act": (C.c_int, [_T, _T, _T, C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    "rg_d2s_add_act_bwd": (C.c_int, [_T, _T, _T, _T, C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    "rg_add_pad": (C.c_int, [_T, _T, _T, C.c_void_p, C.c_void_p]),
    "rg_subpixel_weights": (C.c_int, [_T, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rg_subpixel_weights_bwd": (C.c_int, [C.c_void_p, C.c_void_p, _T, C.c_int, C.c_void_p]),
    "rg_diag_sum": (C.c_int, [_T, _T, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rg_loss_fwd_bwd": (C.c_int, [C.c_int, _T, _T, _T, C.c_float, C.c_float, _T, C.c_void_p, C.c_void_p]),
    "rg_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float,
                               C.c_float, C.c_float, C.c_int32, C.c_float, C.c_void_p]),
    "rg_profile_enable": (None, [C.c_int]),
    "rg_profile_drain": (C.c_int, [C.c_void_p, C.c_int]),
    "rg_adam_step_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float,
                               C.c_float, C.c_float, C.c_int32, C.c_float, C.c_void_p, _P(rg_pack_params), C.c_void_p]),
    "rg_adam_step_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_float, C.c_int32, C.c_float, C.c_void_p]),
    "rg_adam_step_multi_hyp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float,
                                         C.c_float, C.c_float, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "rg_lz4_decompress_block": (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    "rg_lz4_compress_bound": (C.c_int64, [C.c_int64]),
    "rg_lz4_compress_block": (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    "rg_blosc_compress_bound": (C.c_int64, [C.c_int64]),
    "rg_blosc_compress": (C.c_int64, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]),
    "rg_blosc_nbytes": (C.c_int64, [C.c_void_p, C.c_int64]),
    "rg_blosc_decompress": (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    "rg_add_affinities": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, _I3, _I3, _I3, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rg_balance_labels_workspace_bytes": (C.c_int64, []),
    "rg_balance_labels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rg_lsd_targets": (C.c_int, [C.c_void_p, _I3, _I3, _I3, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, _I3, C.c_float * 3, C.c_float * 3,
                                 C.c_void_p, C.c_void_p]),
    "rg_mws_workspace_bytes": (C.c_int64, [_P(rg_mws_params)]),
    "rg_mws_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "rg_mws_prepare_u8": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "rg_mws_segment": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _P(rg_mws_params), C.c_void_p,
                                 C.c_size_t, C.c_void_p]),
    "rg_mws_stats": (C.c_int, [C.c_void_p, _P(rg_mws_params), C.c_void_p, C.c_void_p]),
    "rg_mws_cross_edges": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, _P(rg_mws_params),
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "rg_mws_cross_edges_u8": (C.c_int, [C.c_void_p, C.c_float, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p,
                                        _P(rg_mws_params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "rg_mws_segment_u8": (C.c_int, [C.c_void_p, C.c_float, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, _P(rg_mws_params), C.c_void_p,
                                    C.c_size_t, C.c_void_p]),
    "rg_mws_slab_mutexes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _P(rg_mws_params), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "rg_mws_graph_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int64]),
    "rg_mws_graph": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                               C.c_size_t, C.c_void_p]),
    "rg_mws_relabel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rg_ws_mask": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "rg_ws_edt_workspace_bytes": (C.c_int64, [_I3]),
    "rg_ws_edt": (C.c_int, [C.c_void_p, _I3, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rg_ws_maxima_workspace_bytes": (C.c_int64, [_I3]),
    "rg_ws_maxima": (C.c_int, [C.c_void_p, _I3, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rg_ws_label_workspace_bytes": (C.c_int64, [_I3]),
    "rg_ws_label": (C.c_int, [C.c_void_p, _I3, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rg_ws_flood_workspace_bytes": (C.c_int64, [_I3]),
    "rg_ws_flood": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _I3, C.c_void_p, _P(C.c_int32), C.c_void_p, C.c_size_t, C.c_void_p]),
    "rg_ws_rag_count": (C.c_int, [C.c_void_p, _I3, C.c_void_p, C.c_void_p]),
    "rg_ws_rag_emit": (C.c_int, [C.c_void_p, C.c_void_p, _I3, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "rg_ws_agglomerate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "rg_ws_relabel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
}

DECLARED = sorted(_SIGS)
_F16_ABSENT = {"rg_add_affinities", "rg_balance_labels_workspace_bytes", "rg_balance_labels", "rg_lsd_targets", "rg_mws_workspace_bytes", "rg_mws_prepare", "rg_mws_prepare_u8", "rg_mws_segment", "rg_mws_segment_u8", "rg_mws_cross_edges_u8", "rg_mws_stats", "rg_mws_cross_edges", "rg_mws_slab_mutexes",
               "rg_mws_graph_workspace_bytes", "rg_mws_graph", "rg_mws_relabel"} | {n for n in _SIGS if n.startswith("rg_ws_")}  # no 16-bit data on the watershed paths
_libs = {}
import threading
_tls = threading.local()


def lib(kind="bf16"):
    """Load (once) and return the library whose 16-bit storage type is `kind` ("bf16" | "f16").  Raises RuntimeError if it is
    not built — there is no fallback."""
    l = _libs.get(kind)
    if l is None:
        path = LIB_PATHS[kind]
        if not os.path.exists(path):
            raise RuntimeError(f"raygun_amd: {path} is missing — build it with `python -m raygun_amd.build` "
                               "(hipcc --offload-arch=gfx950). The HIP path has no CPU fallback.")
        import torch  # noqa: F401  (loads torch's libamdhip64 first so both share one HIP runtime)
        l = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            if kind == "f16" and name in _F16_ABSENT:
                continue
            fn = getattr(l, name)  # AttributeError => symbol missing: fail loudly
            fn.restype, fn.argtypes = res, args
        if bool(l.rg_version() >> 16) != (kind == "f16"):
            raise RuntimeError(f"raygun_amd: {path} was not built for {kind} storage")
        _libs[kind] = l
    _tls.last = kind
    return l


_DEBUG_SYNC = bool(os.environ.get("RAYGUN_AMD_DEBUG_SYNC"))  # fault localisation: name every call on stderr, then wait for it


def check(rc, what=""):
    if _debug.ON:
        _debug.verify(f"{what} ({lib(getattr(_tls, 'last', 'bf16')).rg_last_kernel_name().decode()})")
    if _DEBUG_SYNC:
        import sys
        import torch
        sys.stderr.write(f"[rg] {what} {lib(getattr(_tls, 'last', 'bf16')).rg_last_kernel_name().decode()}\n"); sys.stderr.flush()
        torch.cuda.synchronize()
    if rc == 0:
        rc = lib(getattr(_tls, "last", "bf16")).rg_hip_status()   # a kernel launch that failed (bad configuration, LDS size, ...) is silent otherwise
    if rc != 0:
        msg = lib(getattr(_tls, "last", "bf16")).rg_last_error_string().decode("utf-8", "replace")
        raise RuntimeError(f"raygun_amd HIP call failed ({what}, status {rc}): {msg}")


def i3(v):
    return _I3(*[int(a) for a in v])

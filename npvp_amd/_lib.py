"""ctypes binding of libnpvp_hip.so (include/npvp_hip.h).  There is NO fallback: if the
shared object is missing or a call fails, a RuntimeError is raised - the product path never
computes on the CPU or through torch ops in place of a HIP kernel."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnpvp_hip.so")

c_int, c_ll, c_f, c_u, c_p = ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_uint, ctypes.c_void_p

# name -> (restype, argtypes)  - mirrors include/npvp_hip.h one to one
SIGNATURES = {
    "npvp_version": (c_int, []),
    "npvp_last_error": (ctypes.c_char_p, []),
    "npvp_launch_count": (c_ll, []),
    "npvp_dp_unique_id": (c_int, [c_p]),
    "npvp_dp_init": (c_int, [c_int, c_int, c_p]),
    "npvp_dp_world": (c_int, []),
    "npvp_dp_rank": (c_int, []),
    "npvp_dp_allreduce_async": (c_int, [c_p, c_ll, c_p]),          # (size_t n: LP64)
    "npvp_dp_wait": (c_int, [c_p]),
    "npvp_dp_finalize": (c_int, []),
    "npvp_stream_create_low_priority": (c_p, [c_p, c_p]),
    "npvp_stream_destroy": (c_int, [c_p]),
    "npvp_event_create": (c_p, []),
    "npvp_event_record": (c_int, [c_p, c_p]),
    "npvp_event_elapsed_ms": (c_f, [c_p, c_p]),
    "npvp_event_destroy": (c_int, [c_p]),
    "npvp_graph_node_counts": (c_ll, [c_p, c_p, c_p, c_int]),
    "npvp_gemm_workspace_bytes": (c_ll, [c_int, c_int, c_int]),
    "npvp_gemm_kernel_id": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "npvp_gemm_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_int, c_p, c_p,
                              c_p, c_ll, c_f, c_int, c_int, c_int, c_p, c_u, c_f, c_int, c_p, c_p, c_int, c_p, c_p, c_p, c_p, c_p,
                              c_f, c_int, c_int, c_u, c_p, c_ll, c_p]),
    "npvp_amax": (c_int, [c_p, c_ll, c_ll, c_ll, c_p, c_p]),
    "npvp_wgrad_f16_chainable": (c_int, [c_int, c_int, c_int]),
    "npvp_wgrad_f16_chain_workspace_bytes": (c_ll, [c_int, c_int, c_int]),
    "npvp_wgrad_f16_chained": (c_int, [c_int, c_int, c_int, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_int, c_p, c_p, c_p, c_f, c_int, c_int,
                                       c_u, c_p, c_p, c_p, c_p, c_ll, c_p]),
    "npvp_splitk_reduce_job": (c_int, [c_p, c_p]),
    "npvp_splitk_reduce_multi": (c_int, [c_p, c_int, c_p]),
    "npvp_linear_bwd_f16_takes": (c_int, [c_int, c_int, c_int]),
    "npvp_linear_bwd_f16": (c_int, [c_int, c_int, c_int, c_p, c_ll, c_p, c_p, c_p, c_p, c_ll, c_int, c_p, c_p, c_ll, c_f, c_int, c_int, c_int,
                                    c_u, c_p, c_p, c_ll, c_p, c_p, c_ll, c_p, c_p, c_f, c_int, c_int, c_u, c_p, c_p, c_p, c_p, c_ll, c_p]),
    "npvp_split_weight_f16": (c_int, [c_p, c_ll, c_int, c_int, c_p, c_p, c_p, c_p]),
    "npvp_split_weights_f16": (c_int, [c_p, c_int, c_p, c_ll, c_p]),
    "npvp_frame_stats_finalize": (c_int, [c_p, c_int, c_f, c_p, c_p, c_int, c_f, c_p]),
    "npvp_split_weight": (c_int, [c_p, c_ll, c_int, c_int, c_p, c_p, c_p]),
    "npvp_split_weights_batched": (c_int, [c_p, c_int, c_p]),
    "npvp_layernorm_fwd": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_int, c_f, c_int, c_p, c_p]),
    "npvp_layernorm_bwd_workspace_bytes": (c_ll, [c_ll, c_int]),
    "npvp_layernorm_bwd": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_int, c_int, c_p, c_int, c_p, c_p, c_ll, c_p]),
    "npvp_layernorm_bwd_reduce": (c_int, [c_p, c_p, c_p, c_ll, c_int, c_int, c_p]),
    "npvp_layernorm_bwd_reduce_job": (c_int, [c_p, c_p, c_p, c_ll, c_int, c_int, c_p]),
    "npvp_frameln_act_bwd_reduce_job": (c_int, [c_p, c_p, c_p, c_int, c_int, c_int, c_p]),
    "npvp_mlpdw_mid_bwd_reduce_job": (c_int, [c_p, c_p, c_p, c_int, c_int, c_p]),
    "npvp_sum_rows_multi": (c_int, [c_p, c_int, c_p]),
    "npvp_layernorm_nchw_fwd": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_f, c_int, c_p]),
    "npvp_frameln_act_bwd_reduce": (c_int, [c_p, c_p, c_p, c_int, c_int, c_int, c_p]),
    "npvp_frame_stats": (c_int, [c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_f, c_p]),
    "npvp_posfuse_fwd": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_f, c_p, c_p]),
    "npvp_posfuse_instance_fwd": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_f, c_p, c_p]),
    "npvp_posfuse_instance_bwd": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_p]),
    "npvp_ln_posfuse_fwd": (c_int, [c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_f,
                                    c_p, c_p, c_p]),
    "npvp_posfuse_bwd_fused": (c_int, [c_int, c_int, c_int]),
    "npvp_posfuse_bwd": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_p, c_ll, c_p]),
    "npvp_frameln_act_fwd": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_f, c_u, c_f, c_u, c_int, c_p, c_p, c_p]),
    "npvp_frameln_act_bwd_workspace_bytes": (c_ll, [c_int, c_int]),
    "npvp_frameln_act_bwd": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_f, c_u, c_f, c_u, c_int,
                                     c_p, c_int, c_p, c_p, c_ll, c_p]),
    "npvp_dwconv3x3": (c_int, [c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_p]),
    "npvp_dwconv3x3_stats": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_f, c_p, c_ll, c_p]),
    "npvp_mlpdw_mid_fwd": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_f, c_p, c_ll, c_p]),
    "npvp_mlpdw_mid_fwd_parts": (c_int, [c_p, c_p, c_int, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_f, c_p]),
    "npvp_frameln_act_fwd_parts": (c_int, [c_p, c_p, c_int, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_f, c_u, c_f, c_u, c_int,
                                           c_p, c_p, c_p]),
    "npvp_mlpdw_mid_bwd_workspace_bytes": (c_ll, [c_int, c_int]),
    "npvp_mlpdw_mid_bwd": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_p, c_ll, c_p]),
    "npvp_mlpdw_mid_bwd_reduce": (c_int, [c_p, c_p, c_int, c_int, c_int, c_p]),
    "npvp_mlpdw_mid_bwd_reduce_into": (c_int, [c_p, c_p, c_p, c_int, c_int, c_p]),
    "npvp_mlpdw_mid_bwd_n2": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_f, c_u, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                      c_p, c_int, c_int, c_int, c_int, c_int, c_p, c_ll, c_p]),
    "npvp_frameln_act_bwd_pgrad": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_f, c_u, c_f, c_u, c_int,
                                           c_p, c_int, c_p, c_ll, c_p]),
    "npvp_frameln_act_bwd_apply": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_p, c_p, c_p, c_int, c_int, c_int, c_p, c_p, c_ll, c_p]),
    "npvp_im2col3x3": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_p]),
    "npvp_dwconv3x3_wgrad_workspace_bytes": (c_ll, [c_int, c_int]),
    "npvp_dwconv3x3_wgrad": (c_int, [c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_p, c_ll, c_p]),
    "npvp_attn_fwd": (c_int, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                              c_int, c_int, c_int, c_f, c_p, c_u, c_p, c_p]),
    "npvp_attn_bwd": (c_int, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_int, c_int,
                              c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_f, c_p, c_u, c_p, c_p, c_p, c_p]),
    "npvp_drop_apply": (c_int, [c_p, c_p, c_ll, c_int, c_f, c_int, c_int, c_int, c_p, c_u, c_p, c_p]),
    "npvp_transpose": (c_int, [c_p, c_p, c_int, c_int, c_int, c_p]),
    "npvp_dwtb_accumulate": (c_int, [c_p, c_p, c_p, c_int, c_p]),
    "npvp_dwtb_build": (c_int, [c_p, c_p, c_p, c_int, c_p]),
    "npvp_reduce_mid": (c_int, [c_p, c_p, c_int, c_int, c_ll, c_f, c_int, c_p]),
    "npvp_broadcast_mid": (c_int, [c_p, c_p, c_int, c_int, c_ll, c_f, c_p]),
    "npvp_colsum_workspace_bytes": (c_ll, [c_ll, c_int]),
    "npvp_colsum": (c_int, [c_p, c_ll, c_int, c_ll, c_p, c_int, c_p, c_ll, c_p]),
    "npvp_grad_norm_clip": (c_int, [c_p, c_ll, c_f, c_p, c_p, c_ll, c_p]),
    "npvp_l1_mean": (c_int, [c_p, c_p, c_ll, c_f, c_p, c_p, c_ll, c_p]),
    "npvp_l1_mean_bwd": (c_int, [c_p, c_p, c_ll, c_p, c_f, c_p, c_p]),
    "npvp_sum_all": (c_int, [c_p, c_ll, c_p, c_p, c_ll, c_p]),
    "npvp_adamw_step": (c_int, [c_p, c_p, c_p, c_p, c_ll, c_p, c_f, c_f, c_f, c_f, c_p, c_ll, c_ll, c_int, c_p]),
    "npvp_sqdiff_workspace_bytes": (c_ll, [c_int, c_ll]),
    "npvp_sqdiff_per_image": (c_int, [c_p, c_p, c_int, c_ll, c_f, c_f, c_p, c_p, c_ll, c_p]),
    "npvp_ssim_workspace_bytes": (c_ll, [c_int, c_int, c_int, c_int]),
    "npvp_u8hwc_to_f32chw": (c_int, [c_p, c_p, c_ll, c_int, c_int, c_int, c_p, c_p, c_p]),
    "npvp_bias_act": (c_int, [c_p, c_p, c_p, c_p, c_ll, c_ll, c_int, c_int, c_int, c_p]),
    "npvp_act_bwd": (c_int, [c_p, c_p, c_p, c_ll, c_int, c_p]),
    "npvp_ssim_per_image": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_p, c_int, c_p, c_p, c_ll, c_p]),
}

_lib = None


def lib():
    """Load (once) and return the bound library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing - the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `python npvp_amd/build.py`). "
                "npvp_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH, mode=os.RTLD_NOW)        # resolve every symbol now: a broken build fails here
        bound = _Bound()
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
            setattr(bound, name, fn)
        bound._cdll = L
        # The same entry points through C-API wrappers (npvp_amd/_npvp_fast.so, generated by build.py from SIGNATURES): ctypes spends
        # ~0.25 us per argument on its prototype machinery - 7 - 8 ms of a host-bound 8-clip step.  Same functions of the same
        # library, same arguments, same return values; NPVP_FASTCALL=0 keeps ctypes (A/B runs), and so does a missing module.
        bound._fast = 0
        if os.environ.get("NPVP_FASTCALL", "1") == "1":
            try:
                from . import _npvp_fast as F
            except ImportError as e:
                import warnings
                warnings.warn(f"npvp_amd._npvp_fast is not built ({e}): every call goes through ctypes (same kernels, more host time)")
            else:
                for name in SIGNATURES:
                    f = getattr(F, name, None)
                    if f is not None:
                        setattr(bound, name, f)
                        bound._fast += 1
        _lib = bound
    return _lib


class _Bound:
    """the bound entry points of libnpvp_hip.so as plain attributes (C-API wrapper where there is one, ctypes function otherwise)"""


def check(rc, what):
    if rc != 0:
        msg = lib().npvp_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")

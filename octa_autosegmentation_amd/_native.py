"""ctypes binding of liboctahip.so (include/octa_hip.h). Fails loudly when the library is missing."""
import contextlib
import ctypes
import os
import random as _random
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OCTA_HIP_LIB") or os.path.join(HERE, "liboctahip.so")    # OCTA_HIP_LIB: experiment builds (tools/build_sim_variant.py)

_lib = None
_ctxs = {}
_lock = threading.Lock()

c_void_p, c_int, c_double, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t

# symbol -> (restype, argtypes); must list every function include/octa_hip.h declares
SIGNATURES = {
    "octa_abi_version": (c_int, []),
    "octa_last_error": (ctypes.c_char_p, []),
    "octa_ctx_create": (c_int, [c_int, ctypes.POINTER(c_void_p)]),
    "octa_ctx_destroy": (None, [c_void_p]),
    "octa_ctx_scratch_bytes": (c_size_t, [c_void_p]),
    "octa_rasterize_2d": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_double,
                                  c_double, c_void_p, c_void_p]),
    "octa_rasterize_2d_plan": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_double, c_void_p]),
    "octa_rasterize_2d_draw": (c_int, [c_void_p, c_void_p, c_void_p]),
    "octa_raster_prof": (c_int, [c_void_p, c_void_p]),
    "octa_fs_dither": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "octa_voxel_padded_dims": (c_int, [c_void_p, c_void_p]),
    "octa_voxelize_3d": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_int, c_void_p, c_void_p]),
    "octa_edges_read_back": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_void_p, c_void_p]),
    "octa_max_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "octa_bif_native_init": (c_int, [ctypes.c_char_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "octa_bif_native": (None, [c_int, c_void_p, c_void_p, c_void_p]),
    "octa_bif_native_counts": (c_int, [c_void_p]),
    "octa_instnorm_lrelu_fwd": (c_int, [c_void_p] * 7 + [c_int, c_int, ctypes.c_int64, c_int, ctypes.c_float, ctypes.c_float, c_void_p]),
    "octa_instnorm_lrelu_bwd": (c_int, [c_void_p] * 10 + [c_int, c_int, ctypes.c_int64, c_int, ctypes.c_float, c_void_p]),
    "octa_csv_bytes_bound": (ctypes.c_int64, [ctypes.c_int64]),
    "octa_csv_format_edges": (ctypes.c_int64, [c_void_p, ctypes.c_int64, c_void_p, ctypes.c_int64]),
    "octa_csv_write_file": (c_int, [ctypes.c_char_p, c_void_p, ctypes.c_int64]),
    "octa_csv_count_rows": (ctypes.c_int64, [c_void_p, ctypes.c_int64]),
    "octa_csv_parse_edges": (ctypes.c_int64, [c_void_p, ctypes.c_int64, c_void_p, ctypes.c_int64]),
    "octa_py_random_advance": (c_int, [c_void_p, ctypes.c_int64]),
    "octa_png_write_gray8": (c_int, [ctypes.c_char_p, c_void_p, c_int, c_int, c_int]),
    "octa_png_write_bits": (c_int, [ctypes.c_char_p, c_void_p, c_int, c_int, c_int]),
    "octa_write_sample_files": (c_int, [ctypes.c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                        ctypes.c_char_p, ctypes.c_int64, c_int, c_int]),
    "octa_pack_bits": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int, c_void_p]),
    "octa_background_noise": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_void_p, c_void_p, c_void_p]),
    "octa_speckle_brightness": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "octa_sim_create": (c_int, [c_void_p, c_void_p, c_int, ctypes.POINTER(c_void_p)]),
    "octa_sim_create_ex": (c_int, [c_void_p, c_void_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "octa_sim_destroy": (None, [c_void_p]),
    "octa_sim_run": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "octa_sim_run_states": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "octa_sim_np_state": (c_int, [c_void_p, c_int, c_void_p]),
    "octa_thinconv_expand": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, ctypes.c_float, c_void_p]),
    "octa_thinconv_squeeze": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_lrelu_bwd_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, ctypes.c_float, c_void_p]),
    "octa_thinconv_wgrad_scratch_floats": (ctypes.c_longlong, [c_int, c_int, c_int, c_int]),
    "octa_thinconv_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_sim_spans": (c_int, [c_void_p, c_void_p]),
    "octa_sim_edge_offsets": (c_int, [c_void_p, c_void_p, c_void_p]),
    "octa_sim_export_edges": (c_int, [c_void_p, c_void_p]),
    "octa_sim_export_edges_device": (c_int, [c_void_p, c_void_p, c_void_p]),
    "octa_sim_trace": (c_int, [c_void_p, c_void_p]),
    "octa_sim_stats": (c_int, [c_void_p, c_void_p]),
    "octa_sim_timing": (c_int, [c_void_p, c_void_p]),
    "octa_sim_service_stats": (c_int, [c_void_p, c_void_p]),
    "octa_sim_geometry": (c_int, [c_int, c_void_p]),
    "octa_sim_is_large": (c_int, [c_void_p]),
    "octa_conv2d_f32_nchw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 14 + [c_void_p]),
    "octa_convtranspose2x2_f32_nchw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_sim_fields": (c_int, [c_void_p, c_int, c_void_p, ctypes.c_int64, c_void_p, c_void_p, ctypes.c_int64, c_void_p]),
    "octa_instnorm_lrelu_nhwc_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_float, c_void_p]),
    "octa_instnorm_lrelu_nhwc_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_int64, ctypes.c_float, c_void_p]),
    "octa_head1_nhwc_fwd": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_float, ctypes.c_int64, c_int, c_void_p, c_void_p]),
    "octa_head1_nhwc_fwd_b": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int, c_void_p, c_void_p]),
    "octa_pack_conv_weights": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "octa_head1_nhwc_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "octa_conv3x3_nhwc_fwd2": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_conv3x3_nhwc_fwd3": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_instnorm_nhwc_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_int64, ctypes.c_float, c_void_p]),
    "octa_scale_shift_lrelu_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_int64, ctypes.c_float, c_void_p]),
    "octa_resize_bilinear": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "octa_resize_bilinear_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "octa_flip_rot90_rotate": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_int, c_void_p]),
    "octa_remove_small_objects": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.c_uint8, c_void_p, c_void_p]),
    "octa_reflect_pad_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_reflect_pad_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_blur_down_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_blur_down_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_blur_up_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_blur_up_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_conv4x4_nhwc_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_conv4x4_nhwc_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_dice_bce_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, ctypes.c_int64, c_void_p, c_void_p]),
    "octa_dice_bce_finish": (c_int, [c_void_p, c_void_p, c_int, ctypes.c_int64, c_double, c_double, c_void_p, c_void_p]),
    "octa_dice_bce_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, ctypes.c_int64, c_void_p, c_void_p, ctypes.c_float, ctypes.c_float, c_void_p, c_void_p]),
    "octa_conv3x3_c1_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_conv3x3_nhwc_wgrad_acc": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_conv3x3_nhwc_wgrad_pad_acc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_sim_launch_count": (ctypes.c_longlong, []),
    "octa_order_wait_launch": (c_int, [c_void_p, ctypes.c_longlong, c_int, c_int, c_void_p, c_void_p]),
    "octa_conv3x3_c1_fwd2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "octa_conv3x3_c1_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_conv_stat_tiles": (c_int, [c_int, c_int]),
    "octa_conv3x3_nhwc_fwd5": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p, c_void_p]),
    "octa_conv3x3_nhwc_fwd6": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p, c_void_p, c_void_p]),
    "octa_instnorm_lrelu_head1_nhwc_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_float, c_void_p]),
    "octa_instnorm_lrelu_head1_nhwc_fwd_s": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_float, c_void_p, c_int, c_void_p]),
    "octa_instnorm_lrelu_head1_nhwc_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_int64, ctypes.c_float, c_void_p]),
    "octa_instnorm_lrelu_nhwc_fwd_p": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_float, c_void_p, c_int, c_void_p]),
    "octa_conv3x3_s2t_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "octa_conv3x3_nhwc_fwd7": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "octa_instnorm_lrelu_nhwc_fwd_s": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_float, c_void_p, c_int, c_void_p]),
    "octa_conv3x3_nhwc_fwd4": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p]),
    "octa_conv3x3_nhwc_wgrad3": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p]),
    "octa_conv3x3_nhwc_wgrad4": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p]),
    "octa_conv3x3_nhwc_fwd_pad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_conv3x3_nhwc_fwd_pad_s": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "octa_conv3x3_nhwc_wgrad_pad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_conv3x3_nhwc_wgrad2": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_conv3x3_nhwc_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_conv3x3_nhwc_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octa_sim_kat_kd_order": (c_int, [c_void_p, c_void_p, ctypes.c_int64, c_void_p, c_void_p]),
}


class OctaHipError(RuntimeError):
    pass


def lib():
    """Load liboctahip.so once. Raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise OctaHipError(
                        f"{LIB_PATH} is missing: build it with `python -m octa_autosegmentation_amd.build` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
                # torch bundles its own libamdhip64 (same soname as /opt/rocm's): import it first so that
                # liboctahip.so binds to the runtime torch's allocator and streams live in.
                import torch  # noqa: F401
                l = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(l, name)
                    fn.restype = res
                    fn.argtypes = args
                _lib = l
    return _lib


_SYNC_DEBUG = bool(os.environ.get("OCTA_SYNC_DEBUG"))


def check(rc, what):
    if _SYNC_DEBUG and rc == 0:     # development aid: wait for the launch and name it, so that an asynchronous fault has an owner
        import sys
        import torch
        torch.cuda.current_stream().synchronize()
        print(f"[octa-sync] {threading.current_thread().name} {what} ok", file=sys.stderr, flush=True)
    if rc != 0:
        msg = lib().octa_last_error().decode("utf-8", "replace")
        raise OctaHipError(f"{what} failed (rc={rc}): {msg}")


_tls = threading.local()


def new_ctx(device_index=None):
    """A private context (own grow-only scratch); the owner frees it with free_ctx()."""
    import torch
    if not torch.cuda.is_available():
        raise OctaHipError("no ROCm GPU visible to torch; the HIP path cannot run (no CPU fallback)")
    if device_index is None:
        device_index = torch.cuda.current_device()
    out = c_void_p()
    check(lib().octa_ctx_create(int(device_index), ctypes.byref(out)), "octa_ctx_create")
    return out


def free_ctx(h):
    if h is not None:
        lib().octa_ctx_destroy(h)


@contextlib.contextmanager
def use_ctx(h):
    """Route every ctx() lookup of the calling thread to `h` (a pipeline slot keeps its own scratch, whichever
    pool thread happens to run it)."""
    prev = getattr(_tls, "ctx", None)
    _tls.ctx = h
    try:
        yield h
    finally:
        _tls.ctx = prev


def ctx(device_index=None):
    """Context (grow-only device scratch) for one GPU, one per calling THREAD and current STREAM unless use_ctx() is active: an
    octa_ctx must not be used from two threads at once (include/octa_hip.h), pipelines keep several steps in flight from a thread
    pool, and its scratch is stream-ordered (common.h: one context = one stream at a time) -- the autograd thread runs the backward
    nodes of a step that used two streams on those two streams, interleaved (models/gan_seg_model.py)."""
    import torch
    if getattr(_tls, "ctx", None) is not None:
        return _tls.ctx
    if not torch.cuda.is_available():
        raise OctaHipError("no ROCm GPU visible to torch; the HIP path cannot run (no CPU fallback)")
    if device_index is None:
        device_index = torch.cuda.current_device()
    key = (int(device_index), threading.get_ident(), torch.cuda.current_stream(int(device_index)).cuda_stream)
    with _lock:
        h = _ctxs.get(key)
    if h is None:
        out = c_void_p()
        check(lib().octa_ctx_create(int(device_index), ctypes.byref(out)), "octa_ctx_create")
        with _lock:
            _ctxs[key] = out
        h = out
    return h


def current_stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def advance_python_random(n):
    """`random.random()` n times, natively (same final state of the global generator as the Python loop)."""
    if n <= 0:
        return
    if n < 64:
        for _ in range(n):
            _random.random()
        return
    ver, state, gauss = _random.getstate()
    arr = np.array(state, dtype=np.uint32)
    check(lib().octa_py_random_advance(arr.ctypes.data, int(n)), "octa_py_random_advance")
    _random.setstate((ver, tuple(arr.tolist()), gauss))

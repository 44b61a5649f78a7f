"""ctypes binding of liblograst.so (C ABI: include/lograst.h).  Fails loudly when the library is
missing -- there is no CPU or PyTorch fallback for the product path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LOGRAST_LIB") or os.path.join(_HERE, "lib", "liblograst.so")  # env: experiment builds only

FILTER_NONE, FILTER_DILATE, FILTER_CLAMP = 0, 1, 2
FORM_AUTO, FORM_ROWS, FORM_QUADRANT = 0, 1, 2
GRAD_ROW_FLOATS = 16   # LOGRAST_GRAD_ROW_FLOATS
REC_FLOATS = 16
BWD_ROW_FLOATS = 16   # LOGRAST_BWD_ROW_FLOATS: the reverse walk's accumulator row (64 B per Gaussian)
NUM_KERNELS = 20

c_void_p, c_int32, c_uint32, c_float, c_size_t = (ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32,
                                                  ctypes.c_float, ctypes.c_size_t)


class LograstView(ctypes.Structure):
    """struct lograst_view (include/lograst.h)."""
    _fields_ = [
        ("width", c_int32), ("height", c_int32),
        ("tanfovx", c_float), ("tanfovy", c_float),
        ("scale_modifier", c_float),
        ("filter_mode", c_int32), ("ndc_cull", c_int32), ("extras", c_int32),
        ("viewmatrix", c_void_p), ("projmatrix", c_void_p), ("bg", c_void_p),
        ("tile_row_begin", c_int32), ("tile_row_end", c_int32),
        ("cov3d_precomp", c_void_p), ("dl_dcov3d", c_void_p),
        ("walk_form", c_int32),
        ("hit_masks", c_void_p), ("hit_mask_words", ctypes.c_uint64), ("hit_mask_form", c_int32),
    ]


class LograstAdamKey(ctypes.Structure):
    """struct lograst_adam_key (include/lograst.h)."""
    _fields_ = [("model_param", c_void_p), ("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p),
                ("exp_avg_sq", c_void_p), ("max_exp_avg_sq", c_void_p), ("width", c_int32), ("step_size", c_float)]


class LograstError(RuntimeError):
    pass


_lib = None

_SIGNATURES = {
    "lograst_version": (ctypes.c_int, []),
    "lograst_last_error": (ctypes.c_char_p, []),
    "lograst_tile_state_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "lograst_geom_bytes": (c_size_t, [c_int32]),
    "lograst_keys_bytes": (c_size_t, [c_uint32]),
    "lograst_hit_mask_bytes": (c_size_t, [c_uint32, c_int32, c_int32]),
    "lograst_forward_form": (ctypes.c_int, [ctypes.POINTER(LograstView)]),
    "lograst_list_bytes": (c_size_t, [c_uint32]),
    "lograst_tile_offsets": (c_void_p, [c_void_p, c_int32, c_int32]),
    "lograst_ordered_lengths": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "lograst_finish_lists": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_uint32, c_void_p]),
    "lograst_compute_radius": (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                              c_float, c_float, c_float, c_void_p, c_void_p]),
    "lograst_forward_project": (ctypes.c_int, [ctypes.POINTER(LograstView), c_int32, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               ctypes.POINTER(c_uint32), ctypes.POINTER(c_uint32), c_void_p]),
    "lograst_forward_render": (ctypes.c_int, [ctypes.POINTER(LograstView), c_int32, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_uint32, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "lograst_forward": (ctypes.c_int, [ctypes.POINTER(LograstView), c_int32] + [c_void_p] * 10 + [c_uint32, c_uint32]
                        + [c_void_p] * 7 + [c_int32, c_void_p, c_void_p]),
    "lograst_forward_speculative": (ctypes.c_int, [ctypes.POINTER(LograstView), c_int32] + [c_void_p] * 10
                                    + [c_uint32, c_uint32] + [c_void_p] * 7 + [c_int32, c_void_p]
                                    + [ctypes.POINTER(c_uint32), ctypes.POINTER(c_uint32), c_void_p]),
    "lograst_tile_rows": (ctypes.c_int, [ctypes.POINTER(LograstView), c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p]),
    "lograst_stream_copy": (ctypes.c_int, [c_void_p, c_void_p, c_size_t, c_int32, c_void_p]),
    "lograst_sparse_segment_floats": (c_size_t, [c_int32]),
    "lograst_pack_rows": (ctypes.c_int, [c_void_p, c_int32, ctypes.c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "lograst_pack_rows_clear": (ctypes.c_int, [c_void_p, c_int32, ctypes.c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "lograst_pack_rows_hinted": (ctypes.c_int, [c_void_p, c_int32, ctypes.c_int64, c_int32, c_void_p, c_void_p, c_int32, c_void_p,
                                                ctypes.c_int64, c_void_p]),
    "lograst_add_visible": (ctypes.c_int, [c_void_p, c_void_p, ctypes.c_int64, c_void_p]),
    "lograst_add_visible_n": (ctypes.c_int, [c_void_p, c_void_p, c_int32, ctypes.c_int64, c_void_p]),
    "lograst_unpack_rows": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, ctypes.c_int64, ctypes.c_int64, c_int32,
                                           c_void_p]),
    "lograst_read_state": (ctypes.c_int, [c_void_p, ctypes.POINTER(c_uint32), ctypes.POINTER(c_uint32),
                                          ctypes.POINTER(c_uint32), ctypes.POINTER(c_uint32), c_void_p]),
    "lograst_set_tile_cull": (ctypes.c_int, [ctypes.c_int]),
    "lograst_knob_count": (ctypes.c_int, []),
    "lograst_knob_info": (ctypes.c_int, [c_int32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_int32),
                                         ctypes.POINTER(c_int32), ctypes.POINTER(c_int32), ctypes.POINTER(ctypes.c_char_p)]),
    "lograst_set_knob": (ctypes.c_int, [ctypes.c_char_p, c_int32]),
    "lograst_get_knob": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(c_int32)]),
    "lograst_reset_knobs": (ctypes.c_int, []),
    "lograst_backward": (ctypes.c_int, [ctypes.POINTER(LograstView), c_int32] + [c_void_p] * 18 + [c_int32, c_void_p]),
    "lograst_project_backward": (ctypes.c_int, [ctypes.POINTER(LograstView), c_int32] + [c_void_p] * 10),
    "lograst_sh_forward": (ctypes.c_int, [c_int32, c_int32, c_int32] + [c_void_p] * 6),
    "lograst_sh_backward": (ctypes.c_int, [c_int32, c_int32, c_int32] + [c_void_p] * 7 + [c_int32, c_void_p]),
    "lograst_knn_scratch_bytes": (c_size_t, [c_int32]),
    "lograst_knn_mean_dist2": (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lograst_lod_scratch_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "lograst_lod_traverse": (ctypes.c_int, [c_int32, c_int32, c_int32] + [c_void_p] * 6 + [c_int32, c_void_p, c_void_p,
                                            c_float, c_float, c_float, c_float, c_float, c_int32, c_void_p, c_uint32,
                                            c_void_p, c_size_t, c_void_p]),
    "lograst_lod_read": (ctypes.c_int, [c_void_p, ctypes.POINTER(c_uint32), ctypes.POINTER(c_uint32),
                                        ctypes.POINTER(c_uint32), c_void_p]),
    "lograst_id_histogram_scratch_bytes": (c_size_t, [c_int32]),
    "lograst_id_histogram": (ctypes.c_int, [c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lograst_id_histogram_read": (ctypes.c_int, [c_void_p, ctypes.POINTER(c_uint32), c_void_p]),
    "lograst_counter_update": (ctypes.c_int, [c_int32] + [c_void_p] * 4 + [c_int32, c_void_p, c_void_p, c_int32]
                               + [c_void_p] * 10),
    "lograst_sparse_adam": (ctypes.c_int, [c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                           ctypes.POINTER(LograstAdamKey), ctypes.c_double, ctypes.c_double,
                                           ctypes.c_double, ctypes.c_double, c_void_p]),
    "lograst_activate_backward_adam": (ctypes.c_int, [c_int32] + [c_void_p] * 4 + [c_int32, c_int32] + [c_void_p] * 6 +
                                       [c_int32, c_void_p, c_void_p, ctypes.POINTER(LograstAdamKey), ctypes.c_double,
                                        ctypes.c_double, ctypes.c_double, ctypes.c_double, c_void_p]),
    "lograst_gather_activate": (ctypes.c_int, [c_int32, c_int32] + [c_void_p] * 7 + [c_int32, c_int32] + [c_void_p] * 12),
    "lograst_activate_backward": (ctypes.c_int, [c_int32] + [c_void_p] * 4 + [c_int32, c_int32] + [c_void_p] * 11),
    "lograst_profile_enable": (None, [ctypes.c_int]),
    "lograst_profile_reset": (None, []),
    "lograst_profile_read": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
    "lograst_kernel_name": (ctypes.c_char_p, [ctypes.c_int]),
}

EXPORTS = tuple(_SIGNATURES)


def lib():
    """Loads liblograst.so (after torch, so both share one HIP runtime).  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LograstError(
                f"{LIB_PATH} is missing: build it with `python -m log_amd.build` (hipcc, gfx950). "
                "log_amd has no CPU fallback.")
        import torch  # noqa: F401  -- loads libamdhip64 first so the .so binds to torch's runtime
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.lograst_version() != 4:
            raise LograstError("liblograst.so version mismatch; rebuild")
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise LograstError(f"liblograst error {rc}: {lib().lograst_last_error().decode()}")


def profile_enable(on=True):
    lib().lograst_profile_enable(1 if on else 0)


def profile_reset():
    lib().lograst_profile_reset()


def profile_read():
    """-> {kernel_name: (total_ms, launches)} since the last reset (synchronises recorded events)."""
    ms = (ctypes.c_double * NUM_KERNELS)()
    cnt = (ctypes.c_int64 * NUM_KERNELS)()
    check(lib().lograst_profile_read(ms, cnt))
    return {lib().lograst_kernel_name(i).decode(): (ms[i], cnt[i]) for i in range(NUM_KERNELS) if cnt[i]}

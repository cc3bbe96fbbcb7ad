"""ctypes binding of libradian_hip.so (include/radian_hip.h).  Fails loudly when the library is missing."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RADIAN_HIP_LIB") or os.path.join(HERE, "libradian_hip.so")   # override: kernel experiments only

c_i = ctypes.c_int
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64 = ctypes.c_int64
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_d = ctypes.c_double
c_dp = ctypes.POINTER(ctypes.c_double)
c_fp = ctypes.POINTER(ctypes.c_float)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_vp = ctypes.c_void_p
c_sz = ctypes.c_size_t

# every symbol include/radian_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "rd_last_error": (ctypes.c_char_p, []),
    "rd_version": (c_i, []),
    "rd_device_count": (c_i, [ctypes.POINTER(c_i)]),
    "rd_decode_max_width": (c_i, []),
    "rd_decode_lane_width": (c_i, []),
    "rd_create": (c_i, [c_i, ctypes.POINTER(c_vp)]),
    "rd_destroy": (c_i, [c_vp]),
    "rd_sync": (c_i, [c_vp]),
    "rd_set_precision": (c_i, [c_vp, c_i]),
    "rd_load_weights": (c_i, [c_vp, c_vp, c_sz]),
    "rd_load_lm": (c_i, [c_vp, c_vp, c_i]),
    "rd_load_lm_absent": (c_i, [c_vp, c_i]),
    "rd_load_lm_hashed": (c_i, [c_vp, c_vp, c_i, c_i]),
    "rd_set_logits": (c_i, [c_vp, c_i]),
    "rd_set_decode_math": (c_i, [c_vp, c_i]),
    "rd_set_decode_partition": (c_i, [c_vp, c_i]),
    "rd_forward": (c_i, [c_vp, c_vp, c_i, c_i, c_vp]),
    "rd_assemble": (c_i, [c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp, c_i64, c_i64p, ctypes.POINTER(c_i)]),
    "rd_decode_batch": (c_i, [c_vp, c_vp, c_i, c_vp, c_vp, c_i, c_i, c_i, c_d, c_d, c_vp, c_vp, c_vp, c_vp]),
    "rd_basecall_chunk": (c_i, [c_vp, c_vp, c_i, c_i, c_vp, c_i, c_vp, c_vp]),
    "rd_basecall_global": (c_i, [c_vp, c_vp, c_i, c_i, c_vp, c_vp, c_i, c_i, c_i, c_d, c_d, c_vp, c_vp, c_vp]),
    "rd_count_windows": (c_i, [c_i64, c_i, c_i]),
    "rd_basecall_reads_chunk": (c_i, [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp, c_vp]),
    "rd_basecall_reads_global": (c_i, [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_d, c_d, c_vp, c_vp, c_vp]),
    "rd_basecall_reads_chunk_resident": (c_i, [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp, c_vp]),
    "rd_basecall_reads_global_resident": (c_i, [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_d, c_d, c_vp, c_vp, c_vp]),
    "rd_pipe_submit_reads": (c_i, [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp, c_vp]),
    "rd_pipe_submit_reads_global": (c_i, [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_d, c_d, c_vp, c_vp, c_vp]),
    "rd_pipe_submit_raw_global": (c_i, [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_d, c_d, c_vp, c_vp, c_vp, c_vp]),
    "rd_pipe_submit_raw_chunk": (c_i, [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp, c_vp]),
    "rd_pipe_progress": (c_i, [c_vp, c_i64, c_i64p]),
    "rd_pipe_submitted": (c_i, [c_vp, c_i64p]),
    "rd_normalise_reads": (c_i, [c_vp, c_vp, c_vp, c_i, c_i, c_vp, c_vp]),
    "rd_basecall_raw_chunk": (c_i, [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp, c_vp]),
    "rd_basecall_raw_global": (c_i, [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_d, c_d, c_vp, c_vp, c_vp, c_vp]),
    "rd_stitch_chunk": (c_i, [c_vp, c_vp, c_i, c_vp, c_i, c_vp, c_vp, c_vp, c_i]),
    "rd_fast5_open": (c_i, [ctypes.c_char_p, ctypes.POINTER(c_vp)]),
    "rd_fast5_open_mem": (c_i, [c_vp, c_sz, ctypes.POINTER(c_vp)]),
    "rd_fast5_close": (None, [c_vp]),
    "rd_fast5_count": (c_i, [c_vp, c_i64p]),
    "rd_fast5_lengths": (c_i, [c_vp, c_i64, c_i64, c_vp]),
    "rd_fast5_read_batch": (c_i, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_i]),
    "rd_lm_json_probe": (c_i, [c_vp, ctypes.c_size_t, ctypes.POINTER(c_i)]),
    "rd_lm_json_fill": (c_i, [c_vp, ctypes.c_size_t, c_i, c_vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "rd_dev_alloc": (c_i, [c_vp, c_sz, ctypes.POINTER(c_vp)]),
    "rd_mem_info": (c_i, [c_vp, ctypes.POINTER(c_sz), ctypes.POINTER(c_sz)]),
    "rd_dev_free": (c_i, [c_vp, c_vp]),
    "rd_memcpy_h2d": (c_i, [c_vp, c_vp, c_vp, c_sz]),
    "rd_memcpy_d2h": (c_i, [c_vp, c_vp, c_vp, c_sz]),
    "rd_forward_resident": (c_i, [c_vp, c_vp, c_i, c_i, c_vp]),
    "rd_forward_reads_resident": (c_i, [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp]),
    "rd_forward_reads": (c_i, [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp, c_i64, c_vp]),
    "rd_basecall_chunk_resident": (c_i, [c_vp, c_vp, c_i, c_i, c_vp, c_i, c_vp, c_vp]),
    "rd_decode_resident": (c_i, [c_vp, c_vp, c_i, c_i, c_vp, c_i, c_vp, c_vp]),
    "rd_pipe_config": (c_i, [c_vp, c_i]),
    "rd_pipe_set_lanes": (c_i, [c_vp, c_i]),
    "rd_pipe_submit": (c_i, [c_vp, c_vp, c_i, c_i, c_vp, c_i, c_vp, c_vp]),
    "rd_pipe_flush": (c_i, [c_vp]),
    "rd_rccl_probe": (c_i, []),
    "rd_rccl_unique_id": (c_i, [c_vp]),
    "rd_rccl_init": (c_i, [c_vp, c_i, c_i, c_vp]),
    "rd_rccl_bcast_model": (c_i, [c_vp, c_i]),
    "rd_clone_artifacts": (c_i, [c_vp, c_vp]),
    "rd_rccl_allreduce_max": (c_i, [c_vp, c_vp, c_i]),
    "rd_rccl_barrier": (c_i, [c_vp]),
    "rd_rccl_comm_count": (c_i, [c_vp, ctypes.POINTER(c_i)]),
    "rd_rccl_finalize": (c_i, [c_vp]),
}

# every symbol include/radian_hip_diag.h declares (measurement switches, kernel timers, pipeline read-outs: bench.py, tools/, tests/;
# the command line calls none of them)
DIAG_SIGNATURES = {
    "rd_set_conv_fuse": (c_i, [c_vp, c_i]),
    "rd_set_conv_shape": (c_i, [c_vp, c_i]),
    "rd_split3": (c_i, [c_vp, c_vp, c_sz, c_vp]),
    "rd_set_decode_form": (c_i, [c_vp, c_i]),
    "rd_pipe_policy_read": (c_i, [c_vp, c_i, c_i, c_i, c_vp, c_vp, c_vp]),
    "rd_pipe_stats": (c_i, [c_vp, c_i64p, c_i]),
    "rd_set_trie_budget": (c_i, [c_vp, c_i64]),
    "rd_timer_enable": (c_i, [c_vp, c_i, c_i]),
    "rd_timer_read": (c_i, [c_vp, c_i, c_dp, ctypes.POINTER(c_i), c_dp, c_dp]),
    "rd_timer_read_launches": (c_i, [c_vp, c_i, c_i, c_vp, c_vp, c_vp, ctypes.POINTER(c_i)]),
}

_lib = None


def load():
    """dlopen the in-tree library and bind every declared symbol.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m radian_amd.build` (hipcc --offload-arch=gfx950). "
            "radian_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in {**SIGNATURES, **DIAG_SIGNATURES}.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib

"""ctypes binding of libraiko_hip.so (the C ABI declared in include/raiko_hip.h).

There is no CPU fallback: if the shared library is missing or fails to load the
import of anything that computes raises `HipLibraryError`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAIKO_HIP_LIB: another build of the same library (tests/asan/run_sanitized.sh points it at the build whose HOST code is
# compiled with -fsanitize=address,undefined); there is still no fallback -- a path that does not load is an error
LIB_PATH = os.environ.get("RAIKO_HIP_LIB") or os.path.join(_HERE, "libraiko_hip.so")


class HipLibraryError(RuntimeError):
    pass


class RkError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__(f"libraiko_hip status {status}: {detail}")


u32p = C.POINTER(C.c_uint32)


class RkTaps(C.Structure):
    _fields_ = [
        ("group_size", C.c_uint32 * 3),
        ("n_regs", C.c_uint32),
        ("reg_group", u32p),
        ("reg_offset", u32p),
        ("reg_combo", u32p),
        ("n_combos", C.c_uint32),
        ("combo_off", u32p),
        ("combo_backs", u32p),
    ]


class RkSegment(C.Structure):
    _fields_ = [
        ("po2", C.c_uint32),
        ("on_device", C.c_uint32),
        ("taps", RkTaps),
        ("group", C.c_void_p * 3),
        ("check", C.c_void_p),
        ("globals", u32p),
        ("n_globals", C.c_uint32),
        ("n_accum_mix", C.c_uint32),
        ("proof_system_info", C.c_uint8 * 16),
        ("circuit_info", C.c_uint8 * 16),
        ("hooks", C.c_void_p),
    ]


class RkCircuitView(C.Structure):
    _fields_ = [
        ("ctx", C.c_void_p),
        ("stream", C.c_void_p),
        ("po2", C.c_uint32),
        ("group_size", C.c_uint32 * 3),
        ("d_trace", C.c_void_p * 3),
        ("d_lde", C.c_void_p * 3),
        ("globals", u32p),
        ("n_globals", C.c_uint32),
        ("mix", u32p),
        ("n_mix", C.c_uint32),
    ]


ACCUMULATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(RkCircuitView), C.c_void_p)
EVAL_CHECK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(RkCircuitView), u32p, C.c_void_p)
POLY_EXT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(RkSegment), u32p, u32p, C.c_size_t, u32p, C.c_uint32, u32p)


class RkCircuitHooks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("accumulate", ACCUMULATE_FN), ("eval_check", EVAL_CHECK_FN),
                ("program", C.c_void_p)]


class RkParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("ext_w", C.c_uint32), ("root_2_27", C.c_uint32), ("coset_shift", C.c_uint32),
                ("p2_width", C.c_uint32), ("p2_m4", C.c_uint32), ("p2_pad_free", C.c_uint32),
                ("p2_rc_ext", u32p), ("p2_rc_int", u32p), ("p2_diag", u32p),
                ("queries", C.c_uint32), ("blowup_log2", C.c_uint32), ("fri_fold_log2", C.c_uint32),
                ("fri_min_degree", C.c_uint32), ("pow_bits", C.c_uint32)]


RK_PRESET_RISC0, RK_PRESET_SP1 = 0, 1


class RkMatrix(C.Structure):
    _fields_ = [("d_values", C.c_void_p), ("height", C.c_uint32), ("width", C.c_uint32), ("row_major", C.c_uint32)]


class RkVerifyOpts(C.Structure):
    _fields_ = [("p2_rc_ext", u32p), ("p2_rc_int", u32p), ("p2_diag", u32p), ("poly_ext", POLY_EXT_FN),
                ("user", C.c_void_p), ("program", C.c_void_p), ("params", C.POINTER(RkParams))]


class RkSessionOpts(C.Structure):
    _fields_ = [("device", C.c_int), ("inflight", C.c_int), ("upload_ahead", C.c_int), ("verify", C.c_int),
                ("devices", C.POINTER(C.c_int)), ("n_devices", C.c_int), ("verify_opts", C.POINTER(RkVerifyOpts)),
                ("params", C.POINTER(RkParams))]


RK_ERR_INVALID = -1
RK_ERR_CAPACITY = -5
RK_ERR_INTERNAL = -6
RK_ERR_VERIFY = -7
RK_ERR_CALLBACK = -8


class RkKernelStat(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("ms", C.c_double), ("bytes", C.c_double)]


KCLASS_COUNT = 5


class RkTiming(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("ntt", "hash", "deep", "fri", "query", "total", "circuit")]


class RkAirInfo(C.Structure):
    _fields_ = [("n_steps", C.c_uint64), ("n_ops", C.c_uint64), ("n_constraints", C.c_uint32), ("max_degree", C.c_uint32),
                ("log_quotient_degree", C.c_uint32), ("n_fp_slots", C.c_uint32)]


class RkP3Table(C.Structure):
    _fields_ = [("trace", C.c_void_p), ("log_height", C.c_uint32), ("width", C.c_uint32), ("air", C.c_void_p),
                ("public_values", u32p), ("n_public", C.c_uint32), ("on_device", C.c_uint32)]


class RkP3Shard(C.Structure):
    _fields_ = [("tables", C.POINTER(RkP3Table)), ("n_tables", C.c_uint32), ("init_words", u32p), ("n_init", C.c_size_t),
                ("h_proof", u32p), ("capacity_words", C.c_size_t), ("proof_words", C.c_size_t)]


class RkP3SessionOpts(C.Structure):
    _fields_ = [("device", C.c_int), ("batch", C.c_int), ("verify", C.c_int), ("devices", C.POINTER(C.c_int)), ("n_devices", C.c_int),
                ("params", C.POINTER(RkParams))]


class RkP3Timing(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("lde", "commit", "quotient", "open", "fri", "query", "total", "perm")]


class RkExecOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("segment_limit_po2", C.c_uint32), ("session_limit", C.c_uint64),
                ("input_words", u32p), ("n_input_words", C.c_size_t), ("record_trace", C.c_uint32), ("profile", C.c_uint32)]


class RkExecSummary(C.Structure):
    _fields_ = [("total_cycles", C.c_uint64), ("n_segments", C.c_uint32), ("exit_code", C.c_uint32),
                ("journal_bytes", C.c_size_t), ("input_words_read", C.c_size_t), ("status", C.c_int)]


class RkExecSegment(C.Structure):
    _fields_ = [("index", C.c_uint32), ("po2", C.c_uint32), ("cycles", C.c_uint64), ("start_pc", C.c_uint32),
                ("end_pc", C.c_uint32), ("exit", C.c_uint32), ("pre_state", C.c_uint32 * 8), ("post_state", C.c_uint32 * 8)]


# every symbol include/raiko_hip.h declares: name -> (restype, argtypes)
_vp, _sz, _u32 = C.c_void_p, C.c_size_t, C.c_uint32
SYMBOLS = {
    "rk_prove_session": (C.c_int, [C.POINTER(RkSessionOpts), C.POINTER(RkSegment), _sz, C.POINTER(u32p), C.POINTER(_sz),
                                   C.POINTER(_sz), C.POINTER(_sz)]),
    "rk_session_last_error": (C.c_char_p, [C.c_int]),
    "rk_stream_open": (C.c_int, [C.POINTER(RkSessionOpts), C.POINTER(C.c_void_p)]),
    "rk_stream_submit": (C.c_int, [C.c_void_p, C.POINTER(RkSegment), u32p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rk_stream_wait": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rk_stream_close": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "rk_session_release": (C.c_int, []),
    "rk_session_last_proven": (C.c_int, [C.c_int, C.POINTER(C.c_size_t)]),
    "rk_abi_version": (C.c_int, []),
    "rk_strerror": (C.c_char_p, [C.c_int]),
    "rk_last_error": (C.c_char_p, [_vp]),
    "rk_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rk_ctx_create": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "rk_ctx_destroy": (C.c_int, [_vp]),
    "rk_sync": (C.c_int, [_vp]),
    "rk_alloc": (C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    "rk_free": (C.c_int, [_vp, _vp]),
    "rk_h2d": (C.c_int, [_vp, _vp, _vp, _sz]),
    "rk_d2h": (C.c_int, [_vp, _vp, _vp, _sz]),
    "rk_set_poseidon2_params": (C.c_int, [_vp, u32p, u32p, u32p]),
    "rk_batch_interpolate_ntt": (C.c_int, [_vp, _vp, _sz, _sz]),
    "rk_batch_evaluate_ntt": (C.c_int, [_vp, _vp, _sz, _sz, _u32]),
    "rk_zk_shift": (C.c_int, [_vp, _vp, _sz, _sz]),
    "rk_batch_expand_into_evaluate_ntt": (C.c_int, [_vp, _vp, _vp, _sz, _sz, _u32]),
    "rk_batch_bit_reverse": (C.c_int, [_vp, _vp, _sz, _sz]),
    "rk_hash_rows": (C.c_int, [_vp, _vp, _vp, _sz, _sz]),
    "rk_hash_fold": (C.c_int, [_vp, _vp, _sz, _sz]),
    "rk_batch_evaluate_any": (C.c_int, [_vp, _vp, _sz, _sz, u32p, u32p, _sz, u32p]),
    "rk_mix_poly_coeffs": (C.c_int, [_vp, _vp, u32p, u32p, _vp, u32p, _sz, _sz]),
    "rk_eltwise_add_elem": (C.c_int, [_vp, _vp, _vp, _vp, _sz]),
    "rk_eltwise_sum_extelem": (C.c_int, [_vp, _vp, _vp, _sz, _sz]),
    "rk_eltwise_copy_elem": (C.c_int, [_vp, _vp, _vp, _sz]),
    "rk_eltwise_zeroize_elem": (C.c_int, [_vp, _vp, _sz]),
    "rk_fri_fold": (C.c_int, [_vp, _vp, _vp, _sz, u32p]),
    "rk_fri_fold_evals": (C.c_int, [_vp, _vp, _vp, _sz, u32p]),
    "rk_pcs_coset_lde_rows": (C.c_int, [_vp, _vp, _vp, _sz, _sz]),
    "rk_pcs_coset_lde_cols": (C.c_int, [_vp, _vp, _vp, _sz, _sz]),
    "rk_pcs_eval_at_many_cols": (C.c_int, [_vp, _vp, _vp, _sz, _sz, C.c_uint32, u32p]),
    "rk_pcs_reduce_openings_cols": (C.c_int, [_vp, _vp, _vp, _sz, _sz, C.c_uint32, u32p, u32p, u32p, C.c_uint64]),
    "rk_pcs_eval_at": (C.c_int, [_vp, _vp, _vp, _sz, _sz, u32p]),
    "rk_pcs_eval_at_many": (C.c_int, [_vp, _vp, _vp, _sz, _sz, C.c_uint32, u32p]),
    "rk_duplex_grind": (C.c_int, [_vp, u32p, u32p, C.c_uint32, C.c_uint32, u32p]),
    "rk_pcs_reduce_openings": (C.c_int, [_vp, _vp, _vp, _sz, _sz, C.c_uint32, u32p, u32p, u32p, C.c_uint64]),
    "rk_gather_sample": (C.c_int, [_vp, _vp, _vp, _sz, _sz, _sz]),
    "rk_merkle_build": (C.c_int, [_vp, _vp, _vp, _sz, _sz]),
    "rk_poly_divide": (C.c_int, [_vp, _vp, _sz, u32p, u32p]),
    "rk_prove_segment": (C.c_int, [_vp, C.POINTER(RkSegment), u32p, _sz, C.POINTER(_sz)]),
    "rk_verify_segment": (C.c_int, [C.POINTER(RkSegment), u32p, _sz]),
    "rk_verify_segment_ex": (C.c_int, [C.POINTER(RkSegment), C.POINTER(RkVerifyOpts), u32p, _sz]),
    "rk_prefix_products": (C.c_int, [_vp, _vp, _sz]),
    "rk_scatter": (C.c_int, [_vp, _vp, _sz, u32p, _sz, u32p, u32p]),
    "rk_seal_bound_words": (_sz, [C.POINTER(RkSegment)]),
    "rk_seal_bound_words_for": (_sz, [C.POINTER(RkSegment), _u32]),
    "rk_seal_bound_words_params": (_sz, [C.POINTER(RkSegment), C.POINTER(RkParams)]),
    "rk_pow_grind": (C.c_int, [_vp, u32p, _u32, u32p]),
    "rk_mmcs_commit": (C.c_int, [_vp, C.POINTER(RkMatrix), _u32, _vp, u32p]),
    "rk_mmcs_open": (C.c_int, [_vp, C.POINTER(RkMatrix), _u32, _vp, _u32, u32p, u32p]),
    "rk_mmcs_verify": (C.c_int, [C.POINTER(RkParams), u32p, u32p, _u32, _u32, u32p, u32p, u32p]),
    "rk_params_preset": (C.c_int, [C.POINTER(RkParams), C.c_int]),
    "rk_set_params": (C.c_int, [_vp, C.POINTER(RkParams)]),
    "rk_get_params": (C.c_int, [_vp, C.POINTER(RkParams)]),
    "rk_last_timing": (C.c_int, [_vp, C.POINTER(RkTiming)]),
    "rk_set_kernel_timing": (C.c_int, [_vp, C.c_int]),
    "rk_kernel_stats": (C.c_int, [_vp, C.c_int, C.POINTER(RkKernelStat)]),
    "rk_kernel_class_name": (C.c_char_p, [C.c_int]),
    "rk_exec_elf": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(RkExecOpts), C.POINTER(C.c_void_p)]),
    "rk_exec_open": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(RkExecOpts), C.POINTER(C.c_void_p)]),
    "rk_exec_next_segment": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "rk_exec_summary_get": (C.c_int, [C.c_void_p, C.POINTER(RkExecSummary)]),
    "rk_exec_segment_get": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(RkExecSegment)]),
    "rk_exec_journal": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rk_exec_profile": (C.c_int, [C.c_void_p, u32p, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_size_t)]),
    "rk_exec_witness": (C.c_int, [C.c_void_p, C.c_uint32, u32p, u32p]),
    "rk_exec_lookup_tables": (C.c_int, [C.c_void_p, C.c_uint32, u32p, u32p, C.POINTER(C.c_size_t)]),
    "rk_exec_witness_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "rk_exec_witness_device_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "rk_exec_error": (C.c_char_p, [C.c_void_p]),
    "rk_exec_free": (C.c_int, [C.c_void_p]),
    "rk_program_create": (C.c_int, [_vp, _sz, _u32, C.POINTER(RkTaps), C.POINTER(_vp)]),
    "rk_program_destroy": (C.c_int, [_vp]),
    "rk_program_get_info": (C.c_int, [_vp, _vp]),
    "rk_program_eval_check": (C.c_int, [_vp, C.POINTER(RkCircuitView), u32p, _vp]),
    "rk_program_compile": (C.c_int, [_vp, _vp]),
    "rk_program_source": (C.c_int, [_vp, C.c_char_p, _sz, C.POINTER(_sz)]),
    "rk_program_poly_ext": (C.c_int, [_vp, _u32, u32p, u32p, _sz, u32p, _u32, u32p, _u32, u32p]),
    "rk_air_create": (C.c_int, [_vp, _sz, _u32, _u32, C.POINTER(_vp)]),
    "rk_air_create_lookup": (C.c_int, [_vp, _sz, _u32, _u32, u32p, _u32, _sz, _u32, C.POINTER(_vp)]),
    "rk_p2_chip_width": (_u32, [C.POINTER(RkParams)]),
    "rk_p2_chip_air": (C.c_int, [C.POINTER(RkParams), _u32, C.POINTER(_vp)]),
    "rk_p2_chip_trace": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "rk_air_get_steps": (C.c_int, [_vp, _vp, _sz, C.POINTER(_sz)]),
    "rk_air_destroy": (C.c_int, [_vp]),
    "rk_air_get_info": (C.c_int, [_vp, C.POINTER(RkAirInfo)]),
    "rk_air_compile": (C.c_int, [_vp, _vp]),
    "rk_p3_prove": (C.c_int, [_vp, C.POINTER(RkP3Table), _u32, u32p, _sz, u32p, _sz, C.POINTER(_sz)]),
    "rk_p3_verify": (C.c_int, [C.POINTER(RkParams), C.POINTER(RkP3Table), _u32, u32p, _sz, u32p, _sz]),
    "rk_p3_proof_bound_words": (_sz, [C.POINTER(RkParams), C.POINTER(RkP3Table), _u32]),
    "rk_p3_verify_hashes": (C.c_int, [C.POINTER(RkParams), C.POINTER(RkP3Table), _u32, u32p, _sz, u32p, _sz, u32p, _sz, C.POINTER(_sz)]),
    "rk_p3_last_timing": (C.c_int, [_vp, C.POINTER(RkP3Timing)]),
    "rk_p3_prove_shards": (C.c_int, [C.POINTER(RkP3SessionOpts), C.POINTER(RkP3Shard), _sz, C.POINTER(_sz)]),
    "rk_comm_unique_id": (C.c_int, [C.c_char_p]),
    "rk_comm_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "rk_comm_destroy": (C.c_int, [_vp]),
    "rk_comm_last_error": (C.c_char_p, [_vp]),
    "rk_gather_seals": (C.c_int, [_vp, C.POINTER(u32p), C.POINTER(_sz), _sz, _sz, C.POINTER(u32p), C.POINTER(_sz), C.POINTER(_sz)]),
    "rk_gather_unpack": (C.c_int, [u32p, u32p, C.c_int, _sz, _sz, _sz, C.POINTER(u32p), C.POINTER(_sz), C.POINTER(_sz)]),
    "rk_session_set_kernel_timing": (C.c_int, [C.c_int, C.c_int]),
    "rk_session_kernel_stats": (C.c_int, [C.c_int, C.c_int, C.POINTER(RkKernelStat)]),
}

_lib = None


def load():
    """Load the shared library (once) and attach prototypes.  Raises HipLibraryError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  raiko_amd has no CPU fallback."
        )
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing ROCm runtime etc.
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx_handle, status):
    if status != 0:
        lib = load()
        msg = lib.rk_strerror(status).decode()
        if ctx_handle:
            detail = lib.rk_last_error(ctx_handle).decode()
            if detail:
                msg += " (" + detail + ")"
        raise RkError(status, msg)

"""ctypes binding of libbfstark_hip.so (C ABI: include/bfstark.h).

There is no CPU fallback: if the library is missing or a HIP call fails, the caller gets an exception.
Error codes that correspond to the reference's `assert`s are raised as AssertionError with the reference's
message, everything else as RuntimeError.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BFS_LIB_PATH") or os.path.join(_HERE, "libbfstark_hip.so")   # override: development builds only

u64 = ctypes.c_uint64
u32 = ctypes.c_uint32
vp = ctypes.c_void_p
sz = ctypes.c_size_t
ci = ctypes.c_int

BFS_OK = 0
ASSERT_CODES = {1, 2, 3, 4, 5, 8, 9}

_SIGNATURES = {
    "bfs_version": (ci, []),
    "bfs_last_error": (ctypes.c_char_p, []),
    "bfs_device_count": (ci, [ctypes.POINTER(ci)]),
    "bfs_set_device": (ci, [ci]),
    "bfs_malloc": (ci, [ctypes.POINTER(vp), sz]),
    "bfs_free": (ci, [vp]),
    "bfs_malloc_async": (ci, [ctypes.POINTER(vp), sz, vp]),
    "bfs_free_async": (ci, [vp, vp]),
    "bfs_pool_trim": (ci, []),
    "bfs_pool_stats": (ci, [ctypes.POINTER(sz), ctypes.POINTER(sz)]),
    "bfs_host_alloc": (ci, [ctypes.POINTER(vp), sz]),
    "bfs_host_free": (ci, [vp]),
    "bfs_memcpy_h2d": (ci, [vp, vp, sz, vp]),
    "bfs_memcpy_d2h": (ci, [vp, vp, sz, vp]),
    "bfs_memcpy_d2d": (ci, [vp, vp, sz, vp]),
    "bfs_memset": (ci, [vp, ci, sz, vp]),
    "bfs_stream_synchronize": (ci, [vp]),
    "bfs_stream_create": (ci, [ctypes.POINTER(vp)]),
    "bfs_stream_destroy": (ci, [vp]),
    "bfs_event_create": (ci, [ctypes.POINTER(vp)]),
    "bfs_event_destroy": (ci, [vp]),
    "bfs_event_record": (ci, [vp, vp]),
    "bfs_event_elapsed_ms": (ci, [vp, vp, ctypes.POINTER(ctypes.c_float)]),
    "bfs_gl_primitive_root": (u64, [u32]),
    "bfs_gl_mul": (u64, [u64, u64]),
    "bfs_gl_inv": (u64, [u64]),
    "bfs_gl_pow": (u64, [u64, u64]),
    "bfs_gl_ntt": (ci, [vp, u64, u64, vp, u64, u32, u32, u64, u64, u64, vp]),
    "bfs_ntt_route_probe_info": (ci, [vp, vp, vp]),
    "bfs_ntt_tune": (ci, [vp, u64, vp, u64, u32, u32, u64, vp, ctypes.POINTER(ci)]),
    "bfs_ntt_route_forget": (ci, [vp, ctypes.POINTER(sz)]),
    "bfs_gl_scale": (ci, [vp, vp, u64, u64, u32, u64, vp]),
    "bfs_gl_mul_pointwise": (ci, [vp, vp, vp, u64, vp]),
    "bfs_gl_batch_inverse": (ci, [vp, vp, u64, vp]),
    "bfs_xfe_mul_pointwise": (ci, [vp, u64, vp, u64, vp, u64, u64, vp]),
    "bfs_xfe_batch_inverse": (ci, [vp, u64, vp, u64, u64, vp]),
    "bfs_ps_new": (vp, []),
    "bfs_ps_loads": (vp, [ctypes.c_char_p, sz]),
    "bfs_ps_free": (None, [vp]),
    "bfs_ps_obj_bytes": (u64, [vp, ctypes.c_char_p, sz]),
    "bfs_ps_obj_int": (u64, [vp, u64]),
    "bfs_ps_obj_xfe": (u64, [vp, ctypes.POINTER(u64)]),
    "bfs_ps_obj_bfe": (u64, [vp, u64, ci]),
    "bfs_ps_obj_xfe_from": (u64, [vp, ctypes.POINTER(u64), sz]),
    "bfs_ps_obj_list": (u64, [vp, ctypes.POINTER(u64), sz]),
    "bfs_ps_obj_tuple": (u64, [vp, ctypes.POINTER(u64), sz]),
    "bfs_ps_push": (ci, [vp, u64]),
    "bfs_ps_num_objects": (sz, [vp]),
    "bfs_ps_object_at": (u64, [vp, sz]),
    "bfs_ps_serialize": (ci, [vp, sz, vp, sz, ctypes.POINTER(sz)]),
    "bfs_ps_prefetch_fiat_shamir": (sz, [vp, ctypes.POINTER(sz), sz, sz]),
    "bfs_ps_fiat_shamir": (ci, [vp, sz, vp, sz]),
    "bfs_sample_weights": (ci, [ctypes.c_char_p, sz, sz, ctypes.POINTER(u64)]),
    "bfs_ps_push_digest_fiat_shamir": (ci, [vp, ctypes.c_char_p, vp, sz]),
    "bfs_ps_push_digests_fiat_shamir": (ci, [vp, ctypes.c_char_p, sz, vp, sz, ctypes.POINTER(ci)]),
    "bfs_ps_obj_dumps": (ci, [vp, u64, vp, sz, ctypes.POINTER(sz)]),
    "bfs_ps_merkle_verify": (ci, [vp, u64, u64, u64, u64, ctypes.c_char_p, sz, ctypes.POINTER(ci)]),
    "bfs_xfe_inner_product": (ci, [ctypes.POINTER(u64), ctypes.POINTER(u64), sz, ctypes.POINTER(u64)]),
    "bfs_ps_obj_kind": (ci, [vp, u64]),
    "bfs_ps_obj_len": (sz, [vp, u64]),
    "bfs_ps_obj_item": (u64, [vp, u64, sz]),
    "bfs_ps_obj_get_bytes": (ci, [vp, u64, vp, sz]),
    "bfs_ps_obj_get_limbs": (ci, [vp, u64, ctypes.POINTER(u64)]),
    "bfs_gl_sample": (u64, [ctypes.c_char_p, sz]),
    "bfs_xfe_sample": (None, [ctypes.c_char_p, sz, ctypes.POINTER(u64)]),
    "bfs_gather": (ci, [vp, u32, vp, vp]),
    "bfs_merkle_build_xfe": (ci, [vp, u64, u64, vp, vp]),
    "bfs_merkle_build_bfe": (ci, [vp, u64, vp, vp]),
    "bfs_merkle_build_bytes": (ci, [vp, vp, vp, u64, vp, vp]),
    "bfs_merkle_open": (ci, [vp, u32, u64, vp, vp]),
    "bfs_merkle_build_rows": (ci, [vp, u32, u64, vp, ci, vp, vp]),
    "bfs_stark_push_openings": (ci, [vp, vp, u32, ctypes.c_int32, vp, u32, ctypes.POINTER(u64), u32, u64, vp, vp, ci, vp, vp, ci, vp, u64, vp,
                                     ctypes.POINTER(u64), u32, ctypes.POINTER(u64), u32, ctypes.POINTER(u64), vp]),
    "bfs_merkle_build_rows_range": (ci, [vp, u32, u64, u64, vp, ci, vp, vp]),
    "bfs_merkle_build_rows_root": (ci, [vp, u32, u64, u64, vp, ci, vp, vp, vp]),
    "bfs_row_template_steps": (ci, [vp, u32, u32, ci, vp, vp, u32, vp, u32, vp]),
    "bfs_trace_pad": (ci, [vp, u32, vp]),
    "bfs_row_generated_launches": (u64, []),
    "bfs_random_fill": (ci, [ctypes.c_char_p, vp, u64, vp]),
    "bfs_xfe_sample_fill": (ci, [ctypes.c_char_p, vp, u64, u64, vp]),
    "bfs_xfe_fold": (ci, [vp, u64, vp, u64, u32, ctypes.POINTER(u64), u64, u64, vp]),
    "bfs_fri_session_new": (vp, []),
    "bfs_fri_session_free": (None, [vp]),
    "bfs_fri_commit": (ci, [vp, vp, vp, u64, u32, u64, u64, u32, vp]),
    "bfs_fri_query": (ci, [vp, vp, u32, ctypes.POINTER(u64), vp]),
    "bfs_fri_session_alias": (ci, [vp, vp, u32, u64, u64]),
    "bfs_fri_session_round0_tree": (ci, [vp, vp, ctypes.c_char_p]),
    "bfs_fri_prove": (ci, [vp, vp, u64, u32, u64, u64, u32, u32, ctypes.POINTER(u64), vp]),
    "bfs_fri_last_timing": (None, [ctypes.POINTER(ctypes.c_double)]),
    "bfs_fri_session_rounds": (u32, [vp]),
    "bfs_fri_session_round": (ci, [vp, u32, ctypes.POINTER(vp), ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(vp), vp]),
    "bfs_selftest_field": (ci, [u32, ctypes.POINTER(u64)]),
    "bfs_vm_trace_new": (ci, [ctypes.POINTER(u64), sz, ctypes.POINTER(u32), sz, u64, ctypes.POINTER(vp)]),
    "bfs_vm_trace_free": (None, [vp]),
    "bfs_vm_trace_size": (ci, [vp, ci, ctypes.POINTER(sz)]),
    "bfs_vm_trace_copy": (ci, [vp, ci, vp]),
    "bfs_xfe_scan": (ci, [ci, vp, vp, vp, vp, u64, ctypes.POINTER(u64), ctypes.POINTER(u64), ci, vp, ctypes.POINTER(u64)]),
    "bfs_xfe_scan_device_many": (ci, [vp, ctypes.c_uint32, vp]),
    "bfs_xfe_scan_device": (ci, [ci, vp, vp, vp, u64, vp, u64, ctypes.POINTER(u64), ctypes.POINTER(u64), ci, vp, u64, vp, ctypes.POINTER(u64), vp]),
    "bfs_poly_support": (ci, [vp, u64, u64, u32, ctypes.POINTER(u64), vp]),
    "bfs_poly_randomize": (ci, [vp, u64, u64, u32, u64, ctypes.POINTER(u64), vp]),
    "bfs_air_num_quotients": (ci, [ci]),
    "bfs_air_evaluate": (ci, [ci, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64),
                              ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64)]),
    "bfs_air_quotients": (ci, [ci, vp, vp, vp, u32, u64, u64, u64, u64, u64, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64), vp]),
    "bfs_difference_quotient": (ci, [vp, vp, vp, u32, u64, u64, vp]),
    "bfs_combination": (ci, [vp, u32, vp, ctypes.POINTER(u64), vp, u32, u64, u64, vp]),
    "bfs_air_combine": (ci, [ci, vp, vp, u32, u64, u64, u64, u64, u64, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64), vp, vp,
                             ctypes.POINTER(u64), vp, ctypes.POINTER(vp), vp]),
    "bfs_difference_combine": (ci, [vp, vp, u32, u64, u64, vp, vp, vp, vp]),
    "bfs_zerofier_inverses": (ci, [u32, u64, u64, u32, ctypes.POINTER(u32), ctypes.POINTER(u64), vp, vp]),
    "bfs_host_transpose": (ci, [vp, sz, sz, sz, vp, sz]),
    "bfs_air_combine_rows": (ci, [ci, vp, vp, u32, u64, u64, u64, u64, u64, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64), vp, vp,
                                  ctypes.POINTER(u64), vp, ctypes.POINTER(vp), u64, u64, vp]),
    "bfs_difference_combine_rows": (ci, [vp, vp, u32, u64, u64, vp, vp, vp, u64, u64, vp]),
    "bfs_zerofier_inverses_rows": (ci, [u32, u64, u64, u32, ctypes.POINTER(u32), ctypes.POINTER(u64), vp, u64, u64, vp]),
    "bfs_air_counts": (ci, [ci, ctypes.POINTER(ci)]),
    "bfs_stark_verify_begin": (ci, [vp, vp, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(ci)]),
    "bfs_stark_verify_finish": (ci, [vp, vp, ctypes.POINTER(u64), u32, ctypes.POINTER(ci)]),
    "bfs_stark_session_new": (vp, []),
    "bfs_stark_session_free": (None, [vp]),
    "bfs_stark_commit": (ci, [vp, vp, vp, vp, vp, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(ctypes.c_double), vp]),
    "bfs_stark_finish": (ci, [vp, vp, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64), u32, ctypes.c_int32, ctypes.POINTER(u64), u32,
                              ctypes.POINTER(u64), vp, ctypes.POINTER(u64), ctypes.POINTER(ctypes.c_double), vp]),
}


class GatherRequest(ctypes.Structure):
    """bfs_gather_request (include/bfstark.h)"""
    _fields_ = [("d_base", vp), ("nwords", u32), ("stride", u32), ("out_offset", u64)]


class RowColumn(ctypes.Structure):
    """bfs_row_column (include/bfstark.h)"""
    _fields_ = [("d_values", vp), ("is_ext", ctypes.c_int32), ("field_id", ctypes.c_int32)]


class TracePadTable(ctypes.Structure):
    """bfs_trace_pad_table (include/bfstark.h)"""
    _fields_ = [("d_rows", vp), ("rows", u64), ("row_stride", u64), ("height", u64), ("d_out", vp), ("d_mask0", vp), ("d_mask1", vp),
                ("d_mask2", vp), ("kind", ctypes.c_int32), ("width", u32)]


class CombSource(ctypes.Structure):
    """bfs_comb_source (include/bfstark.h)"""
    _fields_ = [("ptr", vp), ("is_ext", u32), ("pad", u32), ("shift", u64), ("wa", u64 * 3), ("wb", u64 * 3)]

class ScanSpec(ctypes.Structure):
    """bfs_scan_spec (include/bfstark.h)"""
    _fields_ = [("kind", ctypes.c_int32), ("record_before", ctypes.c_int32), ("d_x1", vp), ("d_x2", vp), ("d_x3", vp), ("shift1", u64),
                ("d_mask", vp), ("n", u64), ("constants", u64 * 12), ("initial", u64 * 3), ("d_out", vp), ("out_stride", u64),
                ("d_terminal", vp)]


class StarkParams(ctypes.Structure):
    """bfs_stark_params (include/bfstark.h)"""
    _fields_ = [("log_n", u32), ("expansion_factor", u32), ("num_colinearity_checks", u32), ("security_level", u32), ("offset", u64),
                ("omega", u64), ("max_degree", u64), ("heights", u64 * 3)]


class StarkVerifyParams(ctypes.Structure):
    """bfs_stark_verify_params (include/bfstark.h)"""
    _fields_ = [("log_n", u32), ("expansion_factor", u32), ("num_colinearity_checks", u32), ("security_level", u32), ("offset", u64),
                ("omega", u64), ("heights", u64 * 5), ("lengths", u64 * 5), ("omicrons", u64 * 5), ("num_distances", u32), ("pad", u32),
                ("distances", u64 * 8), ("program", vp), ("program_len", sz), ("input", vp), ("n_input", sz), ("output", vp), ("n_output", sz)]


class StarkTableIn(ctypes.Structure):
    """bfs_stark_table_in (include/bfstark.h)"""
    _fields_ = [("values", vp), ("rows", u64), ("row_stride", u64)]


class StarkRandomness(ctypes.Structure):
    """bfs_stark_randomness (include/bfstark.h)"""
    _fields_ = [("randomizer_seed", vp), ("randomizer_limbs", vp), ("base_randomizers", vp), ("base_salt_seed", vp), ("base_salts", vp),
                ("initials", u64 * 6), ("ext_randomizers", vp), ("ext_salt_seed", vp), ("ext_salts", vp)]


class CombWeight(ctypes.Structure):
    """bfs_comb_weight (include/bfstark.h)"""
    _fields_ = [("wa", u64 * 3), ("wb", u64 * 3), ("shift", u64)]


_lib = None


class BackendUnavailable(RuntimeError):
    pass


def exported_symbols():
    """names include/bfstark.h declares (used by the CPU test that checks the .so exports all of them)."""
    return sorted(_SIGNATURES)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendUnavailable(
            "libbfstark_hip.so is not built (%s). Run `python -m stark_brainfuck_amd.build`; "
            "there is no CPU fallback for the hot path." % LIB_PATH)
    try:
        # torch (used for streams / distributed plumbing) ships its own copy of the HIP runtime.  Load it FIRST so that this
        # library binds to the runtime already in the process: two HIP runtimes initialised in one process leave the second
        # one without a device ("no ROCm-capable device is detected").
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc == BFS_OK:
        return
    msg = load().bfs_last_error().decode("utf-8", "replace")
    if rc in ASSERT_CODES:
        raise AssertionError(msg)
    raise RuntimeError("libbfstark_hip: error %d: %s" % (rc, msg))

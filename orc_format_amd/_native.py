"""ctypes binding of liborcd.so -- the C ABI declared in include/orcd.h.

There is no fallback: if the library is missing, or there is no GPU when a decode is asked for,
the call raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liborcd.so")

OK, OUT_OF_SPEC, INVALID_UTF8, INVALID_COLUMN, INVALID_KIND, DECODE_FLOAT, DECOMPRESSION, \
    INVALID_PROTO = range(8)
UNSUPPORTED, HIP_ERROR, INVALID_ARGUMENT, NO_MEMORY = 100, 101, 102, 103

SIGNED, IN_DEVICE, OUT_DEVICE, CHECK_UTF8, SCATTER_CARRY, ROW_ALIGNED, ENTRIES_HOST, LENGTH_RLE_V1 = 1, 2, 4, 8, 16, 32, 64, 128


class OrcdRun(C.Structure):
    _fields_ = [("payload_offset", C.c_uint64), ("out_offset", C.c_uint64), ("kind", C.c_uint32),
                ("bit_width", C.c_uint32), ("length", C.c_uint32), ("reserved", C.c_uint32)]


class OrcdByteRun(C.Structure):
    _fields_ = [("payload_offset", C.c_uint64), ("out_offset", C.c_uint64), ("length", C.c_uint32),
                ("is_run", C.c_uint32)]


WHOLE_STREAM = C.c_size_t(-1).value


class StripeInfo(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("index_length", C.c_uint64), ("data_length", C.c_uint64),
                ("footer_length", C.c_uint64), ("number_of_rows", C.c_uint64)]


class StreamInfo(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("column", C.c_uint32), ("length", C.c_uint64),
                ("offset", C.c_uint64)]


class Rlev2Slice(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("length", C.c_uint64), ("num_values", C.c_uint64),
                ("entry_points", C.c_void_p), ("n_entries", C.c_size_t)]


class ColumnData(C.Structure):
    _fields_ = [("type_kind", C.c_int), ("on_device", C.c_int), ("row_aligned", C.c_int),
                ("num_rows", C.c_uint64), ("num_values", C.c_uint64), ("num_slots", C.c_uint64),
                ("validity", C.c_void_p), ("values", C.c_void_p), ("value_size", C.c_size_t),
                ("offsets", C.c_void_p), ("bytes", C.c_void_p), ("num_bytes", C.c_uint64),
                ("owner", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/orcd.h declares
P, SZ, U64P = C.c_void_p, C.c_size_t, C.c_void_p
SIGNATURES = {
    "orcd_strerror": (C.c_char_p, [C.c_int]),
    "orcd_ctx_create": (C.c_int, [C.c_int, C.POINTER(P)]),
    "orcd_ctx_destroy": (C.c_int, [P]),
    "orcd_ctx_set_stream": (C.c_int, [P, P]),
    "orcd_ctx_synchronize": (C.c_int, [P]),
    "orcd_ctx_release_cached": (C.c_int, [P]),
    "orcd_ctx_set_cache_limit": (C.c_int, [P, SZ, SZ]),
    "orcd_ctx_copy_to_host": (C.c_int, [P, P, P, SZ]),
    "orcd_ctx_last_elapsed_ms": (C.c_float, [P]),
    "orcd_ctx_phase_ms": (C.c_float, [P, C.c_int]),
    "orcd_ctx_last_hip_error": (C.c_char_p, [P]),
    "orcd_ctx_last_path": (C.c_int, [P]),
    "orcd_ctx_last_value_kernel": (C.c_char_p, [P]),
    "orcd_helper_threads": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "orcd_ctx_last_stripe_ms": (C.c_int, [P, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "orcd_rlev2_decode": (C.c_int, [P, P, SZ, SZ, U64P, SZ, P, C.c_uint]),
    "orcd_rlev2_decode_groups": (C.c_int, [P, P, SZ, SZ, U64P, SZ, C.c_uint64, P, C.c_uint]),
    "orcd_rlev1_decode": (C.c_int, [P, P, SZ, SZ, U64P, SZ, P, C.c_uint]),
    "orcd_inflate_chunks": (C.c_int, [P, P, SZ, P, SZ, P, SZ, P, C.c_uint]),
    "orcd_rlev2_decode_noalloc": (C.c_int, [P, P, SZ, SZ, U64P, SZ, P, C.c_uint]),
    "orcd_rlev2_decode_batch": (C.c_int, [P, P, SZ, C.POINTER(Rlev2Slice), SZ, P, C.POINTER(SZ), C.c_uint]),
    "orcd_boolean_decode_batch": (C.c_int, [P, P, SZ, C.POINTER(Rlev2Slice), SZ, P, C.POINTER(C.c_uint64),
                                            C.POINTER(SZ), C.c_uint]),
    "orcd_rlev2_runs": (C.c_int, [P, P, SZ, SZ, U64P, SZ, C.POINTER(OrcdRun), SZ, C.POINTER(SZ),
                                  C.c_uint]),
    "orcd_byte_rle_runs": (C.c_int, [P, P, SZ, SZ, U64P, SZ, C.POINTER(OrcdByteRun), SZ, C.POINTER(SZ),
                                     C.POINTER(C.c_uint64), C.c_uint]),
    "orcd_byte_rle_decode": (C.c_int, [P, P, SZ, SZ, U64P, SZ, P, C.c_uint]),
    "orcd_boolean_decode": (C.c_int, [P, P, SZ, SZ, U64P, SZ, P, P, C.POINTER(C.c_uint64),
                                      C.c_uint]),
    "orcd_float_decode": (C.c_int, [P, P, SZ, SZ, SZ, P, C.c_uint]),
    "orcd_string_direct_decode": (C.c_int, [P, P, SZ, U64P, SZ, P, SZ, SZ, P, P, C.c_uint]),
    "orcd_string_dictionary_decode": (C.c_int, [P, P, P, SZ, P, SZ, P, P, SZ, C.c_uint]),
    "orcd_file_open": (C.c_int, [C.c_char_p, C.POINTER(P)]),
    "orcd_file_open_memory": (C.c_int, [P, SZ, C.POINTER(P)]),
    "orcd_file_close": (None, [P]),
    "orcd_file_num_stripes": (SZ, [P]),
    "orcd_file_stripe": (C.c_int, [P, SZ, C.POINTER(StripeInfo)]),
    "orcd_file_compression": (C.c_int, [P]),
    "orcd_file_number_of_rows": (C.c_uint64, [P]),
    "orcd_file_row_index_stride": (C.c_uint32, [P]),
    "orcd_file_num_types": (SZ, [P]),
    "orcd_file_type_kind": (C.c_int, [P, SZ]),
    "orcd_read_stripe_footer": (C.c_int, [P, SZ, C.POINTER(P)]),
    "orcd_stripe_footer_free": (None, [P]),
    "orcd_stripe_footer_num_streams": (SZ, [P]),
    "orcd_stripe_footer_stream": (C.c_int, [P, SZ, C.POINTER(StreamInfo)]),
    "orcd_read_stripe_column": (C.c_int, [P, SZ, P, C.c_uint32, C.POINTER(P)]),
    "orcd_column_free": (None, [P]),
    "orcd_column_get_stream": (C.c_int, [P, C.c_int, C.POINTER(P), C.POINTER(SZ)]),
    "orcd_column_encoding": (C.c_int, [P, C.POINTER(C.c_int), C.POINTER(C.c_uint32)]),
    "orcd_column_number_of_rows": (C.c_uint64, [P]),
    "orcd_column_entry_points": (C.c_int, [P, P, C.c_int, C.POINTER(P), C.POINTER(SZ)]),
    "orcd_column_values_per_group": (C.c_uint64, [P, P, C.c_int]),
    "orcd_stripe_column_open": (C.c_int, [P, SZ, P, C.c_uint32, C.POINTER(P)]),
    "orcd_column_region": (C.c_int, [P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "orcd_column_stream_range": (C.c_int, [P, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "orcd_file_read_at": (C.c_int, [P, C.c_uint64, C.c_uint64, P]),
    "orcd_scatter_valid": (C.c_int, [P, P, SZ, P, SZ, SZ, P, C.c_uint]),
    "orcd_decode_column": (C.c_int, [P, P, SZ, P, C.c_uint32, C.c_uint, C.POINTER(ColumnData)]),
    "orcd_column_data_free": (None, [C.POINTER(ColumnData)]),
    "orcd_decode_stripe": (C.c_int, [P, P, SZ, P, C.POINTER(C.c_uint32), SZ, C.c_uint, C.POINTER(ColumnData)]),
    "orcd_stripe_prefetch": (C.c_int, [P, P, SZ, P, C.POINTER(C.c_uint32), SZ]),
}

_lib = None


def lib():
    """The loaded library.  torch is imported first when available so that both share one HIP
    runtime (same SONAME) instead of loading a second copy."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s is missing: build it with `python -m orc_format_amd.build` "
                "(there is no CPU fallback)" % LIB_PATH)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib

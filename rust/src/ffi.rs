//! Raw bindings of include/orcd.h (the decode entry points).  UNCOMPILED: written against the
//! header by hand; the build environment has no Rust toolchain.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_float, c_int, c_uint, c_void};

#[repr(C)]
pub struct orcd_ctx {
    _private: [u8; 0],
}

pub const ORCD_OK: c_int = 0;
pub const ORCD_OUT_OF_SPEC: c_int = 1;
pub const ORCD_INVALID_UTF8: c_int = 2;
pub const ORCD_INVALID_COLUMN: c_int = 3;
pub const ORCD_INVALID_KIND: c_int = 4;
pub const ORCD_DECODE_FLOAT: c_int = 5;
pub const ORCD_DECOMPRESSION: c_int = 6;
pub const ORCD_INVALID_PROTO: c_int = 7;
pub const ORCD_UNSUPPORTED: c_int = 100;
pub const ORCD_HIP_ERROR: c_int = 101;
pub const ORCD_INVALID_ARGUMENT: c_int = 102;
pub const ORCD_NO_MEMORY: c_int = 103;

pub const ORCD_SIGNED: c_uint = 1;
pub const ORCD_IN_DEVICE: c_uint = 2;
pub const ORCD_OUT_DEVICE: c_uint = 4;
pub const ORCD_CHECK_UTF8: c_uint = 8;

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct orcd_run {
    pub payload_offset: u64,
    pub out_offset: u64,
    pub kind: u32,
    pub bit_width: u32,
    pub length: u32,
    pub reserved: u32,
}

extern "C" {
    pub fn orcd_strerror(status: c_int) -> *const c_char;
    pub fn orcd_ctx_create(device_id: c_int, ctx: *mut *mut orcd_ctx) -> c_int;
    pub fn orcd_ctx_destroy(ctx: *mut orcd_ctx) -> c_int;
    pub fn orcd_ctx_set_stream(ctx: *mut orcd_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn orcd_ctx_synchronize(ctx: *mut orcd_ctx) -> c_int;
    pub fn orcd_ctx_last_elapsed_ms(ctx: *mut orcd_ctx) -> c_float;
    pub fn orcd_rlev2_decode(ctx: *mut orcd_ctx, stream: *const u8, len: usize, num_values: usize,
                             entry_points: *const u64, n_entries: usize, out: *mut i64,
                             flags: c_uint) -> c_int;
    pub fn orcd_rlev2_runs(ctx: *mut orcd_ctx, stream: *const u8, len: usize, num_values: usize,
                           entry_points: *const u64, n_entries: usize, runs: *mut orcd_run,
                           max_runs: usize, n_runs: *mut usize, flags: c_uint) -> c_int;
    pub fn orcd_byte_rle_decode(ctx: *mut orcd_ctx, stream: *const u8, len: usize, num_bytes: usize,
                                entry_points: *const u64, n_entries: usize, out: *mut u8,
                                flags: c_uint) -> c_int;
    pub fn orcd_boolean_decode(ctx: *mut orcd_ctx, stream: *const u8, len: usize, num_bits: usize,
                               entry_points: *const u64, n_entries: usize, out_bytes: *mut u8,
                               out_bitmap: *mut u8, set_bits: *mut u64, flags: c_uint) -> c_int;
    pub fn orcd_float_decode(ctx: *mut orcd_ctx, stream: *const u8, len: usize, elem_size: usize,
                             num_values: usize, out: *mut c_void, flags: c_uint) -> c_int;
    pub fn orcd_string_direct_decode(ctx: *mut orcd_ctx, length_stream: *const u8, length_len: usize,
                                     length_entry_points: *const u64, n_length_entries: usize,
                                     data: *const u8, data_len: usize, num_values: usize,
                                     offsets: *mut i64, bytes_out: *mut u8, flags: c_uint) -> c_int;
    pub fn orcd_string_dictionary_decode(ctx: *mut orcd_ctx, dict_offsets: *const i64,
                                         dict_bytes: *const u8, dict_size: usize,
                                         indices: *const i64, num_values: usize,
                                         offsets_out: *mut i64, bytes_out: *mut u8,
                                         bytes_capacity: usize, flags: c_uint) -> c_int;
    pub fn orcd_rlev2_decode_batch(ctx: *mut orcd_ctx, buffer: *const u8, buffer_len: usize,
                                   slices: *const orcd_rlev2_slice, n_slices: usize, out: *mut i64,
                                   bad_slice: *mut usize, flags: c_uint) -> c_int;
    pub fn orcd_boolean_decode_batch(ctx: *mut orcd_ctx, buffer: *const u8, buffer_len: usize,
                                     slices: *const orcd_rlev2_slice, n_slices: usize,
                                     out_bytes: *mut u8, set_bits: *mut u64, bad_slice: *mut usize,
                                     flags: c_uint) -> c_int;
    pub fn orcd_scatter_valid(ctx: *mut orcd_ctx, validity: *const u8, num_rows: usize,
                              dense: *const c_void, num_values: usize, elem_size: usize,
                              out: *mut c_void, flags: c_uint) -> c_int;
}

/// One stream inside the buffer handed to `orcd_rlev2_decode_batch` (include/orcd.h).
#[repr(C)]
pub struct orcd_rlev2_slice {
    pub offset: u64,
    pub length: u64,
    pub num_values: u64,
    pub entry_points: *const u64,
    pub n_entries: usize,
}

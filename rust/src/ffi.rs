//! Raw bindings of include/orcd.h, every entry point of it.  UNCOMPILED: written against the
//! header by hand; the build environment has no Rust toolchain.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_float, c_int, c_uint, c_void};

#[repr(C)]
pub struct orcd_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct orcd_file {
    _private: [u8; 0],
}
#[repr(C)]
pub struct orcd_stripe_footer {
    _private: [u8; 0],
}
#[repr(C)]
pub struct orcd_column {
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
pub const ORCD_SCATTER_CARRY: c_uint = 16;
pub const ORCD_ROW_ALIGNED: c_uint = 32;
pub const ORCD_ENTRIES_HOST: c_uint = 64;
pub const ORCD_LENGTH_RLE_V1: c_uint = 128;
pub const ORCD_NOALLOC_MAX_VALUES: usize = 16777216;
pub const ORCD_NOALLOC_MAX_ENTRIES: usize = 65536;
pub const ORCD_NOALLOC_MAX_RUNS: usize = 524288;
pub const ORCD_PATH_GENERAL: c_int = 0;
pub const ORCD_PATH_STRIDED: c_int = 1;
pub const ORCD_PATH_PLANNED: c_int = 2;

// stream kinds (src/proto.rs:127-136) and column encodings (src/proto.rs:149-154)
pub const ORCD_STREAM_PRESENT: c_int = 0;
pub const ORCD_STREAM_DATA: c_int = 1;
pub const ORCD_STREAM_LENGTH: c_int = 2;
pub const ORCD_STREAM_DICTIONARY_DATA: c_int = 3;
pub const ORCD_STREAM_DICTIONARY_COUNT: c_int = 4;
pub const ORCD_STREAM_SECONDARY: c_int = 5;
pub const ORCD_STREAM_ROW_INDEX: c_int = 6;
pub const ORCD_STREAM_BLOOM_FILTER: c_int = 7;
pub const ORCD_ENC_DIRECT: c_int = 0;
pub const ORCD_ENC_DICTIONARY: c_int = 1;
pub const ORCD_ENC_DIRECT_V2: c_int = 2;
pub const ORCD_ENC_DICTIONARY_V2: c_int = 3;

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

/// One compression chunk for orcd_inflate_chunks (include/orcd.h; a chunk of
/// src/read/decompress/mod.rs:60-80 by its place in the source and in the output).
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct orcd_chunk {
    pub src_offset: u64,
    pub dst_offset: u64,
    pub src_length: u32,
    pub dst_capacity: u32,
}

/// One run of a byte-RLE stream (include/orcd.h; BooleanRun of src/read/decode/boolean_rle.rs:7-12
/// as positions in the stream and in the decoded bytes).
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct orcd_byte_run {
    pub payload_offset: u64,
    pub out_offset: u64,
    pub length: u32,
    pub is_run: u32,
}

/// `num_bytes` of `orcd_byte_rle_runs` that asks for every run of the stream.
pub const ORCD_WHOLE_STREAM: usize = usize::MAX;

/// proto::StripeInformation (src/proto.rs:206-217)
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct orcd_stripe_info {
    pub offset: u64,
    pub index_length: u64,
    pub data_length: u64,
    pub footer_length: u64,
    pub number_of_rows: u64,
}

/// proto::Stream (src/proto.rs:113-137) and its offset from the start of the stripe
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct orcd_stream_info {
    pub kind: u32,
    pub column: u32,
    pub length: u64,
    pub offset: u64,
}

/// One decoded column of one stripe; the arrays belong to the structure until
/// `orcd_column_data_free` (include/orcd.h).
#[repr(C)]
pub struct orcd_column_data {
    pub type_kind: c_int,
    pub on_device: c_int,
    pub row_aligned: c_int,
    pub num_rows: u64,
    pub num_values: u64,
    pub num_slots: u64,
    pub validity: *mut u8,
    pub values: *mut c_void,
    pub value_size: usize,
    pub offsets: *mut i64,
    pub bytes: *mut u8,
    pub num_bytes: u64,
    pub owner: *mut orcd_ctx,
}

extern "C" {
    pub fn orcd_strerror(status: c_int) -> *const c_char;
    pub fn orcd_ctx_create(device_id: c_int, ctx: *mut *mut orcd_ctx) -> c_int;
    pub fn orcd_ctx_destroy(ctx: *mut orcd_ctx) -> c_int;
    pub fn orcd_ctx_set_stream(ctx: *mut orcd_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn orcd_ctx_synchronize(ctx: *mut orcd_ctx) -> c_int;
    pub fn orcd_ctx_release_cached(ctx: *mut orcd_ctx) -> c_int;
    pub fn orcd_ctx_set_cache_limit(ctx: *mut orcd_ctx, device_bytes: usize, host_bytes: usize) -> c_int;
    pub fn orcd_ctx_copy_to_host(ctx: *mut orcd_ctx, dst: *mut c_void, src_device: *const c_void,
                                 bytes: usize) -> c_int;
    pub fn orcd_ctx_last_elapsed_ms(ctx: *mut orcd_ctx) -> c_float;
    pub fn orcd_ctx_phase_ms(ctx: *mut orcd_ctx, phase: c_int) -> c_float;
    pub fn orcd_ctx_last_path(ctx: *mut orcd_ctx) -> c_int;
    pub fn orcd_ctx_last_value_kernel(ctx: *mut orcd_ctx) -> *const c_char;
    pub fn orcd_ctx_last_stripe_ms(ctx: *mut orcd_ctx, feed_ms: *mut f32, decode_ms: *mut f32) -> c_int;
    pub fn orcd_helper_threads(budget: *mut c_int, in_use: *mut c_int, peak: *mut c_int) -> c_int;
    pub fn orcd_ctx_last_hip_error(ctx: *mut orcd_ctx) -> *const c_char;
    pub fn orcd_rlev2_decode(ctx: *mut orcd_ctx, stream: *const u8, len: usize, num_values: usize,
                             entry_points: *const u64, n_entries: usize, out: *mut i64,
                             flags: c_uint) -> c_int;
    pub fn orcd_rlev2_decode_groups(ctx: *mut orcd_ctx, stream: *const u8, len: usize, num_values: usize,
                                    entry_points: *const u64, n_entries: usize, values_per_group: u64,
                                    out: *mut i64, flags: c_uint) -> c_int;
    pub fn orcd_inflate_chunks(ctx: *mut orcd_ctx, src: *const u8, src_len: usize,
                               chunks: *const orcd_chunk, n_chunks: usize, dst: *mut u8,
                               dst_len: usize, sizes: *mut u32, flags: c_uint) -> c_int;
    pub fn orcd_rlev1_decode(ctx: *mut orcd_ctx, stream: *const u8, len: usize, num_values: usize,
                             entry_points: *const u64, n_entries: usize, out: *mut i64,
                             flags: c_uint) -> c_int;
    pub fn orcd_rlev2_decode_noalloc(ctx: *mut orcd_ctx, stream: *const u8, len: usize,
                                     num_values: usize, entry_points: *const u64, n_entries: usize,
                                     out: *mut i64, flags: c_uint) -> c_int;
    pub fn orcd_rlev2_runs(ctx: *mut orcd_ctx, stream: *const u8, len: usize, num_values: usize,
                           entry_points: *const u64, n_entries: usize, runs: *mut orcd_run,
                           max_runs: usize, n_runs: *mut usize, flags: c_uint) -> c_int;
    pub fn orcd_byte_rle_runs(ctx: *mut orcd_ctx, stream: *const u8, len: usize, num_bytes: usize,
                              entry_points: *const u64, n_entries: usize, runs: *mut orcd_byte_run,
                              max_runs: usize, n_runs: *mut usize, total_bytes: *mut u64,
                              flags: c_uint) -> c_int;
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

    // host side: file tail, stripe footers, column regions (read::read_metadata,
    // read_stripe_footer, read_stripe_column, Column -- src/read/mod.rs:46-159, column.rs:38-78)
    pub fn orcd_file_open(path: *const c_char, file: *mut *mut orcd_file) -> c_int;
    pub fn orcd_file_open_memory(bytes: *const u8, len: usize, file: *mut *mut orcd_file) -> c_int;
    pub fn orcd_file_close(file: *mut orcd_file);
    pub fn orcd_file_num_stripes(file: *const orcd_file) -> usize;
    pub fn orcd_file_stripe(file: *const orcd_file, stripe: usize, info: *mut orcd_stripe_info) -> c_int;
    pub fn orcd_file_compression(file: *const orcd_file) -> c_int;
    pub fn orcd_file_number_of_rows(file: *const orcd_file) -> u64;
    pub fn orcd_file_row_index_stride(file: *const orcd_file) -> u32;
    pub fn orcd_file_num_types(file: *const orcd_file) -> usize;
    pub fn orcd_file_type_kind(file: *const orcd_file, type_id: usize) -> c_int;
    pub fn orcd_read_stripe_footer(file: *mut orcd_file, stripe: usize,
                                   footer: *mut *mut orcd_stripe_footer) -> c_int;
    pub fn orcd_stripe_footer_free(footer: *mut orcd_stripe_footer);
    pub fn orcd_stripe_footer_num_streams(footer: *const orcd_stripe_footer) -> usize;
    pub fn orcd_stripe_footer_stream(footer: *const orcd_stripe_footer, i: usize,
                                     info: *mut orcd_stream_info) -> c_int;
    pub fn orcd_read_stripe_column(file: *mut orcd_file, stripe: usize,
                                   footer: *const orcd_stripe_footer, column: u32,
                                   out: *mut *mut orcd_column) -> c_int;
    pub fn orcd_column_free(column: *mut orcd_column);
    pub fn orcd_column_get_stream(column: *const orcd_column, kind: c_int, bytes: *mut *const u8,
                                  len: *mut usize) -> c_int;
    pub fn orcd_column_encoding(column: *const orcd_column, kind: *mut c_int,
                                dictionary_size: *mut u32) -> c_int;
    pub fn orcd_column_number_of_rows(column: *const orcd_column) -> u64;
    pub fn orcd_stripe_column_open(file: *mut orcd_file, stripe: usize,
                                   footer: *const orcd_stripe_footer, column: u32,
                                   out: *mut *mut orcd_column) -> c_int;
    pub fn orcd_column_region(column: *const orcd_column, offset: *mut u64, len: *mut u64) -> c_int;
    pub fn orcd_column_stream_range(column: *const orcd_column, kind: c_int, offset: *mut u64,
                                    len: *mut u64) -> c_int;
    pub fn orcd_file_read_at(file: *mut orcd_file, offset: u64, len: u64, dst: *mut u8) -> c_int;
    pub fn orcd_column_entry_points(file: *mut orcd_file, column: *mut orcd_column, kind: c_int,
                                    entries: *mut *const u64, n_entries: *mut usize) -> c_int;
    pub fn orcd_column_values_per_group(file: *mut orcd_file, column: *mut orcd_column, kind: c_int) -> u64;

    // whole columns (the deserialize_* functions of tests/it/deserialize.rs:13-222)
    pub fn orcd_decode_column(ctx: *mut orcd_ctx, file: *mut orcd_file, stripe: usize,
                              footer: *const orcd_stripe_footer, column: u32, flags: c_uint,
                              out: *mut orcd_column_data) -> c_int;
    pub fn orcd_decode_stripe(ctx: *mut orcd_ctx, file: *mut orcd_file, stripe: usize,
                              footer: *const orcd_stripe_footer, columns: *const u32,
                              n_columns: usize, flags: c_uint, out: *mut orcd_column_data) -> c_int;
    pub fn orcd_stripe_prefetch(ctx: *mut orcd_ctx, file: *mut orcd_file, stripe: usize,
                                footer: *const orcd_stripe_footer, columns: *const u32,
                                n_columns: usize) -> c_int;
    pub fn orcd_column_data_free(data: *mut orcd_column_data);
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

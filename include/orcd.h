/*
 * orcd.h -- C ABI of the MI355X ORC stripe decoder (liborcd.so).
 *
 * The `orc-format` crate has no FFI layer: its decode boundary is the Rust API re-exported at
 * src/read/decode/mod.rs:8-14 plus Column::get_stream (src/read/column.rs:38).  Every entry point
 * below names the item of that API it replaces.  The crate's iterators pull values lazily from a
 * `Read`; the entry points here are batch calls over one whole stream slice (what
 * Column::get_stream returns), which is equivalent because every stream has its own reader.
 *
 * Conventions
 *  - All functions return an ORCD_* status.  1..7 are the variants of `enum Error`
 *    (src/error.rs:6-21) in declaration order; 100+ are conditions the crate answers with a panic.
 *  - Pointers are plain host or device addresses as said by the ORCD_IN_DEVICE / ORCD_OUT_DEVICE
 *    flags; the caller owns every input and output buffer.  The library owns only the context's
 *    arena (run tables, staging), which grows on demand and is released by orcd_ctx_destroy.
 *  - A context belongs to one thread and one GPU.  Work is enqueued on the context's HIP stream;
 *    a decode call returns once its status is final (all validation happens in the header-scan
 *    kernels, which the call waits for).  With ORCD_OUT_DEVICE the value kernels may still be
 *    running on the stream when the call returns; with host outputs the call is synchronous.
 *  - The context's stream is not ordered against any other stream (it is created non-blocking):
 *    device buffers that work on another stream writes or still reads -- inputs being prepared,
 *    outputs being filled or consumed -- must be finished with before the call, or the context
 *    must be put on that stream (orcd_ctx_set_stream).
 *  - Decoding stops after `num_values` values; trailing stream bytes are ignored.
 */
#ifndef ORCD_H
#define ORCD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (src/error.rs:6-21) ------------------------------------------------------ */
enum {
    ORCD_OK = 0,
    ORCD_OUT_OF_SPEC = 1,
    ORCD_INVALID_UTF8 = 2,
    ORCD_INVALID_COLUMN = 3,
    ORCD_INVALID_KIND = 4,
    ORCD_DECODE_FLOAT = 5,
    ORCD_DECOMPRESSION = 6,
    ORCD_INVALID_PROTO = 7,
    ORCD_UNSUPPORTED = 100, /* the crate panics with todo!() here (e.g. LZO-compressed files, timestamp columns) */
    ORCD_HIP_ERROR = 101,
    ORCD_INVALID_ARGUMENT = 102,
    ORCD_NO_MEMORY = 103 /* host memory ran out inside the library */
};

/* ---- flags ---------------------------------------------------------------------------------- */
enum {
    ORCD_SIGNED = 1u,      /* RLE v2: zigzag-decode values (SignedRleV2Iter) */
    ORCD_IN_DEVICE = 2u,   /* stream / entry-point pointers are device memory */
    ORCD_OUT_DEVICE = 4u,  /* output pointers are device memory */
    ORCD_CHECK_UTF8 = 8u,  /* strings: validate each value (tests/it/deserialize.rs:116-118) */
    ORCD_SCATTER_CARRY = 16u, /* orcd_scatter_valid: null rows repeat the next value (offsets) */
    ORCD_ROW_ALIGNED = 32u, /* orcd_decode_column: one slot per row instead of per non-null row */
    ORCD_ENTRIES_HOST = 64u, /* with ORCD_IN_DEVICE: the entry points are host memory all the same */
    ORCD_LENGTH_RLE_V1 = 128u /* orcd_string_direct_decode: the LENGTH stream is integer RLE version 1 (ORC 0.11 files) */
};

/* ---- stream kinds (src/proto.rs:127-136) and column encodings (src/proto.rs:149-154) -------- */
enum {
    ORCD_STREAM_PRESENT = 0, ORCD_STREAM_DATA = 1, ORCD_STREAM_LENGTH = 2,
    ORCD_STREAM_DICTIONARY_DATA = 3, ORCD_STREAM_DICTIONARY_COUNT = 4, ORCD_STREAM_SECONDARY = 5,
    ORCD_STREAM_ROW_INDEX = 6, ORCD_STREAM_BLOOM_FILTER = 7
};
enum { ORCD_ENC_DIRECT = 0, ORCD_ENC_DICTIONARY = 1, ORCD_ENC_DIRECT_V2 = 2, ORCD_ENC_DICTIONARY_V2 = 3 };

typedef struct orcd_ctx orcd_ctx;

/* Human-readable name of a status code. */
const char *orcd_strerror(int status);

/* ---- context -------------------------------------------------------------------------------- */

/* Creates a decode context on GPU `device_id` with a HIP stream of its own.  Stands in for the
 * `scratch: Vec<u8>` every reader of the crate takes and returns (src/read/column.rs:38,76;
 * src/read/decode/rle_v2.rs:488-500): device buffers are recycled between calls. */
int orcd_ctx_create(int device_id, orcd_ctx **ctx);
int orcd_ctx_destroy(orcd_ctx *ctx);
/* Makes the context enqueue on `hip_stream` (a hipStream_t, e.g. torch's current stream) instead
 * of its own stream; NULL restores the context's stream. */
int orcd_ctx_set_stream(orcd_ctx *ctx, void *hip_stream);
/* Blocks until everything enqueued by this context has finished. */
int orcd_ctx_synchronize(orcd_ctx *ctx);
/* Gives the device memory the context keeps for its next column calls (released result arrays and
 * temporaries of orcd_decode_column / orcd_decode_stripe) back to the runtime; waits for the
 * context's queued work first.  Nothing else changes: the next call allocates afresh. */
int orcd_ctx_release_cached(orcd_ctx *ctx);
/* How many bytes of released result arrays and temporaries of the column calls the context keeps
 * for its next calls instead of handing them back to the runtime: device memory, and pinned host
 * memory (defaults 16 GiB and 8 GiB per context; 0 keeps nothing).  Takes effect from the next release on;
 * orcd_ctx_release_cached empties what is held now. */
int orcd_ctx_set_cache_limit(orcd_ctx *ctx, size_t device_bytes, size_t host_bytes);
/* Copies `bytes` bytes from device memory -- typically one array of an orcd_column_data that was
 * asked for with ORCD_OUT_DEVICE, such as the offsets of a string column whose bytes stay on the
 * GPU -- to host memory, on the context's stream, and waits for it.  (The reference keeps
 * everything in host Vecs, src/read/decode/variable_length.rs:5-28; this is the piece a caller
 * needs that keeps the large arrays on the GPU.) */
int orcd_ctx_copy_to_host(orcd_ctx *ctx, void *dst, const void *src_device, size_t bytes);
/* Duration in milliseconds of the GPU work of the most recent decode call on this context,
 * measured with HIP events on the context's stream (waits for that work).  Negative on error. */
float orcd_ctx_last_elapsed_ms(orcd_ctx *ctx);
/* The same, per phase of the most recent decode call: 0 the whole call, 1 header count + segment
 * prefix sums (including the read-back the host waits for), 2 run-table emission, 3 value
 * expansion (for strings: LENGTH sums, their scan and the expansion into offsets), 4 string
 * bytes (copy + UTF-8 validation).  Negative when that phase did not run. */
float orcd_ctx_phase_ms(orcd_ctx *ctx, int phase);
/* Which way the most recent orcd_rlev2_decode call on this context went: ORCD_PATH_GENERAL (header
 * scan, run table, value expansion) or ORCD_PATH_STRIDED (a column whose every 512 values are one
 * Direct run, with equally many equal-sized runs between two entry points, is expanded by a single
 * launch that checks each run header as it goes) or ORCD_PATH_PLANNED (a column whose every 512
 * values are one run of any kind -- full Delta runs of sorted keys, say -- with equally many runs
 * between two entry points, or a known number of values (orcd_rlev2_decode_groups): every run
 * header is found by interpolation between the entry points and proven by its links to its
 * neighbours, in the launch that expands the values or in one ahead of it, without prefix sums or
 * a wait in between).  The library tries the one that worked
 * for the context's last column first, then the other, and falls back to the general path.
 * The results are the same either way; this is for benchmarks and profiles. */
enum { ORCD_PATH_GENERAL = 0, ORCD_PATH_STRIDED = 1, ORCD_PATH_PLANNED = 2 };
int orcd_ctx_last_path(orcd_ctx *ctx);
/* Name of the kernel that expanded the values of the most recent orcd_rlev2_decode /
 * orcd_rlev2_decode_groups call on this context, as a profiler lists it (without the orcd::
 * namespace): the two planned ways and the two variants of the general kernel have one
 * ORCD_PATH_* each.  A string the library owns; "" before the first call.  For benchmarks. */
const char *orcd_ctx_last_value_kernel(orcd_ctx *ctx);
/* Where the time of the most recent successful orcd_decode_stripe call on this context went, by the
 * host's clock: `feed_ms` until the bytes of the stripe's columns were read and their copies to
 * the device queued (next to nothing when orcd_stripe_prefetch had them there already), `decode_ms`
 * from then to the end of the call, every kernel finished -- decoding, the waits for the header
 * totals and counts in between, and handing the arrays over.  ORCD_INVALID_ARGUMENT before the
 * first such call.  ORCD_TRACE_STRIPE=1 in the environment prints the same and more (a line per
 * stage and column) on stderr, with a wait for the GPU after every stage.  For benchmarks. */
int orcd_ctx_last_stripe_ms(orcd_ctx *ctx, float *feed_ms, float *decode_ms);
/* The host threads the library starts beside its callers' -- readers and copiers of a stripe's
 * bytes, the decompression threads of a compressed file -- are one pool for the whole process,
 * however many contexts and open files there are: as many threads as ORCD_HELPER_THREADS says
 * when it is set, else the CPUs the process may use (OMP_NUM_THREADS where a launcher sets it,
 * else the hardware's count) divided by the ranks of the node (LOCAL_WORLD_SIZE, else
 * WORLD_SIZE), at least 2, at most 64.  An operation works on its calling thread and is joined
 * by the pool's idle threads, which go to whichever operation has the fewest: callers that feed
 * stripes side by side share them evenly.  Reports the size of the pool, the helpers at work this
 * moment and the most there ever were at once (any pointer may be null).  The threads bind
 * themselves to the CPUs of the NUMA node that the process's first context's GPU is attached to,
 * where /sys says which (ORCD_NUMA_BIND=0: never).  No counterpart in the crate, which is
 * single-threaded. */
int orcd_helper_threads(int *budget, int *in_use, int *peak);
/* Text of the last HIP runtime error seen by this context (after ORCD_HIP_ERROR). */
const char *orcd_ctx_last_hip_error(orcd_ctx *ctx);

/* ---- RLE v2 integers ------------------------------------------------------------------------ */

/* Replaces SignedRleV2Iter::new(reader, length, scratch) / UnsignedRleV2Iter::new (and the run
 * iterators they flatten): src/read/decode/rle_v2.rs:480-562, 677-758.
 *
 * Decodes `num_values` integers of an RLE v2 stream into out[0..num_values) as i64 (unsigned
 * streams deliver the same 64 bits).  ShortRepeat, Direct, Delta and PatchedBase runs of every
 * bit width of the ORC v1 specification are decoded.
 *
 * `entry_points` (nullable): ascending byte offsets inside the stream at which a run header
 * starts -- for instance the stream positions of the column's ROW_INDEX entries.  They only add
 * parallelism to the header scan (one lane walks the headers between two consecutive entries).
 * The first entry is taken to be offset 0 whatever its value.  An entry that is not a run
 * boundary yields ORCD_OUT_OF_SPEC.
 *
 * After any error return the contents of out[] are undefined: the paths that decode a regular
 * column without a header scan (orcd_ctx_last_path) write as they verify, and what an attempt that
 * did not hold has written is overwritten only when the general path then succeeds. */
int orcd_rlev2_decode(orcd_ctx *ctx, const uint8_t *stream, size_t len, size_t num_values,
                      const uint64_t *entry_points, size_t n_entries, int64_t *out,
                      unsigned flags);

/* orcd_rlev2_decode for a stream whose entry points are the ROW_INDEX positions of a file, one per
 * row group (src/proto.rs:89-99, RowIndexEntry.positions: for an integer stream the byte offset of
 * a run header and the number of that run's values to skip; the crate drops the index,
 * src/read/mod.rs:136).  `values_per_group`: the stream holds that many values per row group --
 * the file's row index stride for a column without nulls -- and entry point k is the header of
 * the run that holds value k * values_per_group.  The writers do not cut a run where a row group
 * ends, so with the default stride of 10 000 a group starts 272, 32, 304 ... values into a run of
 * 512 and the segments between entry points hold 19 or 20 runs; the values-to-skip say exactly
 * that and are implied by values_per_group when every run holds 512 values.  The number only lets
 * the decoder try the paths that need no header scan (orcd_ctx_last_path) on such a column; every
 * header and entry point is still checked, and a stream that is not like that -- or a wrong
 * number -- is decoded the general way with the same result.  0: exactly orcd_rlev2_decode.
 * orcd_column_values_per_group gives the number for a stream of a column handle (0 when the
 * positions of the index are not of that shape). */
int orcd_rlev2_decode_groups(orcd_ctx *ctx, const uint8_t *stream, size_t len, size_t num_values,
                             const uint64_t *entry_points, size_t n_entries, uint64_t values_per_group,
                             int64_t *out, unsigned flags);

/* Compression chunks of a zlib file inflated on the device, a workgroup per chunk: what the crate's
 * Decompressor does one chunk at a time on the host (src/read/decompress/mod.rs:60-80: a chunk
 * whose header says "compressed" is a raw deflate stream, handed to flate2).  chunks[k] names the
 * deflate bytes of chunk k inside src (after its 3-byte header), the place in dst its output goes
 * to and the room there (the file's compression block size: a chunk that decompresses to more does
 * not inflate).  sizes[k] (host memory) receives the bytes chunk k decompressed to, or 0xFFFFFFFF;
 * the call returns ORCD_DECOMPRESSION when any chunk does not inflate -- the same chunks as zlib
 * refuses -- and the chunks that did are in dst all the same.  Flags: ORCD_IN_DEVICE (src), ORCD_OUT_DEVICE (dst); `chunks` and `sizes` are host
 * memory.  The places in dst must not overlap.  The kernel reads src in whole aligned 16-byte
 * words: with ORCD_IN_DEVICE up to 15 bytes in front of src and behind src + src_len are read
 * (never used, and never across a page that src itself does not touch). */
typedef struct {
    uint64_t src_offset;
    uint64_t dst_offset;
    uint32_t src_length;
    uint32_t dst_capacity;
} orcd_chunk;
int orcd_inflate_chunks(orcd_ctx *ctx, const uint8_t *src, size_t src_len, const orcd_chunk *chunks,
                        size_t n_chunks, uint8_t *dst, size_t dst_len, uint32_t *sizes, unsigned flags);

/* Integer RLE version 1: the DATA / LENGTH streams of files written as ORC 0.11 (ColumnEncoding
 * DIRECT / DICTIONARY).  The crate does not decode it (README.md:23-25 lists RLE v1 as missing; its
 * harness reads every integer stream as RLE v2, tests/it/deserialize.rs:48-82); the layout is the
 * ORC v1 specification's -- runs of 3..130 values with a delta of -128..127, groups of 1..128
 * literal varints.  Same contract as orcd_rlev2_decode: `num_values` values into out[], the same
 * flags, `entry_points` at starts of runs or literal groups (a wavefront walks a segment: a stream
 * without entry points is walked by a single wavefront). */
int orcd_rlev1_decode(orcd_ctx *ctx, const uint8_t *stream, size_t len, size_t num_values,
                      const uint64_t *entry_points, size_t n_entries, int64_t *out,
                      unsigned flags);

/* orcd_rlev2_decode without any allocation.  No counterpart in the crate: this is for callers that
 * cannot allocate device memory at the point of the call -- tight serving loops, harnesses that
 * forbid hipMalloc.  It allocates nothing, but it does wait for its stream: the host reads the
 * header totals (and the verdict of the single-launch attempt) back before it launches the rest,
 * so it cannot run under HIP graph capture.  `stream`, `entry_points` and `out` are DEVICE
 * pointers (the ORCD_IN_DEVICE / ORCD_OUT_DEVICE / ORCD_ENTRIES_HOST flags are not consulted; only
 * ORCD_SIGNED is), the results are those of orcd_rlev2_decode, and everything the kernels keep
 * beside input and output lives in a static device array of the library (under 8 MiB, held by every
 * GPU the library runs on until the process ends) instead of the context's growing buffers.  That
 * sets limits, beyond which the call returns ORCD_INVALID_ARGUMENT without decoding:
 *   num_values <= ORCD_NOALLOC_MAX_VALUES, n_entries <= ORCD_NOALLOC_MAX_ENTRIES, and the stream as a
 *   whole (trailing runs included) holds at most ORCD_NOALLOC_MAX_VALUES values in at most
 *   ORCD_NOALLOC_MAX_RUNS runs.
 * A long stream without entry points is walked serially (the entry-point guesser of
 * orcd_rlev2_decode allocates).  All calls on one GPU share the one array, whatever their context:
 * calls that overlap in time on one GPU race and must be serialized by the caller.
 * As for every decode call: after an error return the contents of out[] are undefined (an attempt
 * at the single-launch path that did not hold may have written part of it). */
enum {
    ORCD_NOALLOC_MAX_VALUES = 16777216, /* 2^24 */
    ORCD_NOALLOC_MAX_ENTRIES = 65536,
    ORCD_NOALLOC_MAX_RUNS = 524288
};
int orcd_rlev2_decode_noalloc(orcd_ctx *ctx, const uint8_t *stream, size_t len, size_t num_values,
                              const uint64_t *entry_points, size_t n_entries, int64_t *out,
                              unsigned flags);

/* Several RLE v2 streams that lie in one buffer -- the DATA streams of the integer columns of a
 * stripe, say -- decoded by one set of kernel launches instead of one set per stream (a call costs
 * about 50 microseconds whatever its size, which is what small columns are bounded by).  No
 * counterpart in the crate, whose iterators are per stream (src/read/decode/rle_v2.rs:480-758).
 *
 * Every slice is a whole stream: `length` bytes at `offset` of `buffer` that encode exactly
 * `num_values` values; all slices share the signedness given by ORCD_SIGNED.  `entry_points`
 * (always a HOST pointer, nullable) are offsets relative to the slice, as for orcd_rlev2_decode.
 * Slices come in ascending order of `offset` and do not overlap (gaps between them are fine);
 * anything else is ORCD_INVALID_ARGUMENT.  The values of slice i are written to out + (sum of num_values of the slices before i).
 * A slice that does not hold exactly its num_values values, or does not parse, yields
 * ORCD_OUT_OF_SPEC with *bad_slice (nullable) set to a slice at or before the first bad one;
 * decode the slices one by one to pin it down.  The last slice alone may hold more than its
 * num_values: what follows them is ignored, as at the end of a stream given to orcd_rlev2_decode. */
typedef struct {
    uint64_t offset, length, num_values;
    const uint64_t *entry_points;
    size_t n_entries;
} orcd_rlev2_slice;
int orcd_rlev2_decode_batch(orcd_ctx *ctx, const uint8_t *buffer, size_t buffer_len,
                            const orcd_rlev2_slice *slices, size_t n_slices, int64_t *out,
                            size_t *bad_slice, unsigned flags);

/* The same for boolean streams (PRESENT, boolean DATA): `num_values` of a slice counts values
 * (bits).  Every stream fills whole bytes, so the values of slice i are written from
 * out_bytes + 8 * (sum of ceil(num_values / 8) of the slices before i) on, one byte (0 / 1) each;
 * out_bytes needs 8 * (that sum over all slices) bytes, and the up to 7 bytes between the values
 * of two slices are padding.  set_bits (HOST pointer, nullable) receives the number of true
 * values of every slice. */
int orcd_boolean_decode_batch(orcd_ctx *ctx, const uint8_t *buffer, size_t buffer_len,
                              const orcd_rlev2_slice *slices, size_t n_slices, uint8_t *out_bytes,
                              uint64_t *set_bits, size_t *bad_slice, unsigned flags);

/* One decoded run header as SignedRleV2RunIter / UnsignedRleV2RunIter expose it
 * (src/read/decode/rle_v2.rs:432-439, 629-636: the Direct / Delta / ShortRepeat variants and
 * their len()). */
typedef struct {
    uint64_t payload_offset; /* offset in the stream of the run's packed bytes */
    uint64_t out_offset;     /* index of the run's first value in the decoded column */
    uint32_t kind;           /* 0 ShortRepeat, 1 Direct, 2 PatchedBase, 3 Delta */
    uint32_t bit_width;
    uint32_t length;
    uint32_t reserved;
} orcd_run;

/* Runs the header scan only and copies the run table to host memory: at most `max_runs` entries
 * into `runs`, the run count needed to cover `num_values` into `n_runs`. */
int orcd_rlev2_runs(orcd_ctx *ctx, const uint8_t *stream, size_t len, size_t num_values,
                    const uint64_t *entry_points, size_t n_entries, orcd_run *runs,
                    size_t max_runs, size_t *n_runs, unsigned flags);

/* ---- byte-RLE and booleans ------------------------------------------------------------------ */

/* One run of a byte-RLE stream: what BooleanRleRunIter yields as BooleanRun::Run(value, length) or
 * BooleanRun::Literals(bytes) (src/read/decode/boolean_rle.rs:7-12, 40-63), as positions instead of
 * copies: the repeated byte, or the first literal byte, is stream[payload_offset]. */
typedef struct {
    uint64_t payload_offset; /* offset in the stream of the value byte / of the first literal byte */
    uint64_t out_offset;     /* index of the run's first byte in the decoded bytes */
    uint32_t length;         /* decoded bytes: 3..130 for a run, 1..128 for literals */
    uint32_t is_run;         /* 1: `length` copies of one byte; 0: `length` literal bytes */
} orcd_byte_run;

#define ORCD_WHOLE_STREAM ((size_t)-1)

/* Replaces BooleanRleRunIter::new(reader) (src/read/decode/boolean_rle.rs:14-63): runs the
 * control-byte scan only and copies the run table to host memory -- at most `max_runs` entries
 * into `runs`, the number of runs needed to cover `num_bytes` decoded bytes into `n_runs`, the
 * decoded bytes those runs hold into `total_bytes` (nullable; the last run may reach past
 * num_bytes).  num_bytes = ORCD_WHOLE_STREAM: every run of the stream (the reference's iterator
 * takes no length); a last run that is cut short is ORCD_OUT_OF_SPEC then, as it is an Err there. */
int orcd_byte_rle_runs(orcd_ctx *ctx, const uint8_t *stream, size_t len, size_t num_bytes,
                       const uint64_t *entry_points, size_t n_entries, orcd_byte_run *runs,
                       size_t max_runs, size_t *n_runs, uint64_t *total_bytes, unsigned flags);

/* Replaces BooleanRleRunIter (src/read/decode/boolean_rle.rs:14-63) expanded to its bytes:
 * out[0..num_bytes).  `entry_points`: byte offsets of control bytes, as for orcd_rlev2_decode. */
int orcd_byte_rle_decode(orcd_ctx *ctx, const uint8_t *stream, size_t len, size_t num_bytes,
                         const uint64_t *entry_points, size_t n_entries, uint8_t *out,
                         unsigned flags);

/* Replaces BooleanIter::new(reader, length) (src/read/decode/boolean_rle.rs:65-158) for PRESENT
 * and boolean DATA streams.  Any of the three outputs may be NULL:
 *   out_bytes  [num_bits]           one byte (0/1) per value, the layout of Vec<bool>;
 *   out_bitmap [ceil(num_bits/8)]   least-significant-bit-first validity bitmap (Arrow layout);
 *   set_bits   (always a HOST pointer) number of true values -- `num_of_values` of
 *              tests/it/deserialize.rs:53. */
int orcd_boolean_decode(orcd_ctx *ctx, const uint8_t *stream, size_t len, size_t num_bits,
                        const uint64_t *entry_points, size_t n_entries, uint8_t *out_bytes,
                        uint8_t *out_bitmap, uint64_t *set_bits, unsigned flags);

/* ---- floats --------------------------------------------------------------------------------- */

/* Replaces FloatIter::<f32|f64, R>::new(reader, length) (src/read/decode/float.rs:34-91):
 * `num_values` little-endian IEEE values of `elem_size` (4 or 8) bytes, copied bit-for-bit.
 * A stream shorter than that yields ORCD_DECODE_FLOAT (float.rs:78-81). */
int orcd_float_decode(orcd_ctx *ctx, const uint8_t *stream, size_t len, size_t elem_size,
                      size_t num_values, void *out, unsigned flags);

/* ---- strings -------------------------------------------------------------------------------- */

/* Replaces Values::new(reader, scratch) + Values::next(length) driven by an unsigned RLE v2
 * LENGTH stream (src/read/decode/variable_length.rs:5-28; tests/it/deserialize.rs:106-140,
 * 187-201).  Produces Arrow-style offsets[0..num_values] (offsets[num_values] = total bytes) and,
 * when `bytes_out` is not NULL, the value bytes (offsets[num_values] of them) copied from
 * `data`.  A LENGTH stream asking for more than `data_len` bytes yields ORCD_OUT_OF_SPEC; with
 * ORCD_CHECK_UTF8 an ill-formed value yields ORCD_INVALID_UTF8. */
int orcd_string_direct_decode(orcd_ctx *ctx, const uint8_t *length_stream, size_t length_len,
                              const uint64_t *length_entry_points, size_t n_length_entries,
                              const uint8_t *data, size_t data_len, size_t num_values,
                              int64_t *offsets, uint8_t *bytes_out, unsigned flags);

/* Replaces the gather of deserialize_str_dict_array (tests/it/deserialize.rs:142-185):
 * value i = dictionary entry indices[i].  `dict_offsets` has dict_size + 1 entries.  Writes
 * offsets_out[0..num_values]; when `bytes_out` is not NULL also the gathered bytes, of which
 * there must be room for `bytes_capacity`.  An index >= dict_size yields ORCD_OUT_OF_SPEC
 * (deserialize.rs:162).  All pointers follow ORCD_IN_DEVICE / ORCD_OUT_DEVICE. */
int orcd_string_dictionary_decode(orcd_ctx *ctx, const int64_t *dict_offsets,
                                  const uint8_t *dict_bytes, size_t dict_size,
                                  const int64_t *indices, size_t num_values, int64_t *offsets_out,
                                  uint8_t *bytes_out, size_t bytes_capacity, unsigned flags);

/* ---- host side: file tail, stripe footers, column regions (no GPU involved) ------------------ */

typedef struct orcd_file orcd_file;
typedef struct orcd_stripe_footer orcd_stripe_footer;
typedef struct orcd_column orcd_column;

typedef struct { /* proto::StripeInformation, src/proto.rs:206-217 */
    uint64_t offset, index_length, data_length, footer_length, number_of_rows;
} orcd_stripe_info;

typedef struct { /* proto::Stream, src/proto.rs:113-137, plus its resolved position */
    uint32_t kind, column;
    uint64_t length;
    uint64_t offset; /* offset from the start of the stripe */
} orcd_stream_info;

/* Replaces read::read_metadata(&mut reader) -> FileMetadata (src/read/mod.rs:46-89).  Keeps the
 * file open for the later region reads.  Unlike the crate, a tail larger than 16 KiB is re-read
 * at its exact size instead of panicking.  Files of CompressionKind None and Zlib are read
 * (src/read/decompress/mod.rs:60-80: footers, stripe footers and streams as chains of compression
 * chunks, decompressed on the host; broken chunks are ORCD_DECOMPRESSION), and, where the crate
 * panics (:81), Snappy, LZ4 and Zstandard -- the latter two when the process can load the system's
 * liblz4.so.1 / libzstd.so.1; LZO, an unknown kind or a missing library is ORCD_UNSUPPORTED. */
int orcd_file_open(const char *path, orcd_file **file);
/* Same, over a file image in host memory (borrowed: it must outlive the handle). */
int orcd_file_open_memory(const uint8_t *bytes, size_t len, orcd_file **file);
void orcd_file_close(orcd_file *file);
/* FileMetadata accessors: footer.stripes.len(), footer.stripes[i], postscript.compression(),
 * footer.number_of_rows, footer.types.len(), footer.types[i].kind / .subtypes. */
size_t orcd_file_num_stripes(const orcd_file *file);
int orcd_file_stripe(const orcd_file *file, size_t stripe, orcd_stripe_info *info);
int orcd_file_compression(const orcd_file *file);
uint64_t orcd_file_number_of_rows(const orcd_file *file);
uint32_t orcd_file_row_index_stride(const orcd_file *file);
size_t orcd_file_num_types(const orcd_file *file);
int orcd_file_type_kind(const orcd_file *file, size_t type_id);

/* Replaces read::read_stripe_footer(reader, &metadata, stripe, scratch) -> StripeFooter
 * (src/read/mod.rs:95-111): one seek and one read. */
int orcd_read_stripe_footer(orcd_file *file, size_t stripe, orcd_stripe_footer **footer);
void orcd_stripe_footer_free(orcd_stripe_footer *footer);
size_t orcd_stripe_footer_num_streams(const orcd_stripe_footer *footer);
int orcd_stripe_footer_stream(const orcd_stripe_footer *footer, size_t i, orcd_stream_info *info);

/* Replaces read::read_stripe_column(reader, &metadata, stripe, footer, column, scratch) -> Column
 * (src/read/mod.rs:117-159): one seek and one read of the column's non-ROW_INDEX streams.  The
 * footer is borrowed and must outlive the column.  An absent column yields ORCD_INVALID_COLUMN. */
int orcd_read_stripe_column(orcd_file *file, size_t stripe, const orcd_stripe_footer *footer,
                            uint32_t column, orcd_column **out);
void orcd_column_free(orcd_column *column);
/* Column::get_stream(kind, scratch) (src/read/column.rs:38-55): a borrowed view of the stream's
 * bytes; ORCD_INVALID_KIND when the column has no such stream. */
int orcd_column_get_stream(const orcd_column *column, int kind, const uint8_t **bytes,
                           size_t *len);
/* Column::encoding() / dictionary_size() / number_of_rows() (src/read/column.rs:58-72). */
int orcd_column_encoding(const orcd_column *column, int *kind, uint32_t *dictionary_size);
uint64_t orcd_column_number_of_rows(const orcd_column *column);
/* For readers that fetch a whole stripe at once (one read, one copy to the device) and decode its
 * streams where they lie: a column handle without the column's bytes (orcd_column_get_stream then
 * fails; encoding, row count and entry points work), the region orcd_read_stripe_column would read
 * and the place of one of its streams inside it -- both as offsets from the start of the stripe,
 * and found the way the crate finds them (the region starts at the column's first stream and its
 * streams follow one another, src/read/mod.rs:129-144, column.rs:38-55) -- and a plain positioned
 * read of file bytes. */
int orcd_stripe_column_open(orcd_file *file, size_t stripe, const orcd_stripe_footer *footer,
                            uint32_t column, orcd_column **out);
int orcd_column_region(const orcd_column *column, uint64_t *offset, uint64_t *len);
int orcd_column_stream_range(const orcd_column *column, int kind, uint64_t *offset, uint64_t *len);
int orcd_file_read_at(orcd_file *file, uint64_t offset, uint64_t len, uint8_t *dst);
/* Run-header entry points for orcd_rlev2_decode / orcd_byte_rle_decode taken from the column's
 * ROW_INDEX stream (src/proto.rs:89-99), which the crate skips (src/read/mod.rs:136).  Reads the
 * index stream on first use.  Returns a borrowed array (possibly empty) of ascending stream
 * offsets for stream `kind`. */
int orcd_column_entry_points(orcd_file *file, orcd_column *column, int kind,
                             const uint64_t **entries, size_t *n_entries);
/* For the same stream: the `values_per_group` of orcd_rlev2_decode_groups -- the file's row index
 * stride when the index holds one position per row group for the stream and every position's
 * values-to-skip (the second number of an RLE v2 position, src/proto.rs:89-99) is what runs of 512
 * values throughout make it, (k * stride) mod 512 -- else 0. */
uint64_t orcd_column_values_per_group(orcd_file *file, orcd_column *column, int kind);

/* ---- null scatter and whole columns ------------------------------------------------------------ */

/* Dense non-null values -> one slot per row.  The crate's harness stops at (validity, dense
 * values) (tests/it/deserialize.rs:48-82: the values are not moved to their rows); this is the
 * step a columnar consumer adds on top.  `validity` has num_rows bytes (non-zero = valid, the
 * out_bytes of orcd_boolean_decode), `dense` has num_values entries of elem_size (1, 4 or 8)
 * bytes, `out` receives num_rows entries: out[r] = dense[number of valid rows before r] for valid
 * rows and zero for null rows.  With ORCD_SCATTER_CARRY null rows receive that dense entry too and
 * `dense` has num_values + 1 entries: string offsets scattered this way give every null row an
 * empty value.  The number of valid rows must equal num_values (else ORCD_INVALID_ARGUMENT).
 * All three pointers follow ORCD_IN_DEVICE (validity, dense) / ORCD_OUT_DEVICE (out). */
int orcd_scatter_valid(orcd_ctx *ctx, const uint8_t *validity, size_t num_rows, const void *dense,
                       size_t num_values, size_t elem_size, void *out, unsigned flags);

/* One column of one stripe, decoded: what the deserialize_* functions of the crate's harness return
 * (tests/it/deserialize.rs:13-222), for the column's ORC type.  Arrays are owned by the structure
 * (release with orcd_column_data_free) and live in host memory -- pinned, so that the copy from
 * the device runs at the speed of the link -- or in device memory when the call had
 * ORCD_OUT_DEVICE.  Either way they come from a cache of the context: released arrays go back to
 * it and serve its next calls (orcd_ctx_release_cached empties it).  A device array is handed on
 * without waiting for the device: work on other streams that still reads it must be finished (or
 * ordered before the context's stream) when it is released.  All arrays are released before the
 * context is destroyed. */
typedef struct {
    int type_kind;        /* proto Type.Kind of the column (src/proto.rs:181-203) */
    int on_device;        /* where the arrays below live */
    int row_aligned;      /* ORCD_ROW_ALIGNED was applied */
    uint64_t num_rows;    /* rows of the stripe = entries of `validity` */
    uint64_t num_values;  /* non-null rows */
    uint64_t num_slots;   /* entries of `values` / of `offsets` minus one: num_values, or num_rows when row-aligned */
    uint8_t *validity;    /* [num_rows] 0 / 1 (Vec<bool>); all 1 when the column has no PRESENT stream */
    void *values;         /* SHORT / INT / LONG / DATE: int64; FLOAT: f32; DOUBLE: f64; BOOLEAN, BYTE: one byte; strings: NULL */
    size_t value_size;    /* bytes per entry of `values` (0 for strings) */
    int64_t *offsets;     /* STRING / VARCHAR / CHAR / BINARY: [num_slots + 1] offsets into `bytes` */
    uint8_t *bytes;       /* the values' bytes, concatenated */
    uint64_t num_bytes;
    orcd_ctx *owner;      /* the context whose cache the arrays came from and go back to */
} orcd_column_data;

/* Reads column `column` of stripe `stripe` (the region read_stripe_column reads), finds its
 * streams and ROW_INDEX entry points and runs the decoders above.  Flags: ORCD_OUT_DEVICE,
 * ORCD_CHECK_UTF8 (STRING / VARCHAR / CHAR), ORCD_ROW_ALIGNED.  Departures from the harness, both
 * where the harness departs from the format: a column without PRESENT stream is all valid
 * (deserialize.rs:14 returns InvalidKind), a dictionary column holds one index per non-null row
 * (deserialize.rs:160 uses the row count).  Unsupported types yield ORCD_UNSUPPORTED. */
int orcd_decode_column(orcd_ctx *ctx, orcd_file *file, size_t stripe,
                       const orcd_stripe_footer *footer, uint32_t column, unsigned flags,
                       orcd_column_data *out);
void orcd_column_data_free(orcd_column_data *data);

/* The same for several columns of one stripe: out[i] receives column columns[i].  The DATA streams
 * of the integer columns are decoded together (orcd_rlev2_decode_batch), the other streams column
 * by column.  On failure nothing is left in `out`. */
int orcd_decode_stripe(orcd_ctx *ctx, orcd_file *file, size_t stripe,
                       const orcd_stripe_footer *footer, const uint32_t *columns, size_t n_columns,
                       unsigned flags, orcd_column_data *out);

/* A hint that orcd_decode_stripe will be called for these columns of this stripe: a background
 * thread of the context starts the one read of the region the columns span (what
 * read_stripe_column does with one seek and one read per column, src/read/mod.rs:117-159) and its
 * copy to the device, with pinned chunks, reader threads and a stream of its own, so that they run
 * beside the decode of the stripe before.  The decode call for the same file, stripe and columns
 * then waits for those bytes instead of reading them; with other arguments it reads for itself and
 * the bytes read ahead stay until they are claimed, pushed out or the context is destroyed.  At
 * most two stripes are kept ahead, served in the order given: a further hint pushes out the oldest
 * one if that has been read and never asked for, and is ignored otherwise.  Of compressed files,
 * zlib ones are read ahead (the bytes wait in host memory too, where the decode call walks the chunk
 * headers) unless ORCD_GPU_INFLATE=0; the others are not.  `file` and `footer` must stay open until the stripe has been decoded or the
 * context destroyed.  Returns what finding the region returns (ORCD_INVALID_COLUMN, ORCD_OUT_OF_SPEC
 * for a region outside the file, ...); a failed read shows when the stripe is decoded. */
int orcd_stripe_prefetch(orcd_ctx *ctx, orcd_file *file, size_t stripe,
                         const orcd_stripe_footer *footer, const uint32_t *columns, size_t n_columns);

#ifdef __cplusplus
}
#endif
#endif

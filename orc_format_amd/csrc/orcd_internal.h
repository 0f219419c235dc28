// Shared declarations of the decoder's translation units (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orcd.h"

namespace orcd {

// ---- RLE v2 run table -------------------------------------------------------------------------
// Written by the header-scan kernel, read by the value kernel.
//
// A run is one 64-bit descriptor:
//   [0,39)  payload_off   offset in the stream of the run's packed bytes
//   [39,41) kind          0 ShortRepeat, 1 Direct, 2 PatchedBase, 3 Delta
//   [41,48) width         packed bit width, 0..64
//   [48,57) len - 1       run length 1..512
//   57      delta_neg     Delta: the per-value deltas are subtracted
//   58      single        (per-window copies only) no other run starts in the window
//   [59,64) header bytes  distance from the run's first header byte to payload_off (1..22)
// The run table proper (`runs`) holds descriptors only: the value kernel gets a run's place in the
// output from a prefix sum of the lengths and its base / second field by parsing the header again
// (one lane per run, once).  The per-window copy of a window's first run (RunEntry) carries them
// ready-made, so that a window that is one run -- the bulk of a long column -- needs nothing else.
struct RunEntry {
    uint64_t desc;
    uint64_t out_off;  // index of the run's first value in the output
    uint64_t base;     // ShortRepeat: the value (zigzag applied for signed streams);
                       // Delta: first value; PatchedBase: base
    uint64_t aux;      // Delta: signed delta_base; PatchedBase: patch geometry (see PB_* below)
};
static_assert(sizeof(RunEntry) == 32, "RunEntry is read as two 16-byte loads");

constexpr int KIND_SHORT_REPEAT = 0, KIND_DIRECT = 1, KIND_PATCHED_BASE = 2, KIND_DELTA = 3;
constexpr uint64_t RUN_PAYLOAD_MASK = (1ull << 39) - 1;
constexpr uint64_t DESC_SINGLE = 1ull << 58;
constexpr uint64_t PAYLOAD_MASK = (1ull << 40) - 1;  // byte-RLE run table

__host__ __device__ inline uint64_t make_desc(uint64_t payload_off, uint32_t kind, uint32_t width,
                                              uint32_t len, uint32_t delta_neg, uint32_t header_bytes) {
    return (payload_off & RUN_PAYLOAD_MASK) | ((uint64_t)kind << 39) | ((uint64_t)width << 41) |
           ((uint64_t)(len - 1) << 48) | ((uint64_t)delta_neg << 57) | ((uint64_t)header_bytes << 59);
}
__host__ __device__ inline uint64_t desc_payload(uint64_t d) { return d & RUN_PAYLOAD_MASK; }
__host__ __device__ inline uint32_t desc_kind(uint64_t d) { return (uint32_t)(d >> 39) & 3u; }
__host__ __device__ inline uint32_t desc_width(uint64_t d) { return (uint32_t)(d >> 41) & 127u; }
__host__ __device__ inline uint32_t desc_len(uint64_t d) { return ((uint32_t)(d >> 48) & 511u) + 1u; }
__host__ __device__ inline uint32_t desc_delta_neg(uint64_t d) { return (uint32_t)(d >> 57) & 1u; }
__host__ __device__ inline uint32_t desc_header_bytes(uint64_t d) { return (uint32_t)(d >> 59) & 31u; }

// PatchedBase geometry packed in RunEntry::aux:
//   [0,7) patch width  [7,11) gap width  [11,16) patch list length  [16,23) packed entry width
__host__ __device__ inline uint64_t make_pb_aux(uint32_t pw, uint32_t gw, uint32_t pll, uint32_t cfb) {
    return (uint64_t)pw | ((uint64_t)gw << 7) | ((uint64_t)pll << 11) | ((uint64_t)cfb << 16);
}
__host__ __device__ inline uint32_t pb_patch_width(uint64_t a) { return (uint32_t)a & 127u; }
__host__ __device__ inline uint32_t pb_gap_width(uint64_t a) { return (uint32_t)(a >> 7) & 15u; }
__host__ __device__ inline uint32_t pb_list_len(uint64_t a) { return (uint32_t)(a >> 11) & 31u; }
__host__ __device__ inline uint32_t pb_entry_bits(uint64_t a) { return (uint32_t)(a >> 16) & 127u; }

// Per-segment result of the counting pass (a segment = the headers between two entry points).
struct SegCount {
    uint32_t n_runs;
    uint32_t last_len;  // length of the segment's last run
    uint32_t status;    // SEG_*
    uint32_t pad;
    uint64_t n_vals;
};
// Per-segment offsets produced by the segment scan.
struct SegOffset {
    uint64_t val_off;   // values before this segment
    uint32_t run_off;   // runs before this segment
    uint32_t prev_len;  // length of the last run before this segment (0: none)
};
// Totals of one stream, read back by the host after the segment scan.
struct ScanTotals {
    uint64_t total_vals;
    uint64_t total_runs;
    uint32_t error;      // ORCD_* status; 0 when the first `num_values` values are decodable
    uint32_t error_seg;
    uint32_t max_seg_runs;  // most runs any one segment holds
    uint32_t pad;
};

constexpr uint32_t SEG_OK = 0;
constexpr uint32_t SEG_TRUNCATED = 1;    // a run extends past the end of the stream
constexpr uint32_t SEG_MISALIGNED = 2;   // the walk stepped over the next entry point
constexpr uint32_t SEG_BAD_RUN = 3;      // a header that the specification forbids
constexpr uint32_t SEG_TOO_LONG = 4;     // stream offset does not fit the run table

// Values per output window: one wavefront decodes an aligned window of 512 values (8 per lane).
constexpr uint32_t WINDOW_SHIFT = 9;
constexpr uint32_t WINDOW = 1u << WINDOW_SHIFT;

// ---- byte-RLE run table -------------------------------------------------------------------------
// src: [0,40) offset of the value byte (run) or of the first literal byte, [40,48) len-1, bit 48 is_run
struct ByteRun {
    uint64_t src;
    uint64_t out_off;
};

// (orcd_host.cpp, not part of include/orcd.h) the bytes a column handle holds: its region of the
// file, or for a compressed file its streams decompressed and 16-byte aligned, back to back --
// what orcd_column_get_stream points into.
#include "orcd_column_internal.h"

// ---- launchers (defined in the .hip files, called from the C ABI layer) ---------------------
struct SegTable;  // seg_scan.h

// entries: start offset of every segment; seg_ends (nullable): its end, when the segments are
// slices of a larger buffer rather than one gapless stream.
// The counting pass keeps the descriptors of a segment's first `staged_cap` runs in
// staged[seg * staged_cap + k] (nullable): segments that hold no more than that are then emitted
// by launch_rlev2_emit_staged, one lane per run, without walking the header chain a second time;
// launch_rlev2_emit walks the segments with more than `staged_up_to` runs (0: all of them).
// staged_cap is chosen per stream from the bytes an average segment holds (staged_runs_cap): ROW_INDEX
// positions come every 10 000 values, which is 20 runs of 512 values or 500 runs of 20.
constexpr uint32_t STAGED_RUNS_MIN = 64, STAGED_RUNS_MAX = 512;
inline uint32_t staged_runs_cap(uint64_t stream_len, uint32_t n_seg) {
    const uint64_t avg = stream_len / (n_seg ? n_seg : 1);
    uint32_t cap = STAGED_RUNS_MIN;
    while (cap < STAGED_RUNS_MAX && (uint64_t)cap * 16 < avg) cap <<= 1;  // a run of 16 stream bytes on average
    return cap;
}
hipError_t launch_rlev2_count(hipStream_t st, const uint8_t *stream, uint64_t len,
                              const uint64_t *entries, const uint64_t *seg_ends, uint32_t n_seg,
                              int is_signed, const SegTable &t, uint64_t *staged, uint32_t staged_cap);
// The counting pass for streams of at most COOP_MAX_SEGMENTS segments: a wavefront per segment
// (see rlev2_coop_count_kernel); same results in `t`, same staging.
constexpr uint32_t COOP_MAX_SEGMENTS = 16384;
// A stream longer than this does not stay in the 256 MB last-level cache between its loads:
// rlev2_selfplan_values_kernel loads it with the nontemporal hint.
constexpr uint64_t SELFPLAN_STREAMING_BYTES = 512ull << 20;
hipError_t launch_rlev2_coop_count(hipStream_t st, const uint8_t *stream, uint64_t len,
                                   const uint64_t *entries, const uint64_t *seg_ends, uint32_t n_seg,
                                   int is_signed, const SegTable &t, uint64_t *staged, uint32_t staged_cap);
hipError_t launch_rlev2_emit(hipStream_t st, const uint8_t *stream, uint64_t len,
                             const uint64_t *entries, const uint64_t *seg_ends, uint32_t n_seg,
                             int is_signed, const SegTable &t, uint64_t *runs, uint32_t *group_first,
                             RunEntry *group_run, uint32_t n_group_slots, bool full_table,
                             uint32_t staged_up_to);
hipError_t launch_rlev2_emit_staged(hipStream_t st, const uint8_t *stream, uint64_t len,
                                    const uint64_t *entries, const uint64_t *seg_ends, uint32_t n_seg,
                                    int is_signed, const SegTable &t, const uint64_t *staged,
                                    uint32_t staged_cap, uint64_t *runs, uint32_t *group_first,
                                    RunEntry *group_run, uint32_t n_group_slots, bool full_table,
                                    uint32_t max_seg_runs);
// lanes the staged emit kernels give a segment: the power of two from 8 to 64 that holds the
// most runs any segment has (a segment with more takes them 64 at a time)
inline uint32_t staged_group_lanes(uint32_t max_seg_runs) {
    uint32_t g = 8;
    while (g < 64 && g < max_seg_runs) g <<= 1;
    return g;
}
// Parallel check of a guessed header chain (see rlev2.hip): the size of the run at `pos`, and the
// first k < n for which p + k * stride is not the header of a run of `stride` bytes (*first_fail
// must hold n or more beforehand).
hipError_t launch_rlev2_probe(hipStream_t st, const uint8_t *stream, uint64_t len, uint64_t pos,
                              int is_signed, uint64_t *out);
hipError_t launch_rlev2_verify_stride(hipStream_t st, const uint8_t *stream, uint64_t len, uint64_t p,
                                      uint64_t stride, uint64_t n, int is_signed, uint64_t *first_fail);
// first k < n whose segment first_seg[k] does not have want_off[k] values before it -> host_word[0]
// (0xFFFFFFFF: none), then `stamp` -> host_word[1]; host_word: host memory the device can write
hipError_t launch_seg_offsets_verify(hipStream_t st, const SegTable &t, const uint32_t *first_seg,
                                     const uint64_t *want_off, uint32_t n, uint64_t *host_word, uint32_t stamp);
hipError_t launch_seg_value_offsets(hipStream_t st, const SegTable &t, const uint32_t *first_seg,
                                    uint32_t n, uint64_t *out);
// What the header scan leaves for the value kernels of one stream.
struct Rlev2Table {
    const uint8_t *stream;
    uint64_t len;
    const uint64_t *runs;          // descriptors [n_runs + 1], the last one a sentinel
    const uint32_t *group_first;   // per window of 512 values: index of the first run starting in it
    const RunEntry *group_run;     // ... and a copy of that run, flagged when it is the only one
    uint32_t n_groups;             // windows that hold wanted values
    uint32_t n_runs;
    uint64_t num_values;
    int is_signed;
};
// Integer RLE version 1 (rlev2.hip, last section): values per segment into `t` (then the segment
// scan), and -- the same walk again -- the values themselves at the segments' offsets.
// ---- inflate.hip: the compression chunks of a zlib file, a wavefront each ----
struct InflateChunk {
    uint64_t src_off;   // first byte of the raw deflate stream
    uint64_t dst_off;   // where its output goes
    uint32_t src_len;
    uint32_t dst_cap;   // most it may decompress to
};
struct InflateMove {
    uint64_t from, to, len;
};
// sizes[k]: bytes chunk k decompressed to, 0xFFFFFFFF when it does not inflate
hipError_t launch_inflate_chunks(hipStream_t st, const uint8_t *src, uint64_t src_total, const InflateChunk *chunks,
                                 uint32_t n_chunks, uint8_t *dst, uint32_t *sizes);
hipError_t launch_inflate_compact(hipStream_t st, const uint8_t *from, uint8_t *to, const InflateMove *moves,
                                  uint32_t n_moves);

hipError_t launch_rlev1_count(hipStream_t st, const uint8_t *stream, uint64_t len, const uint64_t *entries,
                              uint32_t n_seg, int is_signed, const SegTable &t);
hipError_t launch_rlev1_emit(hipStream_t st, const uint8_t *stream, uint64_t len, const uint64_t *entries,
                             uint32_t n_seg, int is_signed, const SegTable &t, uint64_t num_values, int64_t *out);
// Speculative entry points every `spacing` bytes for an irregular stream without any (see
// rlev2_resync_kernel): entries[0 .. n_entries), to be verified by the counting pass.
hipError_t launch_rlev2_resync(hipStream_t st, const uint8_t *stream, uint64_t len, int is_signed,
                               uint64_t spacing, uint32_t n_entries, uint64_t *entries);
// abort_flag (nullable): a device word; when it is not zero as the kernel starts, nothing is decoded
hipError_t launch_rlev2_values(hipStream_t st, const Rlev2Table &t, int64_t *out,
                               const uint32_t *abort_flag = nullptr);
// (which of its two variants that launches: the one for columns of more than two runs per window)
bool rlev2_values_short_variant(const Rlev2Table &t);
// The same behind a plan of the column: every window of t.group_run is one run (t.runs and
// t.group_first are not looked at).
hipError_t launch_rlev2_planned_values(hipStream_t st, const Rlev2Table &t, int64_t *out,
                                       const uint32_t *abort_flag);
// The run of every window for the column whose windows of 512 values are one run of 512 values
// each (any kind), equally many of them between two entry points (see rlev2_regular_plan_kernel):
// group_run[0 .. n_win), or *flag raised when the column is not like that.
// rlev2_regular_plan_shape: whether the counts of windows and entry points allow such a column at
// all, and the runs per segment they leave possible (the kernel picks among several).
bool rlev2_regular_plan_shape(uint32_t n_seg, uint32_t n_win, uint32_t *r_lo, uint32_t *r_hi);
// Plan and expansion in one launch (see rlev2_selfplan_values_kernel).  stride_vals: values between
// two entry points (entry k is the header of the run that holds value k * stride_vals; any number
// from 512 to 2^23), or 0: equally many runs in every segment, how many follows from the counts.
// flag: three zeroed words; flag[0] is raised when the column is not like that.  seg_table:
// rlev2_selfplan_table_bytes(n_seg) bytes of device memory.  rlev2_selfplan_shape says whether the
// counts are for this path at all.
hipError_t launch_rlev2_selfplan_values(hipStream_t st, const uint8_t *stream, uint64_t len,
                                        const uint64_t *entries, uint32_t n_seg, uint32_t n_win,
                                        uint32_t stride_vals, uint64_t num_values, int is_signed, int64_t *out,
                                        uint32_t *flag, void *seg_table);
size_t rlev2_selfplan_table_bytes(uint32_t n_seg);
bool rlev2_selfplan_shape(uint32_t n_seg, uint32_t n_win, uint32_t stride_vals);

hipError_t launch_rlev2_regular_plan(hipStream_t st, const uint8_t *stream, uint64_t len,
                                     const uint64_t *entries, uint32_t n_seg, uint32_t n_win,
                                     uint64_t num_values, int is_signed, RunEntry *group_run,
                                     uint32_t *flag);
// *flag into host_totals->error, flag zeroed, host_totals->pad stamped (as launch_rlev2_strided_values ends)
hipError_t launch_spec_publish(hipStream_t st, uint32_t *flag, ScanTotals *host_totals, uint32_t stamp);
// Value expansion without a header scan, for the column whose every window of 512 values is one
// Direct run of 512 values with equally many equal-sized runs between two entry points (see
// rlev2_strided_values_kernel).  *flag, zero beforehand, is raised when the stream is not like
// that; when the work is done it is copied to host_totals->error and zeroed again, and
// host_totals->pad is stamped.  `out` must be 16-byte aligned.
hipError_t launch_rlev2_strided_values(hipStream_t st, const uint8_t *stream, uint64_t len,
                                       const uint64_t *entries, uint32_t n_seg, uint32_t n_win,
                                       uint64_t num_values, int is_signed, int64_t *out, uint32_t *flag,
                                       ScanTotals *host_totals, uint32_t stamp, uint32_t stride_vals = 0);
// The same expansion, reduced: group_sums[g] = saturating sum of the values of the runs that
// start in the g-th window of 512 values (bit 2 of *flags is raised when a PatchedBase run was
// met: its patches are not in the sums) ...
hipError_t launch_rlev2_group_sums(hipStream_t st, const Rlev2Table &t, uint64_t *group_sums,
                                   uint32_t *flags);
// ... and, given the exclusive prefix sums of those, offsets[i] = sum of the values before i.
hipError_t launch_rlev2_offsets(hipStream_t st, const Rlev2Table &t, const uint64_t *group_prefix,
                                int64_t *offsets);

// Entry points of a byte-RLE stream that came without any: entries[k], k < n_entries =
// ceil(len / BC_SEG), is the first control byte at or after k * BC_SEG (len where there is none),
// computed exactly (see byte_rle_exits_kernel).  `work`: byte_rle_compose_bytes(len) bytes of
// device memory.  len must be more than BC_SEG.
constexpr uint32_t BC_SEG = 2048, BC_PITCH = 132, BC_FAN = 64;
size_t byte_rle_compose_bytes(uint64_t len, uint32_t *n_entries);
hipError_t launch_byte_rle_compose_entries(hipStream_t st, const uint8_t *stream, uint64_t len, uint8_t *work,
                                           uint64_t *entries);
// (staged, staged_up_to: as for the RLE v2 passes; a staged word is the run's ByteRun::src)
hipError_t launch_byte_rle_count(hipStream_t st, const uint8_t *stream, uint64_t len,
                                 const uint64_t *entries, const uint64_t *seg_ends, uint32_t n_seg,
                                 const SegTable &t, uint64_t *staged, uint32_t staged_cap);
hipError_t launch_byte_rle_emit(hipStream_t st, const uint8_t *stream, uint64_t len,
                                const uint64_t *entries, const uint64_t *seg_ends, uint32_t n_seg,
                                const SegTable &t, ByteRun *runs, uint32_t *win_first,
                                uint32_t n_win_slots, uint32_t staged_up_to);
hipError_t launch_byte_rle_emit_staged(hipStream_t st, uint32_t n_seg, const SegTable &t,
                                       const uint64_t *staged, uint32_t staged_cap, ByteRun *runs,
                                       uint32_t *win_first, uint32_t n_win_slots, uint32_t max_seg_runs);
hipError_t launch_count_nonzero_slices(hipStream_t st, const uint8_t *bytes, const uint64_t *start,
                                       const uint64_t *count, uint32_t n_slices, uint64_t *out,
                                       uint64_t largest);
// slots of win_first for a stream of total_bytes decoded bytes
size_t byte_rle_window_slots(uint64_t total_bytes);
hipError_t launch_byte_rle_expand(hipStream_t st, const uint8_t *stream, uint64_t len,
                                  const ByteRun *runs, const uint32_t *win_first, uint32_t n_win_slots,
                                  uint32_t n_runs, uint64_t num_bytes, uint8_t *out);
// run table -> booleans in one pass: one byte per value, and/or an LSB-first bitmap, plus the
// number of ones: set_bits (nullable) is BOOL_COUNT_BYTES of device memory whose first word holds
// the count when the work queued here is done
constexpr size_t BOOL_COUNT_BYTES = 8192;
hipError_t launch_bool_expand(hipStream_t st, const uint8_t *stream, uint64_t len,
                              const ByteRun *runs, const uint32_t *win_first, uint32_t n_win_slots,
                              uint32_t n_runs, uint64_t num_bits, uint8_t *out_bytes,
                              uint8_t *out_bitmap, uint64_t *set_bits);

// In-place exclusive prefix sum of n 64-bit values; data[n] receives the total.
// `block_sums` needs room for scan_block_sums_len(n) values.  A value above `limit` (which must
// be below 2^52) sets bit 0 of *over_flag; with every value within the limit the sums of a tile
// cannot wrap, and the sums across tiles saturate at UINT64_MAX instead of wrapping, so a total
// that is <= limit proves that every offset is exact and in [0, limit].
size_t scan_block_sums_len(uint64_t n);
hipError_t launch_exclusive_scan_u64(hipStream_t st, uint64_t *data, uint64_t n,
                                     uint64_t *block_sums, uint64_t limit, uint32_t *over_flag);

// Copies total_bytes of string data to `dst` (when not null) and / or checks (when invalid_flag
// is not null) that every value offsets[i]..offsets[i+1] is well-formed UTF-8; both in one pass
// over the bytes.  offsets: n + 1 ascending values ending at total_bytes.
hipError_t launch_string_bytes(hipStream_t st, const uint8_t *src, uint8_t *dst,
                               const int64_t *offsets, uint64_t n, uint64_t total_bytes,
                               uint32_t *invalid_flag);
// Dictionary strings: bytes per chunk of 64 values (indices and dictionary offsets checked: bit 0
// of *error_flag an index out of range, bit 1 offsets that do not describe the dictionary), and,
// given the exclusive prefix sums of those (n_chunks + 1 entries, the last one the total), the
// values' offsets (n + 1) and -- bytes_out not null -- their bytes.
hipError_t launch_dictionary_max_len(hipStream_t st, const int64_t *dict_offsets, uint64_t dict_size, uint32_t *max_len);
hipError_t launch_dictionary_chunk_sums(hipStream_t st, const int64_t *dict_offsets, uint64_t dict_size,
                                        uint64_t dict_total, const int64_t *indices, uint64_t n,
                                        uint64_t *chunk_sums, uint32_t *error_flag);
hipError_t launch_dictionary_gather(hipStream_t st, const int64_t *dict_offsets,
                                    const uint8_t *dict_bytes, uint64_t dict_size, uint64_t dict_total,
                                    const int64_t *indices, uint64_t n, const uint64_t *chunk_prefix,
                                    int64_t *offsets_out, uint8_t *bytes_out, uint32_t max_len,
                                    uint64_t capacity = ~0ull);

// Null scatter: per-tile counts of valid rows, then (after an exclusive scan of the counts)
// out[r] = dense[number of valid rows before r] for valid rows, zero for null rows -- or the same
// dense entry for null rows too when `carry` (string offsets).
size_t scatter_tiles(uint64_t num_rows);
constexpr uint64_t SC_MAX_TILE_COUNT = 1024;  // rows per tile: no tile count exceeds it
hipError_t launch_valid_count(hipStream_t st, const uint8_t *validity, uint64_t num_rows,
                              uint64_t *tile_counts);
hipError_t launch_scatter_valid(hipStream_t st, const uint8_t *validity, uint64_t num_rows,
                                const uint64_t *tile_prefix, const void *dense, uint64_t dense_len,
                                size_t elem_size, void *out, bool carry);

}  // namespace orcd

// RLE v2 integer decode for gfx950: header scan, segment prefix sums, value expansion.
//
// The math is that of src/read/decode/rle_v2.rs of the orc-format crate (ShortRepeat :283-301,
// Direct :135-175, Delta :195-264/343-408, zigzag :92-98, unpack :105-123, varint :71-89) with the
// full bit-width table and PatchedBase of the ORC v1 specification.  The structure is not the
// crate's: run headers form a serial chain, so they are walked once per *segment* (the bytes
// between two caller-supplied entry points) by one lane each, and the values -- the bulk of the
// bytes -- are then expanded by one wavefront per aligned window of 512 output values.
#include <type_traits>

#include "orcd_internal.h"
#include "seg_scan.h"

namespace orcd {

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t width_from_code(uint32_t code) {
    // ORC v1 "5 bit width encoding table": 0..23 -> 1..24, then 26 28 30 32 40 48 56 64
    if (code <= 23) return code + 1;
    const uint32_t t = code - 24;
    return t < 4 ? 26 + 2 * t : 40 + 8 * (t - 4);
}

__device__ __forceinline__ uint32_t closest_fixed_bits(uint32_t n) {
    if (n == 0) return 1;
    if (n <= 24) return n;
    if (n <= 32) return (n + 1) & ~1u;
    return (n + 7) & ~7u;
}

__device__ __forceinline__ uint64_t zigzag(uint64_t z) { return (z >> 1) ^ (0 - (z & 1)); }

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src_lane) {
    uint32_t lo = __shfl((uint32_t)v, src_lane), hi = __shfl((uint32_t)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_up64(uint64_t v, unsigned d) {
    uint32_t lo = __shfl_up((uint32_t)v, d), hi = __shfl_up((uint32_t)(v >> 32), d);
    return ((uint64_t)hi << 32) | lo;
}
// value of `v` in lane `lane` (wave-uniform lane index), as a scalar
__device__ __forceinline__ uint32_t readlane32(uint32_t v, int lane) {
    return __builtin_amdgcn_readlane(v, lane);
}
__device__ __forceinline__ uint64_t readfirstlane64(uint64_t v) {
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane) {
    uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, lane);
    uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

// ---------------------------------------------------------------------------------------------
// header parsing (one lane)
// ---------------------------------------------------------------------------------------------

// The 16 stream bytes starting at `pos`, fetched with aligned 8-byte loads.  Bytes past the end
// of the stream inside the last aligned word read as whatever follows; callers detect truncation
// from the run's end offset, never from these bytes.
struct Window16 {
    uint64_t lo, hi;
    __device__ __forceinline__ uint32_t byte(uint32_t i) const {
        return (uint32_t)((i < 8 ? lo >> (8 * i) : hi >> (8 * (i - 8))) & 0xFF);
    }
};

__device__ __forceinline__ Window16 load_window(const uint8_t *s, uint64_t len, uint64_t pos) {
    const uintptr_t a = (uintptr_t)(s + pos);
    const uintptr_t a0 = a & ~(uintptr_t)7;
    const uintptr_t end = (uintptr_t)(s + len);
    const uint32_t sh = (uint32_t)(a & 7) * 8;
    const uint64_t *p = (const uint64_t *)a0;
    // all three loads are issued together: a word past the end is replaced by a re-read of the
    // first word (select on the address, not on the load)
    const uint64_t q0 = p[0];
    const uint64_t q1 = *((a0 + 8 < end) ? p + 1 : p);
    const uint64_t q2 = *((a0 + 16 < end) ? p + 2 : p);
    Window16 w;
    w.lo = sh ? (q0 >> sh) | (q1 << (64 - sh)) : q0;
    w.hi = sh ? (q1 >> sh) | (q2 << (64 - sh)) : q1;
    return w;
}

struct ParsedRun {
    uint64_t payload_off, run_end, base, aux;
    uint32_t kind, width, len, delta_neg, header_bytes;
};

// LEB128 starting at byte `i` of the header window; bytes past the window come from memory.
__device__ __forceinline__ bool parse_varint(const Window16 &w, const uint8_t *s, uint64_t len,
                                             uint64_t pos, uint32_t &i, uint64_t &value) {
    uint64_t v = 0;
    for (uint32_t j = 0; j < 10; j++) {  // rle_v2.rs:71-89: more than 10 bytes is out of spec
        uint32_t b;
        if (i < 16) b = w.byte(i);
        else b = (pos + i < len) ? s[pos + i] : 0;
        i++;
        v |= (uint64_t)(b & 0x7F) << (7 * j);
        if ((b & 0x80) == 0) {
            value = v;
            return true;
        }
    }
    return false;
}

// Parses the run whose header starts at `pos` (pos < len) from its 16-byte window.  Returns a
// SEG_* status.
__device__ __forceinline__ uint32_t parse_run(const Window16 &w, const uint8_t *s, uint64_t len,
                                              uint64_t pos, int is_signed, ParsedRun &r) {
    const uint32_t h0 = w.byte(0), h1 = w.byte(1);
    r.kind = h0 >> 6;
    r.base = 0;
    r.aux = 0;
    r.delta_neg = 0;
    if (r.kind == KIND_SHORT_REPEAT) {  // rle_v2.rs:283-301
        const uint32_t nbytes = ((h0 >> 3) & 7) + 1;
        r.len = (h0 & 7) + 3;
        r.width = nbytes * 8;
        uint64_t v = 0;
        for (uint32_t i = 0; i < nbytes; i++) v = (v << 8) | w.byte(1 + i);
        r.base = is_signed ? zigzag(v) : v;
        r.payload_off = pos + 1;
        r.run_end = pos + 1 + nbytes;
    } else if (r.kind == KIND_DIRECT) {  // rle_v2.rs:135-157
        r.width = width_from_code((h0 >> 1) & 31);
        r.len = (((h0 & 1) << 8) | h1) + 1;
        r.payload_off = pos + 2;
        r.run_end = r.payload_off + ((uint64_t)r.width * r.len + 7) / 8;
    } else if (r.kind == KIND_DELTA) {  // rle_v2.rs:195-222, 343-366
        const uint32_t code = (h0 >> 1) & 31;
        r.width = code == 0 ? 0 : width_from_code(code);
        r.len = (((h0 & 1) << 8) | h1) + 1;
        uint32_t i = 2;
        uint64_t base, db;
        if (!parse_varint(w, s, len, pos, i, base)) return SEG_BAD_RUN;
        if (!parse_varint(w, s, len, pos, i, db)) return SEG_BAD_RUN;
        r.base = is_signed ? zigzag(base) : base;
        r.aux = zigzag(db);
        r.delta_neg = (int64_t)r.aux < 0;
        if (r.width != 0 && r.len < 2) return SEG_BAD_RUN;
        r.payload_off = pos + i;
        r.run_end = r.payload_off + (r.len < 2 ? 0 : ((uint64_t)r.width * (r.len - 2) + 7) / 8);
    } else {  // PatchedBase, ORC v1 specification
        const uint32_t h2 = w.byte(2), h3 = w.byte(3);
        r.width = width_from_code((h0 >> 1) & 31);
        r.len = (((h0 & 1) << 8) | h1) + 1;
        const uint32_t base_bytes = ((h2 >> 5) & 7) + 1;
        const uint32_t pw = width_from_code(h2 & 31);
        const uint32_t gw = ((h3 >> 5) & 7) + 1;
        const uint32_t pll = h3 & 31;
        if (pll == 0 || pw + gw > 64) return SEG_BAD_RUN;
        const uint32_t cfb = closest_fixed_bits(pw + gw);
        uint64_t b = 0;
        for (uint32_t i = 0; i < base_bytes; i++) b = (b << 8) | w.byte(4 + i);
        const uint64_t sign = 1ull << (base_bytes * 8 - 1);
        r.base = (b & sign) ? 0 - (b & ~sign) : b;  // sign-magnitude
        r.aux = make_pb_aux(pw, gw, pll, cfb);
        r.payload_off = pos + 4 + base_bytes;
        r.run_end = r.payload_off + ((uint64_t)r.width * r.len + 7) / 8 + ((uint64_t)cfb * pll + 7) / 8;
    }
    r.header_bytes = (uint32_t)(r.payload_off - pos);
    if (r.run_end > len) return SEG_TRUNCATED;
    if (r.payload_off > RUN_PAYLOAD_MASK) return SEG_TOO_LONG;
    return SEG_OK;
}

// The size in bytes of the run whose header the window starts with, by arithmetic alone: every
// kind worked out side by side and picked by selects, for lanes that look at byte offsets most of
// which are no headers at all (a branch per kind makes such a wavefront run all four, the loops
// over a Delta header's varints included).  0: not a header, not one whose run fits `room` bytes,
// or a Delta header whose two varints are not inside the window.  Only a hint: whoever acts on it
// checks with parse_run, which alone decides what is a run and what is out of spec.
__device__ __forceinline__ uint32_t quick_run_size(const Window16 &w, uint64_t room) {
    const uint32_t h = (uint32_t)w.lo;
    const uint32_t kind = (h >> 6) & 3u, code = (h >> 1) & 31u;
    const uint32_t wd = width_from_code(code);
    const uint32_t len9 = (((h & 1u) << 8) | ((h >> 8) & 0xFFu)) + 1u;
    const uint32_t packed = (wd * len9 + 7u) >> 3;
    const uint32_t size_sr = 2u + ((h >> 3) & 7u);
    const uint32_t size_direct = 2u + packed;
    // Delta: the continuation bits of bytes 2 .. 15 say how long the two varints are
    auto msb4 = [](uint32_t v) { return (((v >> 7) & 0x01010101u) * 0x01020408u) >> 24; };  // the top bits of 4 bytes as 4 bits
    const uint64_t x = (w.lo >> 16) | (w.hi << 48);  // bytes 2 .. 9
    const uint32_t c = msb4((uint32_t)x) | (msb4((uint32_t)(x >> 32)) << 4) | (msb4((uint32_t)(w.hi >> 16)) << 8) |
                       ((msb4((uint32_t)(w.hi >> 48)) & 3u) << 12);  // 14 bits
    const uint32_t l1 = (uint32_t)__builtin_ctz(~c) + 1u;  // at most 15
    const uint32_t l2 = (uint32_t)__builtin_ctz(~(c >> l1)) + 1u;
    const bool delta_ok = l1 <= 10u && l2 <= 10u && l1 + l2 <= 14u && !(code != 0 && len9 < 2u);
    const uint32_t size_delta = 2u + l1 + l2 + (len9 < 2u ? 0u : ((code ? wd : 0u) * (len9 - 2u) + 7u) >> 3);
    // PatchedBase
    const uint32_t h2 = (h >> 16) & 0xFFu, h3 = h >> 24;
    const uint32_t pw = width_from_code(h2 & 31u), gw = (h3 >> 5) + 1u, pll = h3 & 31u;
    const bool pb_ok = pll != 0 && pw + gw <= 64u;
    const uint32_t size_pb = 5u + (h2 >> 5) + packed + ((closest_fixed_bits(pw + gw) * pll + 7u) >> 3);
    const uint32_t size = kind == KIND_SHORT_REPEAT ? size_sr
                          : kind == KIND_DIRECT     ? size_direct
                          : kind == KIND_DELTA      ? (delta_ok ? size_delta : 0u)
                                                    : (pb_ok ? size_pb : 0u);
    return size <= room ? size : 0u;
}

// A lane's private stretch of stream bytes in LDS: where runs are short, the next few headers lie
// within the bytes fetched for the current one, and parsing them from here saves the memory round
// trip each would otherwise cost.
constexpr uint32_t HB_WORDS = 12;   // 96 bytes
constexpr uint32_t HB_STRIDE = 13;  // slice pitch in 8-byte words (odd: spreads the LDS banks)

struct HeaderBuf {
    uint64_t *slice;   // HB_WORDS words
    uintptr_t base;    // address the first word was loaded from (8-byte aligned)
    uint32_t words;    // how many are valid
};

// The 16-byte window at `pos` out of the lane's buffer, refilled from memory when the window is
// not inside it: with 3 words (just this window) or, `wide`, with all HB_WORDS.
__device__ __forceinline__ Window16 buffered_window(const uint8_t *s, uint64_t len, uint64_t pos,
                                                    HeaderBuf &hb, bool wide) {
    const uintptr_t a = (uintptr_t)(s + pos);
    const uintptr_t a0 = a & ~(uintptr_t)7;
    const uintptr_t end = (uintptr_t)(s + len);
    const uint32_t sh = (uint32_t)(a & 7) * 8;
    const uintptr_t idx = (a0 - hb.base) >> 3;  // huge when a0 lies before the buffer
    uint64_t q0, q1, q2;
    // A refill costs a memory round trip whether one lane needs it or all do, and the lanes of a
    // wavefront run out of buffered bytes at different headers: when one must go to memory, all
    // that are here refill, so that the round trips of a wavefront coincide.
    const bool hit = hb.words && a0 >= hb.base && idx + 3 <= hb.words;
    if (__builtin_amdgcn_ballot_w64(!hit) == 0) {
        q0 = hb.slice[idx];
        q1 = hb.slice[idx + 1];
        q2 = hb.slice[idx + 2];
    } else {
        const uint32_t n = wide ? HB_WORDS : 3u;
        const uint64_t *p = (const uint64_t *)a0;
        uint64_t v[HB_WORDS];
        // all loads are issued together; a word past the end is replaced by a re-read of the
        // first one (select on the address, not on the load)
#pragma unroll
        for (uint32_t i = 0; i < HB_WORDS; i++)
            v[i] = i < n ? *((a0 + 8 * i < end) ? p + i : p) : 0;
#pragma unroll
        for (uint32_t i = 0; i < HB_WORDS; i++)
            if (i < n) hb.slice[i] = v[i];
        hb.base = a0;
        hb.words = n;
        q0 = v[0];
        q1 = v[1];
        q2 = v[2];
    }
    Window16 w;
    w.lo = sh ? (q0 >> sh) | (q1 << (64 - sh)) : q0;
    w.hi = sh ? (q1 >> sh) | (q2 << (64 - sh)) : q1;
    return w;
}

// Walks the run headers of [pos, end), at most max_runs of them, calling consume(run) for each.
// Headers form a serial chain -- a header's position is known only once its predecessor is
// parsed -- so a plain walk pays one memory round trip per run.  Two ways around that:
//  * long runs of equal size (the bulk of a long column): every round loads the header at `pos`
//    together with three more at pos + k * (size of the previous run) and parses as many of the
//    four as the chain confirms.  A wrong guess costs the extra loads, so the guessing is
//    switched on by the segment's first run, kept on by runs of equal size and switched off by
//    the first that differs;
//  * short runs: the bytes fetched for one header hold the next ones too (buffered_window).
template <typename Consume>
__device__ __forceinline__ uint32_t walk_headers(const uint8_t *s, uint64_t len, uint64_t &pos,
                                                 uint64_t end, uint32_t max_runs, int is_signed,
                                                 uint64_t *lds_slice, Consume &&consume) {
    uint64_t stride = 0, prev_stride = ~0ull;
    uint32_t n = 0, status = SEG_OK;
    ParsedRun r;
    HeaderBuf hb = {lds_slice, 0, 0};
    auto step = [&](const Window16 &w) -> bool {
        status = parse_run(w, s, len, pos, is_signed, r);
        if (status != SEG_OK) return false;
        consume(r);
        prev_stride = stride;
        stride = r.run_end - pos;
        pos = r.run_end;
        n++;
        return true;
    };
    constexpr int GUESS = 4;  // headers fetched per round while the guesses come true
    while (pos < end && n < max_runs) {
        // (short runs are better served by the buffered bytes than by guesses that go to memory)
        const bool guess = stride > 8 * (HB_WORDS - 3) && (n == 1 || stride == prev_stride);
        if (!(guess && pos + stride < end)) {
            // fetch ahead only where the last run was short enough for the next header to be near
            const Window16 w0 = buffered_window(s, len, pos, hb, stride <= 8 * (HB_WORDS - 3));
            if (!step(w0)) break;
            continue;
        }
        // header k of the round is expected at pos + k * stride; one past the segment's end is
        // replaced by a re-read of the first (select on the address, not on the load)
        const uint64_t p0 = pos, st = stride;
        Window16 w[GUESS];
#pragma unroll
        for (int k = 0; k < GUESS; k++) {
            const uint64_t pk = p0 + (uint64_t)k * st;
            w[k] = load_window(s, len, pk < end ? pk : p0);
        }
        bool go = true;
#pragma unroll
        for (int k = 0; k < GUESS; k++) {
            if (!go) continue;
            // still on the guessed chain, inside the segment, and wanted?
            if (k > 0 && !(pos == p0 + (uint64_t)k * st && pos < end && n < max_runs)) {
                go = false;
                continue;
            }
            if (!step(w[k])) go = false;
        }
        if (status != SEG_OK) break;
    }
    return status;
}

// ---------------------------------------------------------------------------------------------
// K1: header scan -- one lane per segment.  COUNT pass: runs and values per segment, finished by
// a workgroup-local prefix sum.  EMIT pass (after the scan of the workgroup totals): the run
// table and, per 512-value window, its first run.
// ---------------------------------------------------------------------------------------------

template <bool EMIT>
__global__ __launch_bounds__(SEG_BLOCK) void rlev2_scan_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const uint64_t *__restrict__ entries,
    const uint64_t *__restrict__ seg_ends, uint32_t n_seg, int is_signed, SegTable t,
    uint64_t *__restrict__ runs, uint32_t *__restrict__ group_first,
    RunEntry *__restrict__ group_run, uint32_t n_group_slots, int full_table,
    uint64_t *__restrict__ staged, uint32_t staged_cap, uint32_t staged_up_to) {
    __shared__ uint64_t header_bytes[SEG_BLOCK * HB_STRIDE];
    __shared__ uint64_t staged_bytes[EMIT ? 1 : SEG_BLOCK * SB_STRIDE];
    uint64_t *lds_slice = header_bytes + threadIdx.x * HB_STRIDE;
    const uint32_t seg = blockIdx.x * SEG_BLOCK + threadIdx.x;
    const bool valid = seg < n_seg;
    // A segment runs from its entry to the next entry -- or, when seg_ends is given (slices of a
    // larger buffer decoded as one stream, with gaps between them), to its own end.  Without
    // seg_ends the first entry is taken to be offset 0 whatever its value.
    uint64_t pos = (valid && entries && (seg > 0 || seg_ends)) ? entries[seg] : 0;
    uint64_t end = (valid && entries && seg + 1 < n_seg) ? entries[seg + 1] : len;
    if (valid && seg_ends) end = seg_ends[seg];
    // a slice is a stream of its own: a run that reaches past its end is cut short, whatever
    // bytes follow it in the buffer
    const uint64_t limit = seg_ends ? (end < len ? end : len) : len;
    const bool last_seg = seg + 1 == n_seg;
    if (pos > len) pos = len;

    if (!EMIT) {
        uint32_t n_runs = 0, last_len = 0, status = SEG_OK;
        uint64_t n_vals = 0;
        if (end > len || end < pos) status = SEG_MISALIGNED;
        if (valid && status == SEG_OK) {
            StagedWriter keep = {staged ? staged + (uint64_t)seg * staged_cap : nullptr,
                                 staged_bytes + threadIdx.x * SB_STRIDE, staged_cap};
            status = walk_headers(s, limit, pos, end, 0xFFFFFFFFu, is_signed, lds_slice, [&](const ParsedRun &r) {
                keep.put(n_runs, make_desc(r.payload_off, r.kind, r.width, r.len, r.delta_neg, r.header_bytes));
                n_runs++;
                n_vals += r.len;
                last_len = r.len;
            });
            keep.finish(n_runs);
        }
        if (status == SEG_OK && pos != end && !last_seg) status = SEG_MISALIGNED;
        seg_publish(t, seg, n_seg, n_runs, last_len, status, n_vals);
    } else {
        if (!valid) return;
        const uint32_t n_runs = t.seg_count[seg].n_runs;
        if (staged_up_to && n_runs <= staged_up_to) return;  // emitted from the staged descriptors
        const SegOffset so = seg_global_offset(t, seg);
        uint64_t out_off = so.val_off;
        uint32_t run_idx = so.run_off;
        uint32_t prev_len = so.prev_len;  // 0 only before the very first run
        // the counting pass found n_runs well-formed runs from here
        walk_headers(s, limit, pos, end, n_runs, is_signed, lds_slice, [&](const ParsedRun &r) {
            RunEntry e;
            e.desc = make_desc(r.payload_off, r.kind, r.width, r.len, r.delta_neg, r.header_bytes);
            e.out_off = out_off;
            e.base = r.base;
            e.aux = r.aux;
            // first run that starts inside its 512-value window?  and the only one?
            const uint64_t win = out_off >> WINDOW_SHIFT;
            const bool first = (out_off == 0) || (((out_off - prev_len) >> WINDOW_SHIFT) != win);
            const bool single = first && ((out_off + r.len) >> WINDOW_SHIFT) != win;
            // The value kernels read a single-run window's run from group_run only: its slot in
            // the run table stays unwritten unless the table itself is what the caller wants.
            if (!single || full_table) runs[run_idx] = e.desc;
            if (first && win < n_group_slots) {
                group_first[win] = run_idx;
                // the window's own copy, so that the value kernel needs no dependent table load
                RunEntry ge = e;
                if (single) ge.desc |= DESC_SINGLE;
                else ge.base = run_idx;  // a shared window: where its runs start in the table
                group_run[win] = ge;
            }
            out_off += r.len;
            prev_len = r.len;
            run_idx++;
        });
        if (last_seg) {
            // sentinel: one-past-the-end run and the window after the last value
            RunEntry e;
            e.desc = make_desc(pos, KIND_SHORT_REPEAT, 0, 1, 0, 1);
            e.out_off = out_off;
            e.base = run_idx;
            e.aux = 0;
            runs[run_idx] = e.desc;
            // Windows in which no run starts exist only at the tail: the last run may reach
            // into one more window.  They, and the slot one past the last window, get the
            // sentinel's index (an empty group).
            const uint64_t g_end = (out_off + WINDOW - 1) >> WINDOW_SHIFT;
            uint64_t g = out_off == 0 ? 0 : ((out_off - prev_len) >> WINDOW_SHIFT) + 1;
            for (; g <= g_end && g < n_group_slots; g++) {
                group_first[g] = run_idx;
                group_run[g] = e;
            }
        }
    }
}

// COUNT for streams of few segments: a wavefront per segment instead of a lane.  One lane parses a
// header in about half a microsecond, and a segment of thousands of short runs -- or a whole
// stream that came without entry points -- is thousands of those in a row.  Here the lanes parse,
// at once, the 64 byte offsets that follow the current header (or, after a run of 64 bytes and
// more, 64 guesses at that run's stride); the first offset is a real header, and the chain that
// starts there is picked out of the 64 guesses by pointer doubling (five rounds of lane
// permutes; a scalar loop that hops along it costs a readlane and a branch per run).  Only the
// offsets on the chain are parsed in full -- the others are sized by quick_run_size alone -- and
// their descriptors collect in an LDS ring and leave for the staging area 64 at a time.
// (A stripe of a file, 450 segments of 2 100 short runs each: 313 -> 140 microseconds, of which
// the bytes in LDS 105, the descriptors kept instead of a second parse and the sizing by
// arithmetic the rest.)
// A segment whose bytes fit the wavefront's LDS slice is copied there first (loads that write LDS
// themselves) and every window of 16 header bytes is then put together from LDS reads: a round
// of 64 offsets no longer waits for memory.  Longer segments are parsed from memory as before.
// One wavefront per workgroup; dynamic LDS: COOP_RING * 8 + slice_bytes + 32.
// Writes t.seg_count only; rlev2_publish_kernel turns that into the scan's inputs.
constexpr uint32_t COOP_RING = 128;
constexpr uint32_t COUNT_SLICE_BYTES = 48 << 10;  // stream bytes a wavefront of rlev2_count_staged_kernel holds

__global__ __launch_bounds__(64) void rlev2_coop_count_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const uint64_t *__restrict__ entries,
    const uint64_t *__restrict__ seg_ends, uint32_t n_seg, int is_signed, SegTable t,
    uint64_t *__restrict__ staged, uint32_t staged_cap, uint32_t slice_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t coop_dyn[];
    uint64_t *ring = (uint64_t *)coop_dyn;
    uint8_t *slice = coop_dyn + COOP_RING * sizeof(uint64_t);
    const uint32_t lane = threadIdx.x;
    const uint32_t seg = blockIdx.x;
    if (seg >= n_seg) return;  // whole wavefront
    // the segment's bounds, as rlev2_scan_kernel takes them
    uint64_t pos = (entries && (seg > 0 || seg_ends)) ? entries[seg] : 0;
    uint64_t end = (entries && seg + 1 < n_seg) ? entries[seg + 1] : len;
    if (seg_ends) end = seg_ends[seg];
    const uint64_t limit = seg_ends ? (end < len ? end : len) : len;
    const bool last_seg = seg + 1 == n_seg;
    if (pos > len) pos = len;
    pos = readfirstlane64(pos);
    end = readfirstlane64(end);
    uint32_t n_runs = 0, last_len = 0, status = SEG_OK, ring_n = 0;
    uint64_t n_vals = 0, last_size = 0;
    if (end > len || end < pos) status = SEG_MISALIGNED;
    uint64_t *mine = staged ? staged + (uint64_t)seg * staged_cap : nullptr;
    // the segment's bytes and the 20 after them (a header at the segment's last byte is looked at
    // as 16 bytes + 4 of alignment), as far as the stream goes
    const uint64_t lo = pos;
    const uint32_t skew = (uint32_t)((uintptr_t)(s + lo) & 15);
    const bool in_lds = status == SEG_OK && end > lo && end - lo + skew + 24 <= slice_bytes;
    if (in_lds) {
        const uintptr_t a0 = (uintptr_t)(s + lo) - skew, stop = (uintptr_t)(s + len);
        const uint32_t n16 = (uint32_t)((end - lo + skew + 24 + 15) >> 4);  // <= slice_bytes / 16 + 1 (the slice has 32 bytes to spare)
        for (uint32_t c0 = 0; c0 < n16; c0 += 64) {
            const uint32_t c = c0 + lane;
            if (c < n16 && a0 + 16 * (uintptr_t)c < stop)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a0 + 16 * (uintptr_t)c),
                                                 (__attribute__((address_space(3))) void *)(slice + 16 * c0), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // (two copies of the walk, so that the one over LDS carries no test of where its bytes are)
    auto walk = [&](auto from_lds) {
        constexpr bool IN_LDS = decltype(from_lds)::value;
        const uint32_t *words = (const uint32_t *)slice;
        auto window_at = [&](uint64_t p) -> Window16 {
            if (!IN_LDS) return load_window(s, limit, p);
            // (bytes past the stream's end: whatever the slice holds there; a run that needs them
            // is found truncated by its end, as with load_window)
            const uint32_t o = (uint32_t)(p - lo) + skew, k = o >> 2, sh = o & 3u;
            const uint32_t w0 = words[k], w1 = words[k + 1], w2 = words[k + 2], w3 = words[k + 3], w4 = words[k + 4];
            Window16 w;
            w.lo = ((uint64_t)__builtin_amdgcn_alignbyte(w2, w1, sh) << 32) | __builtin_amdgcn_alignbyte(w1, w0, sh);
            w.hi = ((uint64_t)__builtin_amdgcn_alignbyte(w4, w3, sh) << 32) | __builtin_amdgcn_alignbyte(w3, w2, sh);
            return w;
        };
        for (;;) {
            // ---- find headers until 64 are waiting or the segment ends -------------------
            while (status == SEG_OK && pos < end && ring_n < 64) {
                const bool strided = last_size >= 64;
                const uint64_t stride = strided ? last_size : 1;
                const uint64_t p = pos + (uint64_t)lane * stride;
                const bool inb = p < end;
                Window16 w = {0, 0};
                if (inb) w = window_at(p);
                uint32_t st = SEG_OK;
                uint64_t size = 0, desc = 0;
                auto parse = [&]() {
                    ParsedRun r;
                    st = parse_run(w, s, limit, p, is_signed, r);
                    size = r.run_end - p;
                    desc = make_desc(r.payload_off, r.kind, r.width, r.len, r.delta_neg, r.header_bytes);
                };
                uint32_t accepted;
                if (strided) {
                    if (inb) parse();
                    const uint32_t st0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)st);
                    if (st0 != SEG_OK) {  // the header at pos is a real one: the walk stops here
                        status = st0;
                        break;
                    }
                    // offset k * stride is a header if every run before it was `stride` long
                    const uint64_t same = __ballot(inb && st == SEG_OK && size == stride);
                    const uint32_t lead = same == ~0ull ? 64u : (uint32_t)__ffsll((unsigned long long)~same) - 1u;
                    if (lead == 0) {  // the run at pos has another size: it alone is taken
                        accepted = 1;
                        last_size = readfirstlane64(size);
                        if (lane == 0) ring[ring_n] = desc;
                        pos += last_size;
                    } else {
                        accepted = lead;
                        if (lane < lead) ring[ring_n + lane] = desc;
                        pos += (uint64_t)lead * stride;
                    }
                } else {
                    // Every offset's guess at where the next header lies, then the chain from
                    // offset 0 (the one real header) by pointer doubling: after k steps lane i
                    // holds the first 2^k offsets of the chain that starts at i.  A run takes two
                    // bytes at least, so 32 links cover the 64 offsets.
                    const uint32_t q = inb ? quick_run_size(w, limit - p) : 0u;
                    uint32_t jump = (q != 0 && lane + q < 64) ? lane + q : lane;  // (itself: the chain stops here)
                    uint64_t links = q != 0 ? 1ull << lane : 0;
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        links |= shfl64(links, (int)jump);
                        jump = __shfl(jump, (int)jump);
                    }
                    uint64_t chain = readfirstlane64(links);
                    if (chain == 0) chain = 1;  // no guess for the header at pos: parse_run says what it is
                    // the offsets on the chain are parsed in full: nothing is taken on the guess alone
                    if ((chain >> lane) & 1) parse();
                    const uint64_t off = __ballot(((chain >> lane) & 1) && (st != SEG_OK || size != q));
                    if (off) {
                        // the chain is cut at the first offset whose run is not what was guessed: a
                        // run of another size is taken (it is the real one), whatever does not
                        // parse is reported once it is the header at pos
                        const uint32_t f = (uint32_t)__ffsll((unsigned long long)off) - 1u;
                        const uint32_t st_f = readlane32(st, (int)f);
                        chain &= (1ull << f) - 1;
                        if (st_f == SEG_OK) chain |= 1ull << f;
                        if (chain == 0) {
                            status = st_f;
                            break;
                        }
                    }
                    const uint32_t last = 63u - (uint32_t)__builtin_clzll((unsigned long long)chain);
                    last_size = readlane64(size, (int)last);
                    pos += last + last_size;  // where the next header is, inside these 64 offsets or past them
                    accepted = (uint32_t)__popcll((unsigned long long)chain);
                    if ((chain >> lane) & 1) ring[ring_n + (uint32_t)__popcll((unsigned long long)(chain & ((1ull << lane) - 1)))] = desc;
                }
                ring_n += accepted;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (ring_n == 0) break;
            // ---- the next (up to) 64 descriptors leave the ring ------------------------------
            const uint32_t take = ring_n < 64 ? ring_n : 64;
            uint32_t rlen = 0;
            if (lane < take) {
                const uint64_t d = ring[lane];
                rlen = desc_len(d);
                if (mine && n_runs + lane < staged_cap) mine[n_runs + lane] = d;
            }
            uint32_t sum = rlen;
            for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
            n_vals += sum;
            last_len = readlane32(rlen, (int)take - 1);
            n_runs += take;
            const uint32_t rest = ring_n - take;  // at most 63
            const uint64_t moved = lane < rest ? ring[take + lane] : 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) ring[lane] = moved;
            ring_n = rest;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };
    if (in_lds) walk(std::true_type{});
    else walk(std::false_type{});
    if (status == SEG_OK && pos != end && !last_seg) status = SEG_MISALIGNED;
    if (lane == 0) {
        SegCount c;
        c.n_runs = n_runs;
        c.last_len = last_len;
        c.status = status;
        c.pad = 0;
        c.n_vals = n_vals;
        t.seg_count[seg] = c;
    }
}

// Entry points for a stream of short runs that came without any and is not regular: entry k is a
// guess at the first header at or after k * spacing.  One wavefront per entry walks 64 header
// chains at once, lane l from (k - 1) * spacing + l, one segment in front of where the answer is
// wanted.  Most of those offsets are no headers, and a walk that starts off the chain parses
// garbage (where a header does not parse, it tries the next byte) -- but it falls in with the true
// chain as soon as it lands on one real header and stays on it, so the walks merge, and with runs
// of a few bytes several of the 64 start on real headers to begin with.  The position that most
// walks have reached when all have passed k * spacing is the guess.  Nothing is trusted: the
// counting pass that follows walks every segment from its entry and demands that it ends exactly
// at the next one, and entry 0 is the start of the stream, so a set of entries that passes is, by
// induction, on the true chain; one that does not is discarded and the stream walked as a single
// segment.  (Streams of long runs give such walks little to land on: they are not what this is
// for -- their serial walk is short, a hop per kilobyte.)
__global__ __launch_bounds__(256) void rlev2_resync_kernel(const uint8_t *__restrict__ s, uint64_t len,
                                                           int is_signed, uint64_t spacing, uint32_t n_entries,
                                                           uint64_t *__restrict__ entries) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n_entries) return;  // whole wavefront
    if (k == 0) {
        if (lane == 0) entries[0] = 0;
        return;
    }
    uint64_t target = (uint64_t)k * spacing;
    if (target > len) target = len;
    uint64_t pos = (uint64_t)(k - 1) * spacing + lane;
    while (__ballot(pos < target) != 0) {
        if (pos < target) {
            ParsedRun r;
            const uint32_t st = parse_run(load_window(s, len, pos), s, len, pos, is_signed, r);
            pos = (st == SEG_OK && r.run_end > pos) ? r.run_end : pos + 1;
        }
    }
    if (pos > len) pos = len;
    // the position most walks agree on
    uint32_t best_votes = 0;
    uint64_t best = pos;
    for (int j = 0; j < 64; j++) {
        const uint64_t cand = readlane64(pos, j);
        const uint32_t votes = (uint32_t)__builtin_popcountll(__ballot(pos == cand));
        if (votes > best_votes) {
            best_votes = votes;
            best = cand;
        }
    }
    if (lane == 0) entries[k] = best;
}

// What rlev2_coop_count_kernel left in t.seg_count, through the workgroup-local scan that the
// counting kernel proper ends with.
__global__ __launch_bounds__(SEG_BLOCK) void rlev2_publish_kernel(uint32_t n_seg, SegTable t) {
    const uint32_t seg = blockIdx.x * SEG_BLOCK + threadIdx.x;
    SegCount c;
    c.n_runs = 0;
    c.last_len = 0;
    c.status = SEG_OK;
    c.n_vals = 0;
    if (seg < n_seg) c = t.seg_count[seg];
    seg_publish(t, seg, n_seg, c.n_runs, c.last_len, c.status, c.n_vals);
}

// COUNT for streams of many small segments (short runs between close entry points): a wavefront
// takes `group` neighbouring segments, copies the bytes they span -- one contiguous piece of the
// stream -- into its slice of LDS with loads that write LDS themselves, and `group` of its lanes
// then walk a segment each out of LDS.  The lane-per-segment walk of rlev2_scan_kernel spends a
// load of its own per lane on every few headers, 64 cache lines per instruction; here every line
// of the stream is fetched once, by a load that uses all of it, and a header costs two LDS reads.
// With its bytes in LDS a wavefront is alone on its SIMD most of the time, and what a step then
// costs is its instructions one after the other, every branch an empty instruction buffer: so
// ShortRepeat and Direct runs, which such columns are made of and whose size follows from their
// first two bytes, are sized side by side without the general parser and picked by selects (the
// counting pass wants a run's descriptor and its end, a fraction of what parse_run works out), and
// the descriptors go out one by one instead of through an LDS collector.  Delta and PatchedBase
// runs go through parse_run on 16 bytes out of LDS.  A wavefront whose segments span more than its
// slice walks them from memory.  Positions are 32-bit offsets from the span's start.
// (5 * 10^8 values in 2.5 * 10^7 runs, 64 to a segment: 0.39 -> 0.23 ms.  How long the pass takes
// is the number of times the chip's LDS fills with stream bytes times a segment's runs times a
// step: 32 segments to a wavefront with half the slice, or 16, measured the same or slower.)
// Writes t.seg_count only; rlev2_publish_kernel turns that into the scan's inputs.
// Dynamic LDS per wavefront: slice_bytes + 32.
__global__ __launch_bounds__(64) void rlev2_count_staged_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const uint64_t *__restrict__ entries,
    const uint64_t *__restrict__ seg_ends, uint32_t n_seg, int is_signed, SegTable t,
    uint64_t *__restrict__ staged, uint32_t staged_cap, uint32_t group, uint32_t slice_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t cs_dyn[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n_waves = blockDim.x >> 6;
    const uint32_t slice_pitch = slice_bytes + 32;
    uint8_t *slice = cs_dyn + (size_t)wave * slice_pitch;
    const uint64_t first64 = ((uint64_t)blockIdx.x * n_waves + wave) * group;
    if (first64 >= n_seg) return;  // whole wavefront; no workgroup barrier below
    const uint32_t seg = (uint32_t)first64 + lane;
    const bool valid = lane < group && seg < n_seg;
    // a segment's bounds: as in rlev2_scan_kernel
    uint64_t pos = (valid && entries && (seg > 0 || seg_ends)) ? entries[seg] : 0;
    uint64_t end = (valid && entries && seg + 1 < n_seg) ? entries[seg + 1] : len;
    if (valid && seg_ends) end = seg_ends[seg];
    const uint64_t limit = seg_ends ? (end < len ? end : len) : len;
    const bool last_seg = seg + 1 == n_seg;
    if (pos > len) pos = len;
    uint32_t status = SEG_OK;
    if (end > len || end < pos) status = SEG_MISALIGNED;
    const bool walking = valid && status == SEG_OK && pos < end;
    // what the walking lanes span
    uint64_t lo = walking ? pos : ~0ull, hi = walking ? end : 0;
    for (int d = 32; d >= 1; d >>= 1) {
        const uint64_t ol = shfl64(lo, (int)lane ^ d), oh = shfl64(hi, (int)lane ^ d);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    lo = readfirstlane64(lo);
    hi = readfirstlane64(hi);
    const uint32_t skew = (uint32_t)((uintptr_t)(s + (hi > lo ? lo : 0)) & 15);
    const bool in_lds = hi > lo && hi - lo + skew + 4 <= slice_bytes;
    if (in_lds) {
        const uintptr_t a0 = (uintptr_t)(s + lo) - skew, stop = (uintptr_t)(s + len);
        // (three bytes past the span may be looked at with a header's four)
        const uint32_t n16 = (uint32_t)((hi - lo + skew + 3 + 15) >> 4);
        // loads that write LDS themselves, 1 KiB per instruction (lane i's 16 bytes land at the
        // instruction's LDS address + 16 i): none of them waits for another or holds a register
        for (uint32_t c0 = 0; c0 < n16; c0 += 64) {
            const uint32_t c = c0 + lane;
            if (c < n16 && a0 + 16 * (uintptr_t)c < stop)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a0 + 16 * (uintptr_t)c),
                                                 (__attribute__((address_space(3))) void *)(slice + 16 * c0), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t n_runs = 0, last_len = 0;
    uint64_t n_vals = 0;
    // the descriptors go out one by one (eight bytes a run, eight runs to a 64-byte piece of a line
    // that L2 puts together): collecting them in LDS first costs the walk more than it saves here
    uint64_t *mine = staged + (uint64_t)seg * staged_cap;
    const uint32_t cap_mine = (staged && valid) ? staged_cap : 0;
    if (walking) {
        const uint32_t *words = (const uint32_t *)slice;
        // offsets from lo: below 2^32 when the span is in LDS; otherwise only while the segment is
        // (a longer one stops as SEG_TOO_LONG would: the host sends such streams elsewhere)
        uint32_t rel = (uint32_t)(pos - lo);
        const uint32_t rel_end = (uint32_t)(end - lo);
        const uint32_t rel_limit = limit - lo > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(limit - lo);
        if ((!in_lds && (end - lo) > 0xFFFF0000ull) || end + 2 > RUN_PAYLOAD_MASK) status = SEG_TOO_LONG;
        // (two copies of the loop, so that the one over LDS carries no test of where its bytes are)
        auto walk = [&](auto from_lds) {
            constexpr bool IN_LDS = decltype(from_lds)::value;
            while (status == SEG_OK && rel < rel_end) {
                const uint64_t p = lo + rel;  // the header's place in the stream
                uint32_t h;
                if (IN_LDS) {
                    const uint32_t o = rel + skew;
                    h = __builtin_amdgcn_alignbyte(words[(o >> 2) + 1], words[o >> 2], o & 3u);
                } else {
                    const uintptr_t a = (uintptr_t)(s + p), a4 = a & ~(uintptr_t)3;
                    const uint32_t d0 = *(const uint32_t *)a4;
                    const uint32_t d1 = *(const uint32_t *)(a4 + 4 < (uintptr_t)(s + len) ? a4 + 4 : a4);
                    h = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)(a & 3));
                }
                const uint32_t kind = (h >> 6) & 3u;
                if (kind > KIND_DIRECT) {
                    // Delta and PatchedBase: variable-length fields decide where the run ends
                    Window16 w;
                    if (IN_LDS) {
                        const uint32_t o = rel + skew, k = o >> 2, sh = o & 3u;
                        // (up to 16 bytes past the span: whatever the slice holds there; a run that
                        // needs them is found truncated by its end, as with load_window)
                        const uint32_t w0 = words[k], w1 = words[k + 1], w2 = words[k + 2], w3 = words[k + 3], w4 = words[k + 4];
                        w.lo = ((uint64_t)__builtin_amdgcn_alignbyte(w2, w1, sh) << 32) | __builtin_amdgcn_alignbyte(w1, w0, sh);
                        w.hi = ((uint64_t)__builtin_amdgcn_alignbyte(w4, w3, sh) << 32) | __builtin_amdgcn_alignbyte(w3, w2, sh);
                    } else {
                        w = load_window(s, limit, p);
                    }
                    ParsedRun r;
                    const uint32_t st = parse_run(w, s, limit, p, is_signed, r);
                    if (st != SEG_OK) {
                        status = st;
                        break;
                    }
                    if (n_runs < cap_mine) mine[n_runs] = make_desc(r.payload_off, r.kind, r.width, r.len, r.delta_neg, r.header_bytes);
                    n_runs++;
                    n_vals += r.len;
                    last_len = r.len;
                    if (r.run_end - lo > 0xFFFFFFFFull) {
                        status = SEG_MISALIGNED;  // (past every segment's end)
                        break;
                    }
                    rel = (uint32_t)(r.run_end - lo);
                    continue;
                }
                // both kinds worked out side by side and picked by selects: a wavefront that walks
                // alone on its SIMD pays for every branch with an empty instruction buffer
                const bool is_sr = kind == KIND_SHORT_REPEAT;
                const uint32_t nbytes = ((h >> 3) & 7u) + 1u;                       // rle_v2.rs:283-301
                const uint32_t code = (h >> 1) & 31u, t24 = code - 24u;              // Direct, rle_v2.rs:135-157
                const uint32_t w_d = code <= 23u ? code + 1u : (t24 < 4u ? 26u + 2u * t24 : 8u + 8u * t24);  // width_from_code
                const uint32_t len_d = (((h & 1u) << 8) | ((h >> 8) & 0xFFu)) + 1u;
                const uint32_t rl = is_sr ? (h & 7u) + 3u : len_d;
                const uint32_t width = is_sr ? nbytes * 8u : w_d;
                const uint32_t hdr = is_sr ? 1u : 2u;
                const uint32_t size = is_sr ? 1u + nbytes : 2u + ((w_d * len_d + 7u) >> 3);
                if (size > rel_limit - rel) {  // (rel < rel_end <= rel_limit)
                    status = SEG_TRUNCATED;
                    break;
                }
                if (n_runs < cap_mine) mine[n_runs] = make_desc(p + hdr, kind, width, rl, 0, hdr);
                n_runs++;
                n_vals += rl;
                last_len = rl;
                rel += size;
            }
        };
        if (in_lds) walk(std::true_type{});
        else walk(std::false_type{});
        if (status == SEG_OK && rel != rel_end && !last_seg) status = SEG_MISALIGNED;
    }
    if (valid) {
        SegCount c;
        c.n_runs = n_runs;
        c.last_len = last_len;
        c.status = status;
        c.pad = 0;
        c.n_vals = n_vals;
        t.seg_count[seg] = c;
    }
}

__device__ __forceinline__ uint32_t wave_scan_dpp(uint32_t v);

// EMIT for the segments whose descriptors the counting pass staged: a group of lanes per segment,
// one lane per run (a segment with more runs than lanes takes them a group at a time).  Nothing
// here waits on the header chain: a run's place follows from a scan of the lengths, and the base /
// second field of a run that has a window to itself from parsing that one header.
__global__ __launch_bounds__(256) void rlev2_emit_staged_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const uint64_t *__restrict__ entries,
    const uint64_t *__restrict__ seg_ends, uint32_t n_seg, int is_signed, SegTable t,
    const uint64_t *__restrict__ staged, uint32_t staged_cap, uint64_t *__restrict__ runs,
    uint32_t *__restrict__ group_first, RunEntry *__restrict__ group_run, uint32_t n_group_slots,
    int full_table, uint32_t group, uint32_t sets) {
    // `group` lanes per segment (a power of two from 8 to 64): a wavefront takes 64 / group
    // segments at a time, and two such sets one after the other -- what both need from memory
    // is asked for before either is worked on: a wavefront that emits 64 runs lives for one memory
    // round trip and little else, and 25 million runs are 400 000 of them.  (2.5 * 10^7 runs:
    // 0.138 -> 0.128 ms with two sets, 0.117 ms with the scan of a full-wavefront group by DPP.)
    // (`sets` = 1: one set only, for tables of few segments, whose wavefronts are too few to
    // keep the chip busy as it is)
    const uint32_t per_pass = 256 / group;  // segments a workgroup takes per pass
    const uint32_t lane = threadIdx.x & (group - 1);
    const uint64_t seg_a = (uint64_t)blockIdx.x * sets * per_pass + threadIdx.x / group, seg_b = seg_a + per_pass;
    if (seg_a >= n_seg) return;
    const bool has_b = sets == 2 && seg_b < n_seg;
    // (a staged slot that was never written holds whatever it held; it is not looked at)
    const uint64_t word_a = staged[seg_a * staged_cap + lane];  // staged_cap >= 64 >= group
    const uint64_t word_b = has_b ? staged[seg_b * staged_cap + lane] : 0;
    const uint32_t n_runs_a = t.seg_count[seg_a].n_runs, n_runs_b = has_b ? t.seg_count[seg_b].n_runs : 0;
    const SegOffset so_a = seg_global_offset(t, (uint32_t)seg_a);
    const SegOffset so_b = seg_global_offset(t, (uint32_t)(has_b ? seg_b : seg_a));
    const uint64_t end_a = seg_ends ? seg_ends[seg_a] : len, end_b = (seg_ends && has_b) ? seg_ends[seg_b] : len;
    auto emit_segment = [&](const uint32_t seg, uint64_t staged_word, const uint32_t n_runs, const SegOffset so,
                            const uint64_t seg_end) {
    const uint64_t *mine = staged + (uint64_t)seg * staged_cap;
    if (n_runs > staged_cap) return;  // the walking kernel emits this one
    const bool last_seg = seg + 1 == n_seg;
    if (n_runs == 0 && !last_seg) return;
    // the stream as this segment sees it (see rlev2_scan_kernel)
    const uint64_t limit = seg_end < len ? seg_end : len;
    uint64_t before = so.val_off;        // values before the chunk's first run
    uint32_t carry_prev = so.prev_len;   // length of the run before the chunk's first
    auto chunk = [&](const uint32_t base) {
        const uint32_t k = base + lane;
        const bool live = k < n_runs;
        const uint64_t desc = live ? staged_word : 0;
        const uint32_t rlen = live ? desc_len(desc) : 0;
        uint32_t incl = rlen;  // inclusive scan of the lengths
        if (group == 64) {  // (six DPP additions; the shuffles below are a permute and a select each)
            incl = wave_scan_dpp(rlen);
        } else {
            for (uint32_t d = 1; d < group; d <<= 1) {
                const uint32_t u = __shfl_up(incl, d, (int)group);
                if (lane >= d) incl += u;
            }
        }
        uint32_t prev_len = __shfl_up(rlen, 1, (int)group);
        if (lane == 0) prev_len = carry_prev;
        const uint64_t out_off = before + (incl - rlen);
        const uint32_t run_idx = so.run_off + k;
        const uint64_t win = out_off >> WINDOW_SHIFT;
        // the first run to start in its window: the run before it (at most 512 values) started in
        // an earlier one; alone in its window: it reaches the next one.  (In 32 bits: the pass is
        // bound by the instructions it issues, 3.3 per run.)
        const uint32_t in_win = ((uint32_t)before + (incl - rlen)) & (WINDOW - 1);
        const bool first = live && ((before == 0 && incl == rlen) || in_win < prev_len);
        const bool single = first && in_win + rlen >= WINDOW;
        const bool is_tail = last_seg && k + 1 == (n_runs ? n_runs : 1u);  // the lane that writes the sentinel
        // the header again: for the base and second field of a run that has a window to itself (a
        // Direct run has neither; the runs of a shared window are parsed by the value kernel's
        // lanes, which never look at these fields of the per-window copy) and for the end of the
        // stream's last run
        ParsedRun r;
        r.base = 0;
        r.aux = 0;
        r.run_end = 0;
        if (live && ((single && win < n_group_slots && desc_kind(desc) != KIND_DIRECT) || is_tail)) {
            const uint64_t pos = desc_payload(desc) - desc_header_bytes(desc);
            (void)parse_run(load_window(s, limit, pos), s, limit, pos, is_signed, r);  // parsed fine once already
        }
        if (live && (!single || full_table)) runs[run_idx] = desc;
        if (first && win < n_group_slots) {
            group_first[win] = run_idx;
            RunEntry ge;
            ge.desc = single ? desc | DESC_SINGLE : desc;
            ge.out_off = out_off;
            ge.base = single ? r.base : run_idx;  // a shared window: where its runs start in the table
            ge.aux = r.aux;
            group_run[win] = ge;
        }
        if (is_tail) {
            // sentinel: one-past-the-end run and the windows after the last run start (see
            // rlev2_scan_kernel)
            uint64_t pos = r.run_end;
            if (n_runs == 0) {
                pos = (entries && (seg > 0 || seg_ends)) ? entries[seg] : 0;
                if (pos > len) pos = len;
            }
            const uint64_t end_off = out_off + rlen;
            const uint32_t end_prev = n_runs ? rlen : so.prev_len;
            const uint32_t end_idx = so.run_off + n_runs;
            RunEntry e;
            e.desc = make_desc(pos, KIND_SHORT_REPEAT, 0, 1, 0, 1);
            e.out_off = end_off;
            e.base = end_idx;
            e.aux = 0;
            runs[end_idx] = e.desc;
            const uint64_t g_end = (end_off + WINDOW - 1) >> WINDOW_SHIFT;
            uint64_t g = end_off == 0 ? 0 : ((end_off - end_prev) >> WINDOW_SHIFT) + 1;
            for (; g <= g_end && g < n_group_slots; g++) {
                group_first[g] = end_idx;
                group_run[g] = e;
            }
        }
        // on to the next chunk: totals of this one, from the group's last lane
        if (n_runs > group) {
            before += __shfl(incl, (int)group - 1, (int)group);
            carry_prev = __shfl(rlen, (int)group - 1, (int)group);
        }
    };
    // a segment with more runs than the group has lanes: a group at a time, the words of the two
    // chunks ahead already asked for (one load per chunk and nothing else to wait for: a file's
    // segment of 2 000 runs is 32 dependent round trips otherwise)
    // (base + lane < staged_cap for base < n_runs: both are multiples of group)
    uint64_t ahead1 = 0, ahead2 = 0;
    if (n_runs > group) {
        ahead1 = mine[group + lane];
        if (2 * group < n_runs) ahead2 = mine[2 * group + lane];
    }
    chunk(0);
    for (uint32_t base = group; base < n_runs; base += group) {
        staged_word = ahead1;
        ahead1 = ahead2;
        if (base + 2 * group < n_runs) ahead2 = mine[base + 2 * group + lane];
        chunk(base);
    }
    };
    emit_segment((uint32_t)seg_a, word_a, n_runs_a, so_a, end_a);
    if (has_b) emit_segment((uint32_t)seg_b, word_b, n_runs_b, so_b, end_b);
}

// ---------------------------------------------------------------------------------------------
// K2: prefix sums over the per-workgroup totals of the counting pass -- one workgroup.
// ---------------------------------------------------------------------------------------------

constexpr int SCAN_THREADS = 1024;

__global__ __launch_bounds__(SCAN_THREADS) void block_totals_scan_kernel(SegTable t, uint32_t n_blocks,
                                                                         uint64_t num_values) {
    __shared__ uint64_t w_vals[SCAN_THREADS / 64];
    __shared__ uint64_t w_runs[SCAN_THREADS / 64];
    __shared__ uint64_t w_key[SCAN_THREADS / 64];
    __shared__ uint32_t first_bad;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) first_bad = 0xFFFFFFFFu;
    __shared__ uint32_t most_runs;
    if (threadIdx.x == 0) most_runs = 0;
    uint32_t my_most = 0;
    uint64_t carry_vals = 0, carry_runs = 0, carry_key = 0;
    for (uint32_t base = 0; base < n_blocks; base += SCAN_THREADS) {
        const uint32_t b = base + threadIdx.x;
        BlockTotals bt;
        bt.n_vals = 0; bt.n_runs = 0; bt.last_len = 0; bt.bad_local_end = 0;
        bt.bad_seg = 0xFFFFFFFFu; bt.bad_status = SEG_OK; bt.max_seg_runs = 0;
        if (b < n_blocks) bt = t.blk[b];
        my_most = my_most > bt.max_seg_runs ? my_most : bt.max_seg_runs;
        uint64_t iv = bt.n_vals, ir = bt.n_runs;
        uint64_t ik = bt.n_runs ? (((uint64_t)b + 1) << 10) | bt.last_len : 0;
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t uv = shfl_up64(iv, d), ur = shfl_up64(ir, d), uk = shfl_up64(ik, d);
            if (lane >= d) {
                iv += uv;
                ir += ur;
                ik = ik > uk ? ik : uk;
            }
        }
        __syncthreads();  // the previous round's reads of w_* are done
        if (lane == 63) { w_vals[wave] = iv; w_runs[wave] = ir; w_key[wave] = ik; }
        __syncthreads();
        uint64_t pv = 0, pr = 0, pk = 0, tv = 0, tr = 0, tk = 0;
        for (int w = 0; w < SCAN_THREADS / 64; w++) {
            if (w < wave) { pv += w_vals[w]; pr += w_runs[w]; pk = pk > w_key[w] ? pk : w_key[w]; }
            tv += w_vals[w];
            tr += w_runs[w];
            tk = tk > w_key[w] ? tk : w_key[w];
        }
        uint64_t incl_key = ik > pk ? ik : pk;
        incl_key = incl_key > carry_key ? incl_key : carry_key;
        uint64_t excl_key = shfl_up64(incl_key, 1);
        if (lane == 0) excl_key = pk > carry_key ? pk : carry_key;
        if (b < n_blocks) {
            BlockOffset o;
            o.val_off = carry_vals + pv + iv - bt.n_vals;
            o.run_off = (uint32_t)(carry_runs + pr + ir - bt.n_runs);
            o.prev_len = (uint32_t)(excl_key & 1023);
            t.boff[b] = o;
            // a stopped walk matters only if values past it are still wanted
            if (bt.bad_seg != 0xFFFFFFFFu && o.val_off + bt.bad_local_end < num_values)
                atomicMin(&first_bad, b);
        }
        carry_vals += tv;
        carry_runs += tr;
        carry_key = carry_key > tk ? carry_key : tk;
    }
    if (my_most) atomicMax(&most_runs, my_most);
    __syncthreads();
    if (threadIdx.x == 0) {
        ScanTotals tt;
        tt.total_vals = carry_vals;
        tt.total_runs = carry_runs;
        tt.error = ORCD_OK;
        tt.error_seg = 0xFFFFFFFFu;
        tt.max_seg_runs = most_runs;
        tt.pad = 0;
        if (first_bad != 0xFFFFFFFFu) {
            const BlockTotals bt = t.blk[first_bad];
            tt.error = bt.bad_status == SEG_TOO_LONG ? ORCD_UNSUPPORTED : ORCD_OUT_OF_SPEC;
            tt.error_seg = bt.bad_seg;
        } else if (carry_vals < num_values) {
            tt.error = ORCD_OUT_OF_SPEC;  // the stream ends before num_values values
        }
        *t.totals = tt;
        if (t.host_totals) {
            volatile ScanTotals *h = t.host_totals;
            h->total_vals = tt.total_vals;
            h->total_runs = tt.total_runs;
            h->error = tt.error;
            h->error_seg = tt.error_seg;
            h->max_seg_runs = tt.max_seg_runs;
            __threadfence_system();
            h->pad = t.host_stamp;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K3: value expansion -- one wavefront per group (the runs that start inside one aligned window
// of 512 output values).  Lane i of step j owns output values 128*j + 2*i and 128*j + 2*i + 1 of
// the window, so every store instruction writes 1 KiB of consecutive, 16-byte aligned output.
//
// The packed bytes a window needs are staged into a wavefront-private LDS slice with 16-byte
// loads and byte-swapped per dword on the way in, so that the big-endian bit stream reads as
// plain 32-bit words.  A group that consists of a single run -- the bulk of any large column,
// whose runs hold 512 values -- takes a path where every run field is wave-uniform.
// ---------------------------------------------------------------------------------------------

constexpr int VAL_WAVES = 4;                   // wavefronts per workgroup
constexpr uint32_t STAGE_BYTES = 4096 + 128;   // LDS slice per wavefront: any single run fits
constexpr uint32_t STAGE_WORDS = STAGE_BYTES / 4 + 4;  // + padding for the 3-word extraction

// What the expansion kernel does with the decoded values.
constexpr int MODE_VALUES = 0;   // out[i] = value i
constexpr int MODE_SUMS = 1;     // one saturating sum per group, nothing per value
constexpr int MODE_OFFSETS = 2;  // out[i] = sum of the values before i (string offsets)

struct Wave {
    const uint8_t *s;      // stream
    uint64_t len;          // its length: no load of the kernel may start at or after s + len
    uint32_t *lds;         // this wavefront's staging slice
    uint32_t *marks;       // 16 words of LDS of its own: one bit per value of a window
    uint32_t *patch_tab;   // PATCH_TAB_WORDS words of LDS of its own, or null (see stage_patches)
    int64_t *out;
    uint64_t num_values;
    int lane;
    int is_signed;
    bool out_aligned;      // out is 16-byte aligned
    bool stream_once = false;  // stage_bytes loads the stream with the nontemporal hint (see launch_rlev2_selfplan_values)
    // MODE_SUMS: this lane's share of the group's sum.  MODE_OFFSETS: sum of every value before
    // the next one this wavefront emits (wave-uniform).
    mutable uint64_t acc;
    mutable uint32_t saw_patched;  // MODE_SUMS: a PatchedBase run was met (its patches are not summed)
};
template <int M, bool S = false>
struct WaveOf : Wave {
    static constexpr int MODE = M;
    static constexpr bool SHORT = S;  // the kernel variant for columns of short runs
};

// Copies stream bytes [b0, b1) into the LDS slice; returns the stream offset that LDS byte 0
// stands for (up to 15 bytes before b0; it may wrap below zero, only differences are used).
// Reads whole aligned 16-byte chunks: each holds at least one byte of [b0, b1).
template <bool ONE_BY_ONE = false>
__device__ __forceinline__ uint64_t stage_bytes(const Wave &wv, uint64_t b0, uint64_t b1) {
    // Belt and braces: the header scan has validated every run against the stream length, but a
    // range that is out of the stream or larger than the slice is never loaded, whatever the
    // tables say.
    if (b1 > wv.len) b1 = wv.len;
    if (b0 >= b1 || b1 - b0 > STAGE_BYTES) return b0;
    const uint32_t skew = (uint32_t)((uintptr_t)(wv.s + b0) & 15);
    const uint4 *src = (const uint4 *)(wv.s + b0 - skew);
    const uint32_t n16 = (uint32_t)((b1 - b0 + skew + 15) >> 4);
    auto put = [&](uint32_t c, uint4 v) {
        v.x = __builtin_bswap32(v.x);
        v.y = __builtin_bswap32(v.y);
        v.z = __builtin_bswap32(v.z);
        v.w = __builtin_bswap32(v.w);
        ((uint4 *)wv.lds)[c] = v;
    };
    // All loads go out before the first is waited for: a copy loop that writes each chunk to LDS
    // before it asks for the next pays a memory round trip per 1 KiB, and a run of 512 values of 17
    // bits is 1090 bytes.  Three chunks per lane cover 3 KiB; longer ranges (fields wider than 48
    // bits) take the rest in a loop.
    // (ONE_BY_ONE: the plain loop, for the kernel variant whose registers are better spent elsewhere
    // and whose runs rarely reach 1 KiB)
    if (ONE_BY_ONE) {
        for (uint32_t c = wv.lane; c < n16; c += 64) put(c, src[c]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return b0 - skew;
    }
    const uint32_t c0 = (uint32_t)wv.lane, c1 = c0 + 64, c2 = c0 + 128;
    uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    auto nt = [&](uint32_t c) {
        const u32x4 t = __builtin_nontemporal_load((const u32x4 *)(src + c));
        return make_uint4(t.x, t.y, t.z, t.w);
    };
    if (wv.stream_once) {  // (wave-uniform)
        if (c0 < n16) v0 = nt(c0);
        if (n16 > 64) {
            if (c1 < n16) v1 = nt(c1);
            if (n16 > 128 && c2 < n16) v2 = nt(c2);
        }
    } else {
        if (c0 < n16) v0 = src[c0];
        if (n16 > 64) {
            if (c1 < n16) v1 = src[c1];
            if (n16 > 128 && c2 < n16) v2 = src[c2];
        }
    }
    if (c0 < n16) put(c0, v0);
    if (n16 > 64) {
        if (c1 < n16) put(c1, v1);
        if (n16 > 128 && c2 < n16) put(c2, v2);
        for (uint32_t c = c0 + 192; c < n16; c += 64) put(c, src[c]);
    }
    // LDS operations of one wavefront execute in order; this only keeps the compiler from
    // moving the reads below above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return b0 - skew;
}

// Field of w bits (1..32) at bit `bit` of the staged (byte-swapped) words.
__device__ __forceinline__ uint32_t lds_bits32(const uint32_t *lds, uint32_t bit, uint32_t w) {
    const uint32_t d = bit >> 5, sh = bit & 31;
    const uint64_t x = ((uint64_t)lds[d] << 32) | lds[d + 1];
    return (uint32_t)((x << sh) >> (64 - w));
}

// Field of w bits (1..64).
__device__ __forceinline__ uint64_t lds_bits64(const uint32_t *lds, uint32_t bit, uint32_t w) {
    const uint32_t d = bit >> 5, sh = bit & 31;
    const uint64_t x = ((uint64_t)lds[d] << 32) | lds[d + 1];
    uint64_t v = x << sh;
    if (sh) v |= (uint64_t)lds[d + 2] >> (32 - sh);
    return v >> (64 - w);
}

// Field of w bits (1..64) at bit `bitpos` of the big-endian bit stream starting at `words`, read
// straight from global memory.  Only words that hold wanted bits are touched: one past them may
// be unmapped.
__device__ __forceinline__ uint64_t global_bits(const uint32_t *words, uint64_t bitpos, uint32_t w) {
    const uint64_t d = bitpos >> 5;
    const uint32_t sh = (uint32_t)bitpos & 31;
    const uint32_t need = sh + w;
    const uint32_t w0 = __builtin_bswap32(words[d]);
    const uint32_t w1 = need > 32 ? __builtin_bswap32(words[d + 1]) : 0;
    const uint32_t w2 = need > 64 ? __builtin_bswap32(words[d + 2]) : 0;
    const uint64_t hi = ((uint64_t)w0 << 32) | w1;
    uint64_t v = hi << sh;
    if (sh) v |= (uint64_t)w2 >> (32 - sh);
    return v >> (64 - w);
}

// Bytes of payload that values [first, last) of a run occupy, as an offset range from payload_off.
__device__ __forceinline__ void payload_range(uint32_t kind, uint32_t w, uint32_t first,
                                              uint32_t last, uint64_t &b0, uint64_t &b1) {
    if (kind == KIND_DELTA) {  // the packed deltas start at the run's third value
        first = first > 2 ? first - 2 : 0;
        last = last > 2 ? last - 2 : 0;
    }
    b0 = ((uint64_t)first * w) >> 3;
    b1 = ((uint64_t)last * w + 7) >> 3;
}

// The decoded column is written once and not read back by this kernel: nontemporal stores keep
// it from displacing the stream bytes and run tables in L2.
__device__ __forceinline__ void stream_store2(int64_t *dst, uint64_t v0, uint64_t v1) {
    u64x2 v;
    v.x = v0;
    v.y = v1;
    __builtin_nontemporal_store(v, (u64x2 *)dst);
}
__device__ __forceinline__ void stream_store1(int64_t *dst, uint64_t v) {
    __builtin_nontemporal_store((int64_t)v, dst);
}

__device__ __forceinline__ uint64_t sat_add(uint64_t a, uint64_t b) {
    const uint64_t t = a + b;
    return t < a ? ~0ull : t;
}

__device__ __forceinline__ uint64_t wave_scan_dpp(uint64_t v);

// Values -> what is stored in their place.  Every lane of the wavefront calls this for every
// step, in ascending order of the values, with a0 / a1 false for values that are not this
// wavefront's; MODE_OFFSETS relies on both.  Returns false when nothing is to be stored.
template <class W>
__device__ __forceinline__ bool transform_pair(const W &wv, uint64_t &v0, uint64_t &v1, bool a0,
                                               bool a1) {
    if (W::MODE == MODE_SUMS) {
        wv.acc = sat_add(sat_add(wv.acc, a0 ? v0 : 0), a1 ? v1 : 0);
        return false;
    }
    if (W::MODE == MODE_OFFSETS) {
        // the totals were checked against the size of the data before this pass: no sum wraps
        const uint64_t x = a0 ? v0 : 0, y = a1 ? v1 : 0;
        const uint64_t incl = wave_scan_dpp(x + y);
        v0 = wv.acc + incl - (x + y);
        v1 = v0 + x;
        wv.acc += readlane64(incl, 63);
    }
    return true;
}

// Two adjacent output values of one lane.
template <class W>
__device__ __forceinline__ void store_pair(const W &wv, uint64_t e0, uint64_t v0, uint64_t v1,
                                           bool a0, bool a1) {
    if (!transform_pair(wv, v0, v1, a0, a1)) return;
    int64_t *dst = wv.out + e0;
    if (a0 && a1 && wv.out_aligned) {
        stream_store2(dst, v0, v1);
    } else {
        if (a0) stream_store1(dst, v0);
        if (a1) stream_store1(dst + 1, v1);
    }
}

// Inclusive prefix sum over the 64 lanes.
__device__ __forceinline__ uint64_t wave_scan_u64(uint64_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t up = shfl_up64(v, d);
        if (lane >= d) v += up;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(v, d);
        if (lane >= d) v += up;
    }
    return v;
}

// PatchedBase: the patch list of one run (all fields wave-uniform), lane j taking entry j.
// Returns whether this lane's entry patches a value of the run, and which (posn) with what.
template <class W>
__device__ __forceinline__ bool read_patch_list(const W &wv, uint64_t desc, uint64_t aux, uint32_t &posn,
                                                uint64_t &patch) {
    const uint32_t w = desc_width(desc), rlen = desc_len(desc);
    const uint32_t pw = pb_patch_width(aux), pll = pb_list_len(aux), cfb = pb_entry_bits(aux);
    const uint8_t *p = wv.s + desc_payload(desc);
    const uint32_t sk = (uint32_t)((uintptr_t)p & 3);
    const uint32_t *words = (const uint32_t *)(p - sk);
    const uint64_t list_bit = (uint64_t)sk * 8 + (((uint64_t)w * rlen + 7) >> 3) * 8;
    posn = 0;
    patch = 0;
    // never read past the stream, whatever the table says (the header scan has checked this)
    if (desc_payload(desc) + (((uint64_t)w * rlen + 7) >> 3) + (((uint64_t)cfb * pll + 7) >> 3) > wv.len)
        return false;
    uint64_t gap = 0;
    if ((uint32_t)wv.lane < pll) {
        const uint64_t entry = global_bits(words, list_bit + (uint64_t)wv.lane * cfb, cfb);
        gap = pw >= 64 ? 0 : entry >> pw;
        patch = pw >= 64 ? entry : entry & ((1ull << pw) - 1);
    }
    posn = wave_scan_u32((uint32_t)gap, wv.lane);  // position = sum of the gaps
    const bool filler = gap == 255 && patch == 0;   // only extends the gap
    // A zero gap after the first entry points back at a position the sequential readers have
    // already passed: they apply neither that entry nor any later one.
    const bool prev_filler = __shfl_up((uint32_t)filler, 1) != 0;  // a filler moves the position on
    const uint64_t stuck =
        __ballot((uint32_t)wv.lane < pll && wv.lane >= 1 && gap == 0 && !prev_filler);
    const bool reachable = (stuck & ((2ull << wv.lane) - 1)) == 0;
    return (uint32_t)wv.lane < pll && !filler && reachable && posn < rlen;
}

// PatchedBase: overwrites the patched positions of one run after its unpatched values have been
// written.
template <class W>
__device__ __forceinline__ void apply_patches(const W &wv, uint64_t desc, uint64_t base,
                                              uint64_t aux, uint64_t run_out) {
    if (W::MODE != MODE_VALUES) {  // sums and offsets of patched runs go through MODE_VALUES
        wv.saw_patched = 1;
        return;
    }
    uint32_t posn;
    uint64_t patch;
    if (read_patch_list(wv, desc, aux, posn, patch) && run_out + posn < wv.num_values) {
        const uint32_t w = desc_width(desc);
        const uint8_t *p = wv.s + desc_payload(desc);
        const uint32_t sk = (uint32_t)((uintptr_t)p & 3);
        uint64_t m = global_bits((const uint32_t *)(p - sk), (uint64_t)sk * 8 + (uint64_t)posn * w, w);
        if (w < 64) m |= patch << w;
        stream_store1(wv.out + run_out + posn, base + m);
    }
}

// PatchedBase, ahead of the expansion: the patches of a run that has its windows to itself go into
// the wavefront's LDS table -- byte p: 1 + the list entry that patches the run's value p, or 0;
// behind the 512 bytes the entries' patches, 8 bytes each -- and the expansion ORs them into the
// values before they are written, once (written afterwards, a patched value costs a wait for every
// store of the run, two dependent reads and a partial rewrite of a line just written).
constexpr uint32_t PATCH_TAB_WORDS = WINDOW / 4 + 2 * 32;

template <class W>
__device__ __forceinline__ void stage_patches(const W &wv, uint64_t desc, uint64_t aux) {
    uint32_t posn;
    uint64_t patch;
    const bool mine = read_patch_list(wv, desc, aux, posn, patch);
    wv.patch_tab[wv.lane] = 0;
    wv.patch_tab[64 + wv.lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (mine) {
        ((uint8_t *)wv.patch_tab)[posn] = (uint8_t)(wv.lane + 1);
        ((uint64_t *)(wv.patch_tab + WINDOW / 4))[wv.lane] = patch;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- a group that is one run: every run field is wave-uniform -----------------------------------
//
// FULL: the window is entirely this run's and `out` is 16-byte aligned, so there is no lane
// masking and every store is a 16-byte one.  lo_rel / hi_rel bound the active values inside the
// window otherwise.

template <bool FULL, class W>
__device__ __forceinline__ void store_window_pair(const W &wv, int64_t *wout, uint32_t i_rel,
                                                  uint64_t v0, uint64_t v1, uint32_t lo_rel,
                                                  uint32_t hi_rel) {
    if (FULL) {
        if (!transform_pair(wv, v0, v1, true, true)) return;
        stream_store2(wout + i_rel, v0, v1);
    } else {
        const bool a0 = i_rel >= lo_rel && i_rel < hi_rel;
        const bool a1 = i_rel + 1 >= lo_rel && i_rel + 1 < hi_rel;
        if (!transform_pair(wv, v0, v1, a0, a1)) return;
        if (a0 && a1 && wv.out_aligned) {
            stream_store2(wout + i_rel, v0, v1);
        } else {
            if (a0) stream_store1(wout + i_rel, v0);
            if (a1) stream_store1(wout + i_rel + 1, v1);
        }
    }
}

// Direct and (unpatched) PatchedBase values of one window.  bit_lane: LDS bit position of the
// packed field of this lane's first value of step 0.
// PATCHED: the run is a PatchedBase run whose patches are staged (stage_patches); the lane's
// first value of step 0 is value idx0 of the run (an even number; negative before the run).
template <bool FULL, bool WIDE, bool PATCHED = false, class W>
__device__ __forceinline__ void direct_window(const W &wv, int64_t *wout, uint32_t w, bool zz,
                                              uint64_t add, uint32_t bit_lane, uint32_t lo_rel,
                                              uint32_t hi_rel, int32_t idx0 = 0) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i_rel = 128u * j + 2u * wv.lane;
        uint64_t or0 = 0, or1 = 0;
        if (PATCHED) {
            const int32_t at = idx0 + 128 * j;  // the table covers the run's values 0..511
            const uint32_t who = (at >= 0 && at < (int32_t)WINDOW) ? ((const uint16_t *)wv.patch_tab)[at >> 1] : 0u;
            if (who) {
                const uint64_t *pv = (const uint64_t *)(wv.patch_tab + WINDOW / 4);
                if (who & 0xFFu) or0 = pv[(who & 0xFFu) - 1];
                if (who >> 8) or1 = pv[(who >> 8) - 1];
                if (w < 64) {
                    or0 <<= w;
                    or1 <<= w;
                } else {
                    or0 = or1 = 0;
                }
            }
        }
        uint32_t bA = bit_lane + (128u * j) * w, bB = bA + w;
        if (!FULL) {  // keep the LDS reads of inactive lanes inside the slice
            if (!(i_rel >= lo_rel && i_rel < hi_rel)) bA = 0;
            if (!(i_rel + 1 >= lo_rel && i_rel + 1 < hi_rel)) bB = 0;
        }
        uint64_t v0, v1;
        if (!WIDE) {
            const uint32_t m0 = lds_bits32(wv.lds, bA, w), m1 = lds_bits32(wv.lds, bB, w);
            if (zz) {  // zigzag in 32 bits, then sign-extend
                v0 = (uint64_t)(int64_t)(int32_t)((m0 >> 1) ^ (0u - (m0 & 1u)));
                v1 = (uint64_t)(int64_t)(int32_t)((m1 >> 1) ^ (0u - (m1 & 1u)));
            } else {
                v0 = add + (m0 | or0);
                v1 = add + (m1 | or1);
            }
        } else {
            const uint64_t m0 = lds_bits64(wv.lds, bA, w), m1 = lds_bits64(wv.lds, bB, w);
            v0 = zz ? zigzag(m0) : add + (m0 | or0);
            v1 = zz ? zigzag(m1) : add + (m1 | or1);
        }
        store_window_pair<FULL>(wv, wout, i_rel, v0, v1, lo_rel, hi_rel);
    }
}

// Inclusive prefix sums over the 64 lanes with DPP row shifts and row broadcasts (no LDS traffic):
// shifts by 1, 2, 4, 8 scan each row of 16 lanes, then lane 15 of rows 0 and 2 is added to rows 1
// and 3, then lane 31 to rows 2 and 3.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_scan_dpp(uint32_t v) {
    v += dpp_or_zero<0x111, 0xF>(v);
    v += dpp_or_zero<0x112, 0xF>(v);
    v += dpp_or_zero<0x114, 0xF>(v);
    v += dpp_or_zero<0x118, 0xF>(v);
    v += dpp_or_zero<0x142, 0xA>(v);
    v += dpp_or_zero<0x143, 0xC>(v);
    return v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_or_zero64(uint64_t v) {
    return ((uint64_t)dpp_or_zero<CTRL, ROW_MASK>((uint32_t)(v >> 32)) << 32) |
           dpp_or_zero<CTRL, ROW_MASK>((uint32_t)v);
}
__device__ __forceinline__ uint64_t wave_scan_dpp(uint64_t v) {
    v += dpp_or_zero64<0x111, 0xF>(v);
    v += dpp_or_zero64<0x112, 0xF>(v);
    v += dpp_or_zero64<0x114, 0xF>(v);
    v += dpp_or_zero64<0x118, 0xF>(v);
    v += dpp_or_zero64<0x142, 0xA>(v);
    v += dpp_or_zero64<0x143, 0xC>(v);
    return v;
}

// Delta values of one window: value[i] = base (+ delta_base for i >= 1) +/- the sum of the packed
// deltas up to i.  Only the packed magnitudes are scanned; SUM is uint32_t when 510 of them cannot
// overflow 32 bits (width <= 23).  idx0: index in the run of this lane's first value of step 0
// (negative before the run).  carry: sum of the run's magnitudes before the current step.
template <bool FULL, bool WIDE, typename SUM, class W>
__device__ __forceinline__ void delta_window(const W &wv, int64_t *wout, uint32_t w, bool neg,
                                             uint64_t base, uint64_t delta_base, uint32_t bit_lane,
                                             int32_t idx0, uint32_t lo_rel, uint32_t hi_rel,
                                             uint64_t &carry) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i_rel = 128u * j + 2u * wv.lane;
        const int32_t i0 = idx0 + 128 * j, i1 = i0 + 1;
        const bool a0 = FULL || (i_rel >= lo_rel && i_rel < hi_rel);
        const bool a1 = FULL || (i_rel + 1 >= lo_rel && i_rel + 1 < hi_rel);
        const bool p0 = a0 && i0 >= 2, p1 = a1 && i1 >= 2;  // values that have a packed delta
        const uint32_t bA = p0 ? bit_lane + (128u * j) * w : 0u;
        const uint32_t bB = p1 ? bit_lane + (128u * j + 1) * w : 0u;
        SUM m0, m1;
        if (!WIDE) {
            m0 = (SUM)lds_bits32(wv.lds, bA, w);
            m1 = (SUM)lds_bits32(wv.lds, bB, w);
        } else {
            m0 = (SUM)lds_bits64(wv.lds, bA, w);
            m1 = (SUM)lds_bits64(wv.lds, bB, w);
        }
        if (!p0) m0 = 0;
        if (!p1) m1 = 0;
        const SUM incl = wave_scan_dpp((SUM)(m0 + m1));  // magnitudes of this step up to value i1
        const uint64_t s1 = carry + incl, s0 = s1 - m1;
        const uint64_t f0 = i0 >= 1 ? base + delta_base : base, f1 = i1 >= 1 ? base + delta_base : base;
        const uint64_t v0 = neg ? f0 - s0 : f0 + s0, v1 = neg ? f1 - s1 : f1 + s1;
        carry = readlane64(s1, 63);
        store_window_pair<FULL>(wv, wout, i_rel, v0, v1, lo_rel, hi_rel);
    }
}

// The same for a window that is all one run of 512 values (the bulk of a sorted column): the run
// starts with the window, so lane i of step j owns the run's values 128 j + 2 i and + 1, only
// values 0 and 1 (lane 0 of step 0) lack a packed delta, and nothing has to be masked.  What the
// general form computes per value is computed per step here: the running value -- first value plus
// or minus the deltas of the steps before -- is a scalar, a lane adds its step's inclusive sum to
// it for its second value and takes its second delta off again for the first.
// PAIRED: fields of at most 16 bits; a lane's two fields lie side by side and come out of one
// 64-bit word.  bit0: LDS bit position of the run's packed field 0.
template <bool PAIRED, bool WIDE, typename SUM, bool NEG, class W>
__device__ __forceinline__ void delta_window_full(const W &wv, int64_t *wout, uint32_t w, uint64_t base,
                                                  uint64_t delta_base, uint32_t bit0) {
    const uint32_t lane = (uint32_t)wv.lane;
    uint64_t run = base + delta_base;  // (wave-uniform) value 1, then the last value of the step before
    const uint32_t bit = bit0 + (2u * lane - 2u) * w;  // lane 0, step 0: no field, see below
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t bA = bit + (128u * j) * w;
        if (j == 0 && lane == 0) bA = bit0;  // (any place inside the slice)
        SUM m0, m1;
        if (PAIRED) {
            const uint32_t d = bA >> 5, sh = bA & 31;
            const uint64_t x = ((uint64_t)wv.lds[d] << 32) | wv.lds[d + 1];
            const uint32_t top = (uint32_t)((x << sh) >> 32);
            m0 = (SUM)(top >> (32 - w));
            m1 = (SUM)((top << w) >> (32 - w));
        } else if (!WIDE) {
            m0 = (SUM)lds_bits32(wv.lds, bA, w);
            m1 = (SUM)lds_bits32(wv.lds, bA + w, w);
        } else {
            m0 = (SUM)lds_bits64(wv.lds, bA, w);
            m1 = (SUM)lds_bits64(wv.lds, bA + w, w);
        }
        if (j == 0 && lane == 0) {
            m0 = 0;
            m1 = 0;
        }
        const SUM incl = wave_scan_dpp((SUM)(m0 + m1));
        uint64_t v1 = NEG ? run - incl : run + incl;
        uint64_t v0 = NEG ? v1 + m1 : v1 - m1;
        if (j == 0 && lane == 0) v0 = base;
        SUM total;
        if (sizeof(SUM) == 4) total = (SUM)readlane32((uint32_t)incl, 63);
        else total = (SUM)readlane64((uint64_t)incl, 63);
        run = NEG ? run - total : run + total;
        store_window_pair<true>(wv, wout, 128u * j + 2u * lane, v0, v1, 0, WINDOW);
    }
}

// ShortRepeat (step 0) and Delta without packed deltas: base + i * step.
template <bool FULL, class W>
__device__ __forceinline__ void linear_window(const W &wv, int64_t *wout, uint64_t base,
                                              uint64_t step, int32_t idx0, uint32_t lo_rel,
                                              uint32_t hi_rel) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i_rel = 128u * j + 2u * wv.lane;
        const uint64_t i0 = (uint64_t)(int64_t)(idx0 + 128 * j);
        store_window_pair<FULL>(wv, wout, i_rel, base + i0 * step, base + (i0 + 1) * step, lo_rel,
                                hi_rel);
    }
}

template <class W>
__device__ __forceinline__ void decode_single_run(const W &wv, uint64_t desc, uint64_t run_out,
                                                  uint64_t base, uint64_t aux) {
    const uint32_t kind = desc_kind(desc), w = desc_width(desc), rlen = desc_len(desc);
    uint64_t run_end = run_out + rlen;
    if (run_end > wv.num_values) run_end = wv.num_values;
    if (run_out >= run_end) return;  // the run starts at or after the last wanted value
    const bool packed_kind = w != 0 && kind != KIND_SHORT_REPEAT;
    // a PatchedBase run whose first value is at an even place of its window: patches go in ahead
    const bool patch_ahead = kind == KIND_PATCHED_BASE && W::MODE == MODE_VALUES && wv.patch_tab != nullptr &&
                             ((run_out & 1) == 0);
    if (patch_ahead) stage_patches(wv, desc, aux);
    uint64_t carry = 0;  // Delta: sum of the run's packed deltas before the current step
    // The packed fields of all the run's wanted values are staged once (a run fits the slice): a
    // run that starts inside a window lies in two, and a staging per window was a second trip to
    // memory for bytes that follow right behind the first's.
    uint32_t bit0 = 0;  // bit position in LDS of packed field 0 of the run (modulo 2^32)
    const bool two_windows = ((run_end - 1) >> WINDOW_SHIFT) != (run_out >> WINDOW_SHIFT);
    if (packed_kind && two_windows) {
        uint64_t b0, b1;
        payload_range(kind, w, 0u, (uint32_t)(run_end - run_out), b0, b1);
        if (b1 > b0) {
            const uint64_t origin = stage_bytes<W::SHORT>(wv, desc_payload(desc) + b0, desc_payload(desc) + b1);
            bit0 = (uint32_t)((desc_payload(desc) - origin) << 3);
        }
    }
    for (uint64_t wbase = (run_out >> WINDOW_SHIFT) << WINDOW_SHIFT; wbase < run_end; wbase += WINDOW) {
        // active values of this window, relative to its start
        const uint32_t lo_rel = wbase > run_out ? 0u : (uint32_t)(run_out - wbase);
        const uint32_t hi_rel = (wbase + WINDOW) < run_end ? WINDOW : (uint32_t)(run_end - wbase);
        const bool full = lo_rel == 0 && hi_rel == WINDOW && wv.out_aligned;
        // index in the run of the window's first value (negative before the run)
        const int32_t win_idx = (int32_t)(int64_t)(wbase - run_out);
        if (packed_kind && !two_windows) {
            uint64_t b0, b1;
            payload_range(kind, w, (uint32_t)(win_idx + (int32_t)lo_rel),
                          (uint32_t)(win_idx + (int32_t)hi_rel), b0, b1);
            if (b1 > b0) {
                const uint64_t origin = stage_bytes<W::SHORT>(wv, desc_payload(desc) + b0, desc_payload(desc) + b1);
                bit0 = (uint32_t)((desc_payload(desc) - origin) << 3);
            }
        }
        int64_t *wout = wv.out + wbase;
        const int32_t idx0 = win_idx + 2 * wv.lane;
        if (kind == KIND_DIRECT || kind == KIND_PATCHED_BASE) {
            const uint64_t add = kind == KIND_PATCHED_BASE ? base : 0;
            const bool zz = kind == KIND_DIRECT && wv.is_signed;
            const uint32_t bit_lane = bit0 + (uint32_t)idx0 * w;
            if (patch_ahead) {
                if (w <= 32) {
                    if (full) direct_window<true, false, true>(wv, wout, w, zz, add, bit_lane, lo_rel, hi_rel, idx0);
                    else direct_window<false, false, true>(wv, wout, w, zz, add, bit_lane, lo_rel, hi_rel, idx0);
                } else {
                    if (full) direct_window<true, true, true>(wv, wout, w, zz, add, bit_lane, lo_rel, hi_rel, idx0);
                    else direct_window<false, true, true>(wv, wout, w, zz, add, bit_lane, lo_rel, hi_rel, idx0);
                }
            } else if (w <= 32) {
                if (full) direct_window<true, false>(wv, wout, w, zz, add, bit_lane, lo_rel, hi_rel);
                else direct_window<false, false>(wv, wout, w, zz, add, bit_lane, lo_rel, hi_rel);
            } else {
                if (full) direct_window<true, true>(wv, wout, w, zz, add, bit_lane, lo_rel, hi_rel);
                else direct_window<false, true>(wv, wout, w, zz, add, bit_lane, lo_rel, hi_rel);
            }
        } else if (kind == KIND_DELTA && w != 0) {
            const bool neg = desc_delta_neg(desc) != 0;
            const uint32_t bit_lane = bit0 + (uint32_t)(idx0 - 2) * w;
            if (!W::SHORT && full && rlen == WINDOW) {  // (a full window of a run of 512 values: the run starts with it)
                if (w <= 16) {
                    if (neg) delta_window_full<true, false, uint32_t, true>(wv, wout, w, base, aux, bit0);
                    else delta_window_full<true, false, uint32_t, false>(wv, wout, w, base, aux, bit0);
                } else if (w <= 23) {
                    if (neg) delta_window_full<false, false, uint32_t, true>(wv, wout, w, base, aux, bit0);
                    else delta_window_full<false, false, uint32_t, false>(wv, wout, w, base, aux, bit0);
                } else if (w <= 32) {
                    if (neg) delta_window_full<false, false, uint64_t, true>(wv, wout, w, base, aux, bit0);
                    else delta_window_full<false, false, uint64_t, false>(wv, wout, w, base, aux, bit0);
                } else {
                    if (neg) delta_window_full<false, true, uint64_t, true>(wv, wout, w, base, aux, bit0);
                    else delta_window_full<false, true, uint64_t, false>(wv, wout, w, base, aux, bit0);
                }
            } else if (w <= 23) {
                if (full) delta_window<true, false, uint32_t>(wv, wout, w, neg, base, aux, bit_lane, idx0, lo_rel, hi_rel, carry);
                else delta_window<false, false, uint32_t>(wv, wout, w, neg, base, aux, bit_lane, idx0, lo_rel, hi_rel, carry);
            } else if (w <= 32) {
                if (full) delta_window<true, false, uint64_t>(wv, wout, w, neg, base, aux, bit_lane, idx0, lo_rel, hi_rel, carry);
                else delta_window<false, false, uint64_t>(wv, wout, w, neg, base, aux, bit_lane, idx0, lo_rel, hi_rel, carry);
            } else {
                if (full) delta_window<true, true, uint64_t>(wv, wout, w, neg, base, aux, bit_lane, idx0, lo_rel, hi_rel, carry);
                else delta_window<false, true, uint64_t>(wv, wout, w, neg, base, aux, bit_lane, idx0, lo_rel, hi_rel, carry);
            }
        } else {
            const uint64_t step = kind == KIND_DELTA ? aux : 0;
            if (full) linear_window<true>(wv, wout, base, step, idx0, lo_rel, hi_rel);
            else linear_window<false>(wv, wout, base, step, idx0, lo_rel, hi_rel);
        }
    }
    if (kind == KIND_PATCHED_BASE && !patch_ahead) {
        __builtin_amdgcn_s_waitcnt(0);  // the unpatched values are written
        apply_patches(wv, desc, base, aux, run_out);
    }
}

// ---- any group: up to 64 runs at a time, one per lane ------------------------------------------

struct RunFields {
    uint64_t desc, base, aux;
    uint32_t rel_out;  // out_off relative to the batch's first value
};

// ---- batches of runs that carry no running sum: a lane per eight values ---------------------------
//
// ShortRepeat, Direct and Delta runs without packed deltas: each value follows from its run's
// fields and its index in the run.  Every run of a batch starts inside the wavefront's own window,
// which is decoded with lane i computing the eight consecutive values 8 i .. 8 i + 7.  Which run a
// value belongs to is counted, not searched: each run sets the bit of its first value in a 512-bit
// mask in LDS, lane i reads the mask's byte i, a prefix sum of the bytes' bit counts over the
// lanes gives the number of runs that start before the lane's values, and the bits of the byte
// below and at a value give the rest.  The run's fields then come in one 32-bit word from the lane
// that holds the run; the value of a ShortRepeat run is read from the staged bytes like a packed
// field (of which it is one).  The eight values go through the wavefront's LDS slice -- free again
// once the packed bytes have been read -- into the layout of the stores (lane i of step j: values
// 128 j + 2 i, + 1).
//   my_ws:  first value of lane k's run relative to the window (>= 512: the lane holds no run)
//   my_rec: my_ws | byte in the slice of the run's packed field 0 << 9 | width << 22 | kind << 29
//   lo_w / hi_w: the part of the window that this batch holds, relative to the window
template <bool WIDE, bool LINEAR, bool SIGNED, class W>
__device__ __forceinline__ void expand_window_by_marks(const W &wv, uint64_t wbase, uint32_t lo_w, uint32_t hi_w,
                                                       uint32_t my_ws, uint32_t my_rec, uint64_t my_base,
                                                       uint64_t my_aux) {
    const int lane = wv.lane;
    if (lane < 16) wv.marks[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (my_ws < WINDOW) atomicOr(&wv.marks[my_ws >> 5], 1u << (my_ws & 31));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t starts = ((const uint8_t *)wv.marks)[lane];  // bit t: a run starts at value 8 lane + t
    const uint32_t n_starts = (uint32_t)__builtin_popcount(starts);
    // index of the run that holds the value in front of this lane's first (-1: none does)
    const uint32_t k_before = wave_scan_dpp(n_starts) - n_starts - 1u;
    uint64_t v[8];
    // Fields of at most 32 bits, no Delta runs: the word that travels is prepared once per run so
    // that a value's bit position is one multiply-add -- (bit of field 0 - start * stride + 2^14) :
    // 16 | stride in bits (the width, or 0 for a ShortRepeat run) : 6 | 64 - width : 6.
    uint32_t my_narrow = 0;
    if (!WIDE) {
        const uint32_t w = (my_rec >> 22) & 127u, stride = (my_rec >> 29) == KIND_DIRECT ? w : 0u;
        my_narrow = ((((my_rec >> 6) & 0xFFF8u) + 16384u - __umul24(my_rec & 511u, stride)) & 0xFFFFu) |
                    (stride << 16) | ((64u - (w ? (w > 32 ? 32u : w) : 1u)) << 22);
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const uint32_t p = 8u * (uint32_t)lane + (uint32_t)t;
        const bool act = p - lo_w < hi_w - lo_w;  // lo_w <= p < hi_w
        const uint32_t k = k_before + (uint32_t)__builtin_popcount(starts & ((2u << t) - 1u));
        if (!WIDE) {
            const uint32_t nr = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(k << 2), (int)my_narrow);
            // (lanes with nothing to read look at bit 0 of the slice)
            const uint32_t bit = act ? __umul24(p, (nr >> 16) & 63u) + (nr & 0xFFFFu) - 16384u : 0u;
            const uint32_t d = bit >> 5, shr = nr >> 22;
            const uint64_t x = ((uint64_t)wv.lds[d] << 32) | wv.lds[d + 1];
            const uint32_t m = (uint32_t)(x >> (shr - (bit & 31u))) & (0xFFFFFFFFu >> (shr - 32u));
            // (a zigzag field of at most 32 bits: the sign fills the upper half)
            const uint32_t sign = 0u - (m & 1u);
            v[t] = SIGNED ? ((uint64_t)sign << 32) | ((m >> 1) ^ sign) : (uint64_t)m;
            continue;
        }
        const uint32_t rec = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(k << 2), (int)my_rec);
        const uint32_t w = (rec >> 22) & 127u, kind = rec >> 29;
        const uint32_t idx = p - (rec & 511u);
        // (lanes with nothing to read look at bit 0 of the slice)
        uint32_t bit = ((rec >> 6) & 0xFFF8u) + __umul24(kind == KIND_DIRECT ? idx : 0u, w);
        if (!act || (LINEAR && kind == KIND_DELTA)) bit = 0;
        uint64_t val;
        if (WIDE) {
            const uint64_t m = lds_bits64(wv.lds, bit, w ? w : 1u);
            val = SIGNED ? zigzag(m) : m;
        } else {
            const uint32_t m = lds_bits32(wv.lds, bit, w ? w : 1u);
            // (a zigzag field of at most 32 bits: the sign fills the upper half)
            const uint32_t sign = 0u - (m & 1u);
            val = SIGNED ? ((uint64_t)sign << 32) | ((m >> 1) ^ sign) : (uint64_t)m;
        }
        if (LINEAR) {  // Delta without packed deltas: base + index * step
            const uint64_t base = shfl64(my_base, (int)(k & 63u)), step = shfl64(my_aux, (int)(k & 63u));
            if (kind == KIND_DELTA) val = base + (uint64_t)idx * step;
        }
        v[t] = val;
    }
    // registers -> LDS -> the layout of the stores.  Pairs (2 q, 2 q + 1) of lane i go to 16-byte
    // slot 64 q + i (neighbouring lanes, neighbouring slots); lane i of step j wants the pair
    // 2 (i & 3) of lane 16 j + (i >> 2).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every read of the packed bytes is done
    u64x2 *tile = (u64x2 *)wv.lds;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        u64x2 pair;
        pair.x = v[2 * q];
        pair.y = v[2 * q + 1];
        tile[64 * q + lane] = pair;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int64_t *wout = wv.out + wbase;
    const bool full = lo_w == 0 && hi_w == WINDOW && wv.out_aligned;  // (wave-uniform)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const u64x2 pair = tile[64 * (lane & 3) + 16 * j + (lane >> 2)];
        const uint32_t p = 128u * j + 2u * (uint32_t)lane;
        const bool a0 = p - lo_w < hi_w - lo_w, a1 = p + 1 - lo_w < hi_w - lo_w;
        if (full || (a0 && a1 && wv.out_aligned)) {
            stream_store2(wout + p, pair.x, pair.y);
        } else {
            if (a0) stream_store1(wout + p, pair.x);
            if (a1) stream_store1(wout + p + 1, pair.y);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the slice may be staged into again
}

// A batch of runs without running sums, planned: everything short_expand needs, with the loads of
// the window's stream bytes already on their way (so that a wavefront can plan a second window
// before it expands the first, and the memory hops of the two coincide).
struct ShortPlan {
    uint32_t nb;                       // runs of the batch that start inside the window
    uint64_t batch_beg, batch_all, batch_end, own_end;
    uint64_t desc;                     // lane k: run k of the batch
    uint32_t rel;                      // ... its first value relative to batch_beg
    bool mine;
    uint32_t flags;                    // PLAN_* (one scalar register for the four of them)
    uint64_t b0, b1;                   // stream bytes of the window
    uint4 chunk;                       // lane c: 16-byte chunk c of them, when they are at most 64 chunks
    uint64_t t_start, t_desc;          // the batch's last run, which alone may reach past the window
};
constexpr uint32_t PLAN_ANY_LINEAR = 1;   // some run is a Delta run without packed deltas
constexpr uint32_t PLAN_WIDE = 2;         // some packed field is wider than 32 bits
constexpr uint32_t PLAN_PREFETCHED = 4;   // `chunk` holds the window's bytes
constexpr uint32_t PLAN_TAIL_STAGED = 8;  // b1 reaches to the end of the last run's bytes

constexpr int SHORT_OTHER = 0;    // not a batch of that kind (nothing done: decode_run_batch takes it)
constexpr int SHORT_NOTHING = 1;  // no run, or every run after the last wanted value
constexpr int SHORT_PLANNED = 2;

// desc: lane k's descriptor of run rb + k, loaded by the caller (`held`: the table has such a run).
template <class W>
__device__ __forceinline__ int short_plan(const W &wv, uint64_t desc, bool held, const uint64_t batch_beg,
                                          const uint64_t own_end, ShortPlan &pl) {
    const int lane = wv.lane;
    if (!held) desc = make_desc(0, KIND_SHORT_REPEAT, 0, 1, 0, 1);
    const uint32_t len = held ? desc_len(desc) : 0u;
    const uint32_t incl = wave_scan_dpp(len);
    const uint32_t rel = incl - len;
    // (the batch's first run starts inside the window: own_end - batch_beg is 1..512.  Slots of the
    // table past the window's runs may hold anything; lengths are 1..512 whatever the bits.)
    const uint32_t room = (uint32_t)(own_end - batch_beg);
    const bool mine = held && rel < room;
    pl.nb = (uint32_t)__builtin_popcountll(__ballot(mine));
    pl.batch_beg = batch_beg;
    pl.batch_all = batch_beg;
    pl.own_end = own_end;
    if (pl.nb == 0) return SHORT_NOTHING;
    const uint32_t kind = desc_kind(desc), w = desc_width(desc);
    if (__ballot(mine && !(kind == KIND_SHORT_REPEAT || kind == KIND_DIRECT || (kind == KIND_DELTA && w == 0))))
        return SHORT_OTHER;
    const int last = (int)pl.nb - 1;
    pl.batch_all = batch_beg + readlane32(incl, last);
    pl.batch_end = pl.batch_all < wv.num_values ? pl.batch_all : wv.num_values;
    if (batch_beg >= pl.batch_end) return SHORT_NOTHING;  // every run of the batch lies after the last wanted value
    pl.t_start = batch_beg + readlane32(rel, last);
    pl.t_desc = readlane64(desc, last);
    const uint64_t t_pay = desc_payload(pl.t_desc);
    const uint32_t t_len = desc_len(pl.t_desc), t_kind = desc_kind(pl.t_desc), t_w = desc_width(pl.t_desc);
    const uint64_t hi = own_end < pl.batch_end ? own_end : pl.batch_end;
    // bytes of the window: from the first run's packed bytes to the last byte that the last run's
    // values in front of `hi` take
    const uint32_t t_count = hi > pl.t_start ? ((hi - pl.t_start) < t_len ? (uint32_t)(hi - pl.t_start) : t_len) : 0u;
    uint64_t b0 = readlane64(desc_payload(desc), 0);
    uint64_t b1 = t_pay + (t_kind == KIND_DIRECT ? ((uint64_t)t_count * t_w + 7) >> 3
                                                         : (t_kind == KIND_SHORT_REPEAT ? t_w >> 3 : 0u));
    if (b1 > wv.len) b1 = wv.len;  // (never, for a table the header scan made)
    if (b1 < b0 || b1 - b0 + 15 > STAGE_BYTES) return SHORT_OTHER;
    // what the last run holds past the window follows right behind: staged along with the rest
    // when it fits, so that those values cost no memory round trip of their own
    uint32_t flags = 0;
    if (pl.batch_end > own_end && t_kind != KIND_DELTA) {
        uint64_t full = t_pay + (t_kind == KIND_DIRECT ? ((uint64_t)t_len * t_w + 7) >> 3 : t_w >> 3);
        if (full <= wv.len && full >= b1 && full - b0 + 15 <= STAGE_BYTES) {
            b1 = full;
            flags |= PLAN_TAIL_STAGED;
        }
    }
    pl.b0 = b0;
    pl.b1 = b1;
    pl.desc = desc;
    pl.rel = rel;
    pl.mine = mine;
    if (__ballot(mine && kind == KIND_DELTA)) flags |= PLAN_ANY_LINEAR;
    if (__ballot(mine && kind != KIND_DELTA && w > 32)) flags |= PLAN_WIDE;
    // the window's bytes, asked for now: whole aligned 16-byte chunks as stage_bytes reads them
    pl.chunk = make_uint4(0, 0, 0, 0);
    if (b1 > b0) {
        const uint32_t skew = (uint32_t)((uintptr_t)(wv.s + b0) & 15);
        const uint32_t n16 = (uint32_t)((b1 - b0 + skew + 15) >> 4);
        if (n16 <= 64) {
            flags |= PLAN_PREFETCHED;
            if ((uint32_t)lane < n16) pl.chunk = ((const uint4 *)(wv.s + b0 - skew))[lane];
        }
    }
    pl.flags = flags;
    return SHORT_PLANNED;
}

// Decodes what short_plan has planned (MODE_VALUES).
template <class W>
__device__ __forceinline__ void short_expand(const W &wv, const ShortPlan &pl) {
    const int lane = wv.lane;
    const uint64_t wbase = pl.own_end - WINDOW;
    const uint64_t hi = pl.own_end < pl.batch_end ? pl.own_end : pl.batch_end;
    const uint32_t kind = desc_kind(pl.desc), w = desc_width(pl.desc);
    const bool any_linear = pl.flags & PLAN_ANY_LINEAR, wide = pl.flags & PLAN_WIDE;
    const bool prefetched = pl.flags & PLAN_PREFETCHED, tail_staged = pl.flags & PLAN_TAIL_STAGED;
    const uint64_t t_pay = desc_payload(pl.t_desc);
    const uint32_t t_len = desc_len(pl.t_desc), t_kind = desc_kind(pl.t_desc), t_w = desc_width(pl.t_desc);
    // lane k: base and step of a Delta run without packed deltas (parsed here, not while planning:
    // a plan that is kept across another window's expansion should hold as little as it can)
    uint64_t base = 0, aux = 0;
    if (any_linear && pl.mine && kind == KIND_DELTA) {
        // the header scan accepted this header: the parse cannot fail, and reads what it read
        const uint64_t pos = desc_payload(pl.desc) - desc_header_bytes(pl.desc);
        ParsedRun r;
        r.base = 0;
        r.aux = 0;
        if (pos < wv.len && parse_run(load_window(wv.s, wv.len, pos), wv.s, wv.len, pos, wv.is_signed, r) == SEG_OK) {
            base = r.base;
            aux = r.aux;
        }
    }
    {
        uint64_t origin = pl.b0;
        if (prefetched) {
            const uint32_t skew = (uint32_t)((uintptr_t)(wv.s + pl.b0) & 15);
            uint4 v = pl.chunk;  // (chunks past the span are zero and land inside the slice)
            v.x = __builtin_bswap32(v.x);
            v.y = __builtin_bswap32(v.y);
            v.z = __builtin_bswap32(v.z);
            v.w = __builtin_bswap32(v.w);
            ((uint4 *)wv.lds)[lane] = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            origin = pl.b0 - skew;
        } else if (pl.b1 > pl.b0) {
            origin = stage_bytes<W::SHORT>(wv, pl.b0, pl.b1);
        }
        if (tail_staged) {
            // the last run's values from own_end on, out of the bytes just staged (before the
            // window's own expansion takes the slice for its output tile); lane i takes values
            // i, i + 64, ...
            const uint32_t first = (uint32_t)(pl.own_end - pl.t_start);
            const uint32_t end = (uint32_t)(pl.batch_end - pl.t_start);
            const uint32_t bit0 = (uint32_t)((t_pay - origin) << 3);
            for (uint32_t idx = first + (uint32_t)lane; idx < end; idx += 64) {
                const uint64_t m = lds_bits64(wv.lds, bit0 + (t_kind == KIND_DIRECT ? idx * t_w : 0u), t_w);
                stream_store1(wv.out + pl.t_start + idx, wv.is_signed ? zigzag(m) : m);
            }
        }
        const uint32_t my_ws = pl.mine ? pl.rel + (uint32_t)(pl.batch_beg - wbase) : 0xFFFFFFFFu;
        const uint32_t my_rec = (my_ws & 511u) |
                                ((((uint32_t)desc_payload(pl.desc) - (uint32_t)origin) & 8191u) << 9) |
                                (w << 22) | (kind << 29);
        const uint32_t lo_w = (uint32_t)(pl.batch_beg - wbase), hi_w = (uint32_t)(hi - wbase);
        if (any_linear) {
            if (wv.is_signed) expand_window_by_marks<true, true, true>(wv, wbase, lo_w, hi_w, my_ws, my_rec, base, aux);
            else expand_window_by_marks<true, true, false>(wv, wbase, lo_w, hi_w, my_ws, my_rec, base, aux);
        } else if (wide) {
            if (wv.is_signed) expand_window_by_marks<true, false, true>(wv, wbase, lo_w, hi_w, my_ws, my_rec, base, aux);
            else expand_window_by_marks<true, false, false>(wv, wbase, lo_w, hi_w, my_ws, my_rec, base, aux);
        } else {
            if (wv.is_signed) expand_window_by_marks<false, false, true>(wv, wbase, lo_w, hi_w, my_ws, my_rec, base, aux);
            else expand_window_by_marks<false, false, false>(wv, wbase, lo_w, hi_w, my_ws, my_rec, base, aux);
        }
    }
    // the windows behind (unless done above): what the last run holds from own_end on; lane i
    // takes values i, i + 64, ...
    const int last = (int)pl.nb - 1;
    const uint64_t t_base = readlane64(base, last), t_step = readlane64(aux, last);
    for (uint64_t wb = pl.own_end; wb < pl.batch_end && !tail_staged; wb += WINDOW) {
        const uint64_t we = (wb + WINDOW) < pl.batch_end ? wb + WINDOW : pl.batch_end;
        const uint32_t first = (uint32_t)(wb - pl.t_start), end = (uint32_t)(we - pl.t_start);  // indices in the run
        uint32_t bit0 = 0;  // bit in the slice of the run's packed field 0 (modulo 2^32)
        if (t_kind != KIND_DELTA) {
            const uint64_t f0 = t_kind == KIND_DIRECT ? ((uint64_t)first * t_w) >> 3 : 0u;
            const uint64_t f1 = t_kind == KIND_DIRECT ? ((uint64_t)end * t_w + 7) >> 3 : t_w >> 3;
            const uint64_t origin = stage_bytes<W::SHORT>(wv, t_pay + f0, t_pay + f1);
            bit0 = (uint32_t)((t_pay - origin) << 3);
        }
        for (uint32_t idx = first + (uint32_t)lane; idx < end; idx += 64) {
            uint64_t val = t_base + (uint64_t)idx * t_step;  // Delta without packed deltas
            if (t_kind != KIND_DELTA) {
                const uint64_t m = lds_bits64(wv.lds, bit0 + (t_kind == KIND_DIRECT ? idx * t_w : 0u), t_w);
                val = wv.is_signed ? zigzag(m) : m;
            }
            stream_store1(wv.out + pl.t_start + idx, val);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the slice may be staged into again
    }
}

// The runs rb, rb + 1, ... of the table that start before `own_end` (the end of the wavefront's
// window), at most 64 of them and `avail` (what the table holds from rb on); `batch_beg`: index in
// the output of the first one's first value.  Returns the index after the last one's last value,
// and in `nb` how many runs that were.
template <class W>
__device__ __forceinline__ uint64_t decode_run_batch(const W &wv, const uint64_t *runs, uint32_t rb,
                                                     uint32_t avail, const uint64_t batch_beg,
                                                     const uint64_t own_end, uint32_t &nb) {
    const int lane = wv.lane;
    // lane k holds run rb + k of the batch: its descriptor from the table, its place in the
    // output from a prefix sum of the lengths, its base / second field from its header.  The
    // table's slots past the window's runs may never have been written: a run that turns out to
    // start at or after own_end is not looked at any further (lengths are 1..512 whatever the
    // bits, so the places of the runs before it do not depend on it).
    RunEntry me;
    me.desc = make_desc(0, KIND_SHORT_REPEAT, 0, 1, 0, 1);
    me.base = 0;
    me.aux = 0;
    const bool held = (uint32_t)lane < avail;
    if (held) me.desc = runs[rb + lane];
    const uint32_t incl_all = wave_scan_dpp(held ? desc_len(me.desc) : 0u);
    const bool mine = held && batch_beg + (incl_all - desc_len(me.desc)) < own_end;
    nb = (uint32_t)__builtin_popcountll(__ballot(mine));
    if (nb == 0) return batch_beg;
    if (!mine) me.desc = make_desc(0, KIND_SHORT_REPEAT, 0, 1, 0, 1);
    const uint32_t my_kind = desc_kind(me.desc), my_w = desc_width(me.desc);
    const uint32_t my_len = desc_len(me.desc);
    const uint32_t incl = mine ? incl_all : 0u;
    const uint32_t my_rel = mine ? incl - my_len : 0xFFFFFFFFu;
    const uint64_t batch_all = batch_beg + readlane32(incl_all, (int)nb - 1);
    uint64_t batch_end = batch_all < wv.num_values ? batch_all : wv.num_values;
    if (batch_beg >= batch_end) return batch_all;  // every run of the batch lies after the last wanted value
    if (mine && my_kind == KIND_SHORT_REPEAT) {
        // the repeated value: width / 8 bytes, most significant first, right behind the header byte
        // (rle_v2.rs:283-301); the header scan has checked that they lie inside the stream
        const uint64_t pay = desc_payload(me.desc);
        const uint8_t *p = wv.s + pay;
        const uint32_t sk = (uint32_t)((uintptr_t)p & 3);
        if (my_w >= 8 && pay + (my_w >> 3) <= wv.len) {
            const uint64_t v = global_bits((const uint32_t *)(p - sk), (uint64_t)sk * 8, my_w);
            me.base = wv.is_signed ? zigzag(v) : v;
        }
    }
    if (mine && my_kind != KIND_DIRECT && my_kind != KIND_SHORT_REPEAT) {
        // the header scan accepted this header: the parse cannot fail, and reads what it read
        const uint64_t pos = desc_payload(me.desc) - desc_header_bytes(me.desc);
        ParsedRun r;
        r.base = 0;
        r.aux = 0;
        if (pos < wv.len && parse_run(load_window(wv.s, wv.len, pos), wv.s, wv.len, pos, wv.is_signed, r) == SEG_OK) {
            me.base = r.base;
            me.aux = r.aux;
        }
    }
    uint64_t carry = 0;  // running value of a Delta run that continues into the next step

    for (uint64_t wbase = (batch_beg >> WINDOW_SHIFT) << WINDOW_SHIFT; wbase < batch_end;
         wbase += WINDOW) {
        const uint64_t lo = wbase > batch_beg ? wbase : batch_beg;
        const uint64_t hi = (wbase + WINDOW) < batch_end ? (wbase + WINDOW) : batch_end;

        // ---- which payload bytes does this window need? -----------------------------------
        uint64_t need_b0 = 0, need_b1 = 0;
        bool needs = false, scans = false;
        // (positions relative to the batch's first value fit 32 bits: 64 runs of at most 512)
        const uint32_t lo_rel = (uint32_t)(lo - batch_beg), hi_rel = (uint32_t)(hi - batch_beg);
        if ((uint32_t)lane < nb && my_rel < hi_rel && my_rel + my_len > lo_rel &&
            my_kind != KIND_SHORT_REPEAT && my_w != 0) {
            const uint32_t first = my_rel >= lo_rel ? 0u : lo_rel - my_rel;
            const uint32_t last = (my_rel + my_len) <= hi_rel ? my_len : hi_rel - my_rel;
            uint64_t b0, b1;
            payload_range(my_kind, my_w, first, last, b0, b1);
            needs = b1 > b0;
            need_b0 = desc_payload(me.desc) + b0;
            need_b1 = desc_payload(me.desc) + b1;
            scans = my_kind == KIND_DELTA;
        }
        const uint64_t need_mask = __ballot(needs);
        const bool any_scan = __ballot(scans) != 0;
        const bool any_aux = __ballot((uint32_t)lane < nb && (my_kind == KIND_DELTA || my_kind == KIND_PATCHED_BASE)) != 0;
        bool use_lds = false;
        uint64_t lds_origin = 0;
        if (need_mask) {
            const int k_lo = __ffsll((unsigned long long)need_mask) - 1;
            const int k_hi = 63 - __clzll((unsigned long long)need_mask);
            const uint64_t span_b0 = readlane64(need_b0, k_lo);
            const uint64_t span_b1 = readlane64(need_b1, k_hi);
            if (span_b1 - span_b0 + 15 <= STAGE_BYTES) {
                use_lds = true;
                lds_origin = stage_bytes<W::SHORT>(wv, span_b0, span_b1);
            }
        }

        // ---- decode the window: 4 steps of 128 values -------------------------------------
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t e0 = wbase + 128u * j + 2u * lane;
            uint64_t c[2];
            bool head[2], act[2];
            act[0] = e0 >= lo && e0 < hi;
            act[1] = e0 + 1 >= lo && e0 + 1 < hi;
            // position of the pair's first value in the batch; a value outside [lo, hi) takes the
            // place of its neighbour (or of lo) so that every lane looks up a run that exists
            const uint32_t rel0 = act[0] ? (uint32_t)(e0 - batch_beg)
                                         : (act[1] ? (uint32_t)(e0 + 1 - batch_beg) : (uint32_t)(lo - batch_beg));
            const uint32_t rel1 = act[1] ? (uint32_t)(e0 + 1 - batch_beg) : rel0;
            // largest k with rel_out[k] <= rel0 (lanes >= nb hold 0xFFFFFFFF); runs are not empty,
            // so the pair's second value lies in run k or in run k + 1
            int k = 0;
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) {
                const uint32_t v = __shfl(my_rel, k + sft);
                if (v <= rel0) k += sft;
            }
            const int kn = (k + 1) & 63;
            RunFields fk, fn;
            fk.rel_out = __shfl(my_rel, k);
            fn.rel_out = __shfl(my_rel, kn);
            if (k == 63) fn.rel_out = 0xFFFFFFFFu;  // no run after the batch's 64th
            fk.desc = shfl64(me.desc, k);
            fn.desc = shfl64(me.desc, kn);
            fk.base = shfl64(me.base, k);
            fn.base = shfl64(me.base, kn);
            fk.aux = 0;
            fn.aux = 0;
            if (any_aux) {  // wave-uniform: only Delta and PatchedBase runs carry a second field
                fk.aux = shfl64(me.aux, k);
                fn.aux = shfl64(me.aux, kn);
            }
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const uint32_t e_rel = t ? rel1 : rel0;
                const RunFields f = (t && rel1 >= fn.rel_out) ? fn : fk;
                const uint32_t kind = desc_kind(f.desc), w = desc_width(f.desc);
                const uint32_t idx = e_rel - f.rel_out;
                uint64_t val = f.base;
                bool hd = true;
                // index of the packed field that belongs to this value, if any
                uint32_t fidx = idx;
                bool packed = act[t] && w != 0 && (kind == KIND_DIRECT || kind == KIND_PATCHED_BASE);
                if (kind == KIND_DELTA) {
                    if (w == 0) {
                        val = f.base + (uint64_t)idx * f.aux;  // fixed delta
                    } else if (idx == 1) {
                        val = f.aux;
                        hd = false;
                    } else if (idx >= 2) {
                        packed = act[t];
                        fidx = idx - 2;
                        hd = false;
                    }
                }
                if (packed) {
                    uint64_t m;
                    if (use_lds) {
                        const uint32_t bit =
                            (((uint32_t)desc_payload(f.desc) - (uint32_t)lds_origin) << 3) + fidx * w;
                        m = lds_bits64(wv.lds, bit, w);
                    } else {
                        const uint64_t pay = desc_payload(f.desc);
                        const uint8_t *p = wv.s + pay;
                        const uint32_t sk = (uint32_t)((uintptr_t)p & 3);
                        m = 0;
                        if (pay + (((uint64_t)fidx * w + w + 7) >> 3) <= wv.len)
                            m = global_bits((const uint32_t *)(p - sk), (uint64_t)sk * 8 + (uint64_t)fidx * w, w);
                    }
                    if (kind == KIND_DIRECT) val = wv.is_signed ? zigzag(m) : m;
                    else if (kind == KIND_PATCHED_BASE) val = f.base + m;
                    else val = desc_delta_neg(f.desc) ? 0 - m : m;
                }
                c[t] = val;
                head[t] = hd || !act[t];
            }

            uint64_t v0 = c[0], v1 = c[1];
            if (any_scan) {
                // segmented inclusive sum across the step: heads restart the sum
                bool ph = head[0] || head[1];
                uint64_t ps = head[1] ? c[1] : c[0] + c[1];
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint64_t s_up = shfl_up64(ps, d);
                    const uint32_t h_up = __shfl_up((uint32_t)ph, d);
                    if (lane >= d) {
                        if (!ph) ps += s_up;
                        ph = ph || h_up;
                    }
                }
                uint64_t xs = shfl_up64(ps, 1);
                uint32_t xh = __shfl_up((uint32_t)ph, 1);
                if (lane == 0) { xs = 0; xh = 0; }
                const uint64_t prefix = xh ? xs : xs + carry;
                v0 = head[0] ? c[0] : prefix + c[0];
                v1 = head[1] ? c[1] : v0 + c[1];
                carry = readlane64(v1, 63);
            }
            store_pair(wv, e0, v0, v1, act[0], act[1]);
        }
    }

    // ---- PatchedBase: overwrite the patched positions ----------------------------------------
    uint64_t pb_mask = __ballot((uint32_t)lane < nb && my_kind == KIND_PATCHED_BASE);
    if (pb_mask) {
        __builtin_amdgcn_s_waitcnt(0);  // the unpatched values of this batch are written
        while (pb_mask) {
            const int k = __ffsll((unsigned long long)pb_mask) - 1;
            pb_mask &= pb_mask - 1;
            apply_patches(wv, readlane64(me.desc, k), readlane64(me.base, k), readlane64(me.aux, k),
                          batch_beg + (uint32_t)__builtin_amdgcn_readlane((int)my_rel, k));
        }
    }
    return batch_all;
}

// MODE_SUMS: group_sums[g] receives the (saturating) sum of the values of the runs that start in
// window g; bit 2 of *flags is raised when a PatchedBase run was met, whose patches are not part of
// the sums.  MODE_OFFSETS: group_sums[g] holds the sum of all values before group g, and out[i]
// receives the sum of the values before i.
// SHORT: the variant for columns of short runs (many runs per window), compiled for fewer
// wavefronts per SIMD and more registers each (six: 4 / 5 / 6 / 7 / 8 measured 1.15 / 1.15 / 1.09 /
// 1.14 / 1.32 ms on 5 * 10^8 values in 2.5 * 10^7 runs); only there does a batch of runs without running
// sums take the lane-per-eight-values path (decode_short_batch), whose registers the variant for
// long runs, which wants all eight wavefronts to saturate HBM, cannot afford.
template <int MODE, bool SHORT>
__global__ __launch_bounds__(VAL_WAVES * 64, SHORT ? 6 : 8) void rlev2_values_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const uint64_t *__restrict__ runs,
    const uint32_t *__restrict__ group_first, const RunEntry *__restrict__ group_run,
    uint32_t n_groups, uint32_t n_runs, uint64_t num_values, int is_signed,
    int64_t *__restrict__ out, uint64_t *__restrict__ group_sums, uint32_t *__restrict__ flags) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[VAL_WAVES][STAGE_WORDS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Workgroups are dealt round-robin to the 8 XCDs, each with an L2 of its own: remap the
    // workgroup id so that workgroups sharing an XCD (equal id mod 8) walk one contiguous eighth
    // of the column, and a stream line straddling two neighbouring groups is fetched into one L2.
    uint32_t wg = blockIdx.x;
    {
        const uint32_t nwg = gridDim.x, q = nwg / 8, r = nwg % 8, x = wg % 8;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + wg / 8;
    }
    constexpr bool PAIRED = SHORT && MODE == MODE_VALUES;  // a wavefront takes two neighbouring windows
    const uint32_t g = (wg * VAL_WAVES + wave) * (PAIRED ? 2 : 1);
    if (g >= n_groups) return;  // whole wavefront leaves; no workgroup barrier is used below
    // MODE_VALUES behind a speculative plan of the column (rlev2_regular_plan_kernel): a raised
    // flag means the tables were not all written, and nothing may be decoded from them
    // (the load is only issued here; the entry's loads below go out with it)
    uint32_t aborted = 0;
    // (a plain load: the word is final before this kernel starts, every wavefront reads it, and
    // loads that bypass the caches would all queue at the one memory channel that holds it)
    if (MODE == MODE_VALUES && flags != nullptr) aborted = *flags;
    __shared__ uint32_t marks[VAL_WAVES][WINDOW / 32];
    // (the variant that writes the values of long runs is the one that meets whole PatchedBase runs)
    constexpr bool PATCH_AHEAD = MODE == MODE_VALUES && !SHORT;
    __shared__ uint32_t patch_tab[PATCH_AHEAD ? VAL_WAVES : 1][PATCH_AHEAD ? PATCH_TAB_WORDS : 1];
    WaveOf<MODE, SHORT> wv;
    wv.s = s;
    wv.lds = stage[wave];
    wv.marks = marks[wave];
    wv.patch_tab = PATCH_AHEAD ? patch_tab[wave] : nullptr;
    wv.out = out;
    wv.num_values = num_values;
    wv.len = len;
    wv.lane = threadIdx.x & 63;
    wv.is_signed = is_signed;
    wv.out_aligned = (((uintptr_t)out) & 15) == 0;
    wv.acc = 0;
    wv.saw_patched = 0;
    if (MODE == MODE_OFFSETS) wv.acc = group_sums[g];  // wave-uniform address

    // The runs that start inside window `win`, 64 at a time from run `rb` on (clamped to the table
    // so that a bad index can never walk off it); which run is the last follows from the lengths.
    auto batches_from = [&](uint32_t win, uint64_t rb64, uint64_t beg) {
        const uint64_t own_end = ((uint64_t)win + 1) << WINDOW_SHIFT;
        if (rb64 > n_runs) rb64 = n_runs;
        for (uint32_t rb = (uint32_t)rb64; rb < n_runs && beg < own_end; rb += 64) {
            uint32_t nb;
            beg = decode_run_batch(wv, runs, rb, n_runs - rb, beg, own_end, nb);
            if (nb < 64) break;
        }
    };

    if constexpr (PAIRED) {
        // Two windows per wavefront, so that their dependent memory hops -- the window's copy of
        // its first run, the descriptors of its runs, the stream bytes -- are taken together:
        // both are planned before either is expanded.
        const bool two = g + 1 < n_groups;
        const RunEntry *e0 = group_run + g, *e1 = group_run + (two ? g + 1 : g);
        const uint64_t d0 = readfirstlane64(e0->desc), d1 = readfirstlane64(e1->desc);
        if (__builtin_amdgcn_readfirstlane(aborted)) return;
        const bool multi0 = !(d0 & DESC_SINGLE), multi1 = two && !(d1 & DESC_SINGLE);
        uint64_t r0 = multi0 ? readfirstlane64(e0->base) : n_runs, r1 = multi1 ? readfirstlane64(e1->base) : n_runs;
        if (r0 > n_runs) r0 = n_runs;
        if (r1 > n_runs) r1 = n_runs;
        const bool held0 = multi0 && (uint32_t)wv.lane < n_runs - (uint32_t)r0;
        const bool held1 = multi1 && (uint32_t)wv.lane < n_runs - (uint32_t)r1;
        uint64_t desc0 = 0, desc1 = 0;
        if (held0) desc0 = runs[r0 + wv.lane];
        if (held1) desc1 = runs[r1 + wv.lane];
        ShortPlan p0, p1;
        int st0 = SHORT_OTHER, st1 = SHORT_OTHER;
        if (multi0) st0 = short_plan(wv, desc0, held0, readfirstlane64(e0->out_off), ((uint64_t)g + 1) << WINDOW_SHIFT, p0);
        if (multi1) st1 = short_plan(wv, desc1, held1, readfirstlane64(e1->out_off), ((uint64_t)g + 2) << WINDOW_SHIFT, p1);
        // window g
        if (!multi0) {
            decode_single_run(wv, d0, readfirstlane64(e0->out_off), readfirstlane64(e0->base), readfirstlane64(e0->aux));
        } else if (st0 == SHORT_OTHER) {
            batches_from(g, r0, readfirstlane64(e0->out_off));
        } else {
            if (st0 == SHORT_PLANNED) short_expand(wv, p0);
            if (p0.nb == 64) batches_from(g, r0 + 64, p0.batch_all);  // more than 64 runs start in the window
        }
        // window g + 1
        if (two) {
            if (!multi1) {
                decode_single_run(wv, d1, readfirstlane64(e1->out_off), readfirstlane64(e1->base), readfirstlane64(e1->aux));
            } else if (st1 == SHORT_OTHER) {
                batches_from(g + 1, r1, readfirstlane64(e1->out_off));
            } else {
                if (st1 == SHORT_PLANNED) short_expand(wv, p1);
                if (p1.nb == 64) batches_from(g + 1, r1 + 64, p1.batch_all);
            }
        }
        return;
    }

    // The window's own copy of its first run: when that run is the only one starting in the
    // window (flagged by the header scan) nothing else has to be loaded.  (One kernel for both
    // kinds of group: a split into two launches with separate register budgets measured no
    // faster on single-run columns and slower on columns of short runs.)
    const RunEntry *e = group_run + g;  // wave-uniform address
    const uint64_t desc = readfirstlane64(e->desc);
    if (__builtin_amdgcn_readfirstlane(aborted)) return;
    if (desc & DESC_SINGLE) {
        decode_single_run(wv, desc, readfirstlane64(e->out_off), readfirstlane64(e->base),
                          readfirstlane64(e->aux));
    } else {
        // the window's copy carries the index of its first run
        batches_from(g, readfirstlane64(e->base), readfirstlane64(e->out_off));
    }
    if (MODE == MODE_SUMS) {
        uint64_t t = wv.acc;
        for (int d = 32; d >= 1; d >>= 1) t = sat_add(t, shfl64(t, wv.lane ^ d));
        if (wv.lane == 0) {
            group_sums[g] = t;
            if (wv.saw_patched) atomicOr(flags, 4u);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// MODE_SUMS for tables of long runs, four windows to a wavefront.  A wavefront per window spends
// its start, a chain of three dependent memory hops (the window's entry, its stream bytes, LDS)
// and a 64-lane reduction on 400 bytes of stream -- 2.5 * 10^8 string lengths are half a million
// such wavefronts, 0.23 ms for 0.19 GB.  Here the four entries are fetched by four lanes at once,
// the bytes of all four windows are asked for before any is looked at (a window of one unsigned
// Direct run of at most 16-bit fields is at most 64 chunks of 16 bytes: one load per lane), all
// land in the wavefront's LDS slice side by side, and each window's 512 fields are added up eight
// to a lane with one 32-bit DPP reduction (512 fields of 16 bits cannot wrap).  A wavefront with a
// window of any other kind takes its windows one by one as rlev2_values_kernel<MODE_SUMS> does.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t SUMS_WINDOWS = 4;
constexpr uint32_t SUMS_PITCH_WORDS = 264;  // 64 chunks and the word after, per window

// the runs that start in window g, as rlev2_values_kernel expands them
template <class W>
__device__ __forceinline__ void expand_window_runs(const W &wv, const uint64_t *__restrict__ runs, uint32_t n_runs,
                                                   const RunEntry *__restrict__ group_run, uint32_t g) {
    const RunEntry *e = group_run + g;  // wave-uniform address
    const uint64_t desc = readfirstlane64(e->desc);
    if (desc & DESC_SINGLE) {
        decode_single_run(wv, desc, readfirstlane64(e->out_off), readfirstlane64(e->base), readfirstlane64(e->aux));
        return;
    }
    const uint64_t own_end = ((uint64_t)g + 1) << WINDOW_SHIFT;
    uint64_t rb64 = readfirstlane64(e->base), beg = readfirstlane64(e->out_off);
    if (rb64 > n_runs) rb64 = n_runs;
    for (uint32_t rb = (uint32_t)rb64; rb < n_runs && beg < own_end; rb += 64) {
        uint32_t nb;
        beg = decode_run_batch(wv, runs, rb, n_runs - rb, beg, own_end, nb);
        if (nb < 64) break;
    }
}

static_assert(SUMS_WINDOWS * SUMS_PITCH_WORDS <= STAGE_WORDS, "four windows side by side in a wavefront's slice");

// group_sums[g0 .. g0 + nk) as rlev2_values_kernel<MODE_SUMS> computes them, by one wavefront
__device__ __attribute__((noinline)) void general_window_sums(
    const uint8_t *s, uint64_t len, const uint64_t *runs, const RunEntry *group_run, uint32_t n_runs,
    uint64_t num_values, int is_signed, uint64_t *group_sums, uint32_t *flags, uint32_t *lds, uint32_t *marks,
    uint32_t g0, uint32_t nk) {
    const uint32_t lane = threadIdx.x & 63;
    WaveOf<MODE_SUMS> wv;
    wv.s = s;
    wv.lds = lds;
    wv.marks = marks;
    wv.patch_tab = nullptr;
    wv.out = nullptr;
    wv.num_values = num_values;
    wv.len = len;
    wv.lane = (int)lane;
    wv.is_signed = is_signed;
    wv.out_aligned = false;
    wv.saw_patched = 0;
    for (uint32_t k = 0; k < nk; k++) {
        wv.acc = 0;
        expand_window_runs(wv, runs, n_runs, group_run, g0 + k);
        uint64_t t = wv.acc;
        for (int d = 32; d >= 1; d >>= 1) t = sat_add(t, shfl64(t, (int)lane ^ d));
        if (lane == 0) group_sums[g0 + k] = t;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the slice is staged into again
    }
    if (lane == 0 && wv.saw_patched) atomicOr(flags, 4u);
}

__global__ __launch_bounds__(VAL_WAVES * 64, 8) void rlev2_quad_sums_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const uint64_t *__restrict__ runs,
    const RunEntry *__restrict__ group_run, uint32_t n_groups, uint32_t n_runs, uint64_t num_values,
    int is_signed, uint64_t *__restrict__ group_sums, uint32_t *__restrict__ flags) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[VAL_WAVES][STAGE_WORDS];
    __shared__ uint32_t marks[VAL_WAVES][WINDOW / 32];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    uint32_t wg = blockIdx.x;  // XCD-contiguous remap, as in rlev2_values_kernel
    {
        const uint32_t nwg = gridDim.x, q = nwg / 8, r = nwg % 8, x = wg % 8;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + wg / 8;
    }
    const uint64_t g0_64 = ((uint64_t)wg * VAL_WAVES + wave) * SUMS_WINDOWS;
    if (g0_64 >= n_groups) return;  // whole wavefront; no workgroup barrier below
    const uint32_t g0 = (uint32_t)g0_64;
    const uint32_t nk = n_groups - g0 < SUMS_WINDOWS ? n_groups - g0 : SUMS_WINDOWS;
    uint32_t *lds = stage[wave];
    // lane k: the entry of window g0 + k, and whether the window is of the simple kind
    uint64_t desc = 0, out_off = 0;
    if (lane < nk) {
        desc = group_run[g0 + lane].desc;
        out_off = group_run[g0 + lane].out_off;
    }
    const uint32_t w = desc_width(desc), rlen = desc_len(desc);
    const uint64_t pay = desc_payload(desc);
    const uint32_t n = (lane >= nk || out_off >= num_values) ? 0u
                       : (num_values - out_off < rlen ? (uint32_t)(num_values - out_off) : rlen);
    const uint32_t bytes = (w * n + 7) >> 3;  // of the values that are wanted
    const uint32_t skew = (uint32_t)((uintptr_t)(s + pay) & 15);
    const bool simple = lane >= nk || ((desc & DESC_SINGLE) && desc_kind(desc) == KIND_DIRECT && !is_signed && w <= 16 &&
                                       pay <= len && bytes <= len - pay && bytes + skew <= 1024);
    if (__ballot(!simple) == 0) {
        uint4 v[SUMS_WINDOWS];
#pragma unroll
        for (uint32_t k = 0; k < SUMS_WINDOWS; k++) {
            const uint64_t pay_k = readlane64(pay, (int)k);
            const uint32_t bytes_k = readlane32(bytes, (int)k), skew_k = readlane32(skew, (int)k);
            const uint32_t n16 = bytes_k ? (bytes_k + skew_k + 15) >> 4 : 0;  // at most 64
            v[k] = make_uint4(0, 0, 0, 0);
            if (lane < n16) v[k] = ((const uint4 *)(s + pay_k - skew_k))[lane];
        }
#pragma unroll
        for (uint32_t k = 0; k < SUMS_WINDOWS; k++) {
            uint4 x = v[k];
            x.x = __builtin_bswap32(x.x);
            x.y = __builtin_bswap32(x.y);
            x.z = __builtin_bswap32(x.z);
            x.w = __builtin_bswap32(x.w);
            ((uint4 *)(lds + k * SUMS_PITCH_WORDS))[lane] = x;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (uint32_t k = 0; k < SUMS_WINDOWS; k++) {
            const uint32_t w_k = readlane32(w, (int)k), n_k = readlane32(n, (int)k);
            const uint32_t bit0 = readlane32(skew, (int)k) * 8;
            const uint32_t *slot = lds + k * SUMS_PITCH_WORDS;
            uint32_t part = 0;
#pragma unroll
            for (uint32_t j = 0; j < WINDOW / 64; j++) {
                const uint32_t idx = lane + 64 * j;
                if (idx < n_k) part += lds_bits32(slot, bit0 + idx * w_k, w_k);
            }
            const uint32_t total = wave_scan_dpp(part);
            if (lane == 63 && k < nk) group_sums[g0 + k] = total;
        }
        return;
    }
    // windows of other kinds: one by one (a call, so that the simple windows above do not share
    // their registers with the general expansion)
    general_window_sums(s, len, runs, group_run, n_runs, num_values, is_signed, group_sums, flags, lds, marks[wave], g0, nk);
}

// ---------------------------------------------------------------------------------------------
// K3 behind a plan of the column (rlev2_regular_plan_kernel): every window is one run, and its entry
// in group_run says everything about it.  The same expansion as rlev2_values_kernel makes of a
// single-run window, in a kernel of its own: without the code for windows that several runs share
// it keeps clear of the register limit that eight wavefronts per SIMD set (43 registers instead of
// 64 and a few spilled; a sorted column of 16-bit deltas: 1.75 -> 1.64 ms per 10^9 values).
// (A wavefront that takes two neighbouring windows -- both entries fetched together, then the
// stream bytes of both, then the two expansions -- measured slower, 1.60 -> 1.66 ms: the loads of
// twice the bytes take twice as long to come back, the memory system being what bounds the kernel,
// and the second expansion waits behind the first.)
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(VAL_WAVES * 64, 8) void rlev2_planned_values_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const RunEntry *__restrict__ group_run, uint32_t n_groups,
    uint64_t num_values, int is_signed, int64_t *__restrict__ out, const uint32_t *__restrict__ flags) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[VAL_WAVES][STAGE_WORDS];
    __shared__ uint32_t patch_tab[VAL_WAVES][PATCH_TAB_WORDS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t wg = blockIdx.x;  // XCD-contiguous remap, as in rlev2_values_kernel
    {
        const uint32_t nwg = gridDim.x, q = nwg / 8, r = nwg % 8, x = wg % 8;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + wg / 8;
    }
    const uint32_t g = wg * VAL_WAVES + wave;
    if (g >= n_groups) return;  // whole wavefront; no workgroup barrier below
    // a raised flag: the plan did not hold and the entries are not all written (a plain load: the
    // word is final before this kernel starts; looked at when the entry's loads are out as well)
    const uint32_t aborted = flags ? *flags : 0u;
    WaveOf<MODE_VALUES> wv;
    wv.s = s;
    wv.lds = stage[wave];
    wv.marks = nullptr;
    wv.patch_tab = patch_tab[wave];
    wv.out = out;
    wv.num_values = num_values;
    wv.len = len;
    wv.lane = threadIdx.x & 63;
    wv.is_signed = is_signed;
    wv.out_aligned = (((uintptr_t)out) & 15) == 0;
    wv.acc = 0;
    wv.saw_patched = 0;
    const RunEntry *e = group_run + g;  // wave-uniform address
    const uint64_t desc = readfirstlane64(e->desc), out_off = readfirstlane64(e->out_off);
    const uint64_t base = readfirstlane64(e->base), aux = readfirstlane64(e->aux);
    if (__builtin_amdgcn_readfirstlane(aborted)) return;
    if (desc & DESC_SINGLE) decode_single_run(wv, desc, out_off, base, aux);
}

// ---------------------------------------------------------------------------------------------
// K3 without K1/K2 for the regular column: every window of 512 values is one Direct run of 512
// values, and the runs between two entry points all have one size.  Then nothing about a run has to
// be looked up: the segment of window g is g / R (R runs per segment), the run's size S is the
// distance to the next entry point divided by R, its header sits at entry + k * S and its width is
// (S - 2) / 64.  Each wavefront checks the one thing this rests on -- that the two header bytes in
// front of the payload it stages say "Direct, that width, 512 values" -- and by induction over k
// (the run at the entry point is real, a real run of S bytes ends where the next one starts, the
// R-th ends at the next entry point) the headers are the ones a walk of the chain would find
// (rle_v2.rs:135-157, 480-514).  Any deviation raises *flag and the caller decodes the stream the
// general way; what was written until then is overwritten.
// The stream's last window goes through the general single-run code: its run may be of any kind
// and length (at least the values still wanted).
// ---------------------------------------------------------------------------------------------

// Raises a speculation flag.  A column that is not for a speculative path fails in most of its
// wavefronts, and an atomic from each of them on the one word would queue up for longer than the
// whole launch takes: the word is looked at first -- past the caches, which is affordable on this
// path only -- and the atomic is left to the few that find it still clear.
__device__ __forceinline__ void raise_flag(uint32_t *flag) {
    if (*(const volatile uint32_t *)flag == 0) atomicOr(flag, 1u);
}

__device__ __forceinline__ uint32_t code_from_width(uint32_t w) {
    if (w >= 1 && w <= 24) return w - 1;
    if (w <= 32) return (w & 1) || w < 26 ? 0xFFu : 24 + (w - 26) / 2;
    if (w <= 64) return (w & 7) ? 0xFFu : 28 + (w - 40) / 8;
    return 0xFFu;
}

__global__ __launch_bounds__(VAL_WAVES * 64, 8) void rlev2_strided_values_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const uint64_t *__restrict__ entries, uint32_t n_seg,
    uint32_t n_win, uint32_t runs_hint, uint32_t stride_vals, uint64_t seg_magic, uint64_t num_values,
    int is_signed, int64_t *__restrict__ out, uint32_t *__restrict__ flag) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[VAL_WAVES][STAGE_WORDS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t wg = blockIdx.x;  // XCD-contiguous remap, as in rlev2_values_kernel
    {
        const uint32_t nwg = gridDim.x, q = nwg / 8, r = nwg % 8, x = wg % 8;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + wg / 8;
    }
    const uint32_t g = wg * VAL_WAVES + wave;
    if (g >= n_win) return;
    // somebody found the column irregular: the rest of this launch is wasted work.  (A plain load,
    // through the cache, and only looked at once the loads of the entry points are out as well: a
    // load past the caches that is waited for here costs every wavefront a memory round trip of
    // its own before it can ask for anything else.)
    const uint32_t flag_now = *flag;
    WaveOf<MODE_VALUES> wv;
    wv.s = s;
    wv.lds = stage[wave];
    wv.marks = nullptr;  // (only the variant for short runs uses them)
    wv.patch_tab = nullptr;
    wv.out = out;
    wv.num_values = num_values;
    wv.len = len;
    wv.lane = threadIdx.x & 63;
    wv.is_signed = is_signed;
    wv.out_aligned = true;  // the launcher insists
    wv.acc = 0;
    wv.saw_patched = 0;
    auto fail = [&]() {
        if (wv.lane == 0) raise_flag(flag);
    };
    // How many runs lie between two entry points: the distance from the stream's start to its
    // second entry point over the size of the first run, which must be a Direct run of 512 values
    // like all the others (two bytes that every wavefront finds in cache).  All segments but the
    // last hold that many windows, the last one the rest (at least one).
    // (runs_hint: the counts of windows and entry points leave one possibility -- true of any column
    // of many segments -- and the host passes it; the checks below do not care where the number
    // comes from, and a wavefront that need not work it out starts on its bytes a round trip sooner)
    // stride_vals: the segments hold that many values each, whatever that is in runs -- entry point
    // q is the header of the run that holds value q * stride_vals, the first window of segment q is
    // floor(q * stride_vals / 512), and a segment holds the windows up to the next one's first (the
    // row groups of a file: 10 000 values make 19 or 20 runs).  seg_magic = 2^64 / stride_vals
    // rounded up turns the division into a multiplication (exact below 2^23; checked all the same).
    uint32_t runs_per_seg = n_win, seg = 0, first = 0;
    if (n_seg > 1 && stride_vals != 0) {
        const uint64_t S = stride_vals;
        seg = (uint32_t)__umul64hi((((uint64_t)g + 1) << WINDOW_SHIFT) - 1, seg_magic);
        if (seg >= n_seg) seg = n_seg - 1;
        uint64_t fw = ((uint64_t)seg * S) >> WINDOW_SHIFT;
        if (fw > g && seg > 0) {
            seg--;
            fw = ((uint64_t)seg * S) >> WINDOW_SHIFT;
        }
        uint64_t fw_next = seg + 1 < n_seg ? (((uint64_t)seg + 1) * S) >> WINDOW_SHIFT : (uint64_t)n_win;
        if (fw_next <= g && seg + 1 < n_seg) {
            seg++;
            fw = fw_next;
            fw_next = seg + 1 < n_seg ? (((uint64_t)seg + 1) * S) >> WINDOW_SHIFT : (uint64_t)n_win;
        }
        if (fw > g || fw_next <= g || fw_next > n_win) return fail();
        first = (uint32_t)fw;
        runs_per_seg = (uint32_t)(fw_next - fw);
    } else if (n_seg > 1 && runs_hint != 0) {
        runs_per_seg = runs_hint;
        if ((uint64_t)runs_per_seg * (n_seg - 1) >= n_win) return fail();
    } else if (n_seg > 1) {
        const uint64_t d0 = readfirstlane64(entries[1]);
        if (len < 2 || d0 > 0xFFFFFFFFull) return fail();
        const uint32_t h0 = __builtin_amdgcn_readfirstlane((uint32_t)s[0]);
        const uint32_t h1 = __builtin_amdgcn_readfirstlane((uint32_t)s[1]);
        if ((h0 >> 6) != KIND_DIRECT || (((h0 & 1u) << 8) | h1) != 511u) return fail();
        const uint32_t s0 = 2u + 64u * width_from_code((h0 >> 1) & 31u);
        runs_per_seg = (uint32_t)d0 / s0;
        if (runs_per_seg * s0 != (uint32_t)d0 || runs_per_seg == 0 ||
            (uint64_t)runs_per_seg * (n_seg - 1) >= n_win)
            return fail();
    }
    if (stride_vals == 0 || n_seg <= 1) {
        seg = g / runs_per_seg;
        if (seg >= n_seg) seg = n_seg - 1;  // the last segment holds whatever is left
        first = seg * runs_per_seg;
    }
    const uint32_t k = g - first;
    const uint64_t e0 = (seg && entries) ? readfirstlane64(entries[seg]) : 0;
    if (e0 >= len) return fail();
    uint64_t size;
    if (seg + 1 < n_seg) {
        const uint64_t e1 = readfirstlane64(entries[seg + 1]);
        if (e1 <= e0 || e1 - e0 > 0xFFFFFFFFull) return fail();
        const uint32_t d = (uint32_t)(e1 - e0);
        const uint32_t q = d / runs_per_seg;
        if (q * runs_per_seg != d) return fail();
        size = q;
    } else {
        // no entry point follows the last segment: the size of its first run
        ParsedRun r;
        if (parse_run(load_window(s, len, e0), s, len, e0, is_signed, r) != SEG_OK) return fail();
        size = readfirstlane64(r.run_end - e0);
    }
    const uint64_t pos = e0 + (uint64_t)k * size;
    if (__builtin_amdgcn_readfirstlane(flag_now)) return;
    if (pos >= len) return fail();
    const uint64_t wbase = (uint64_t)g << WINDOW_SHIFT;
    if (g + 1 == n_win) {  // the stream's last window
        ParsedRun r;
        if (parse_run(load_window(s, len, pos), s, len, pos, is_signed, r) != SEG_OK) return fail();
        if ((uint64_t)r.len < num_values - wbase) return fail();  // another run follows inside the window
        const uint64_t desc = make_desc(r.payload_off, r.kind, r.width, r.len, r.delta_neg, r.header_bytes);
        decode_single_run(wv, readfirstlane64(desc), wbase, readfirstlane64(r.base), readfirstlane64(r.aux));
        return;
    }
    if (size < 2 + 64 || ((size - 2) & 63) != 0 || pos + size > len) return fail();
    const uint32_t w = (uint32_t)((size - 2) >> 6);
    const uint32_t code = code_from_width(w);
    if (code > 31) return fail();
    const uint64_t origin = stage_bytes(wv, pos, pos + size);  // size <= 4098: fits the slice
    const uint32_t o = (uint32_t)(pos - origin);
    // "01 code 1" + 0xFF: Direct, this width, length 512 (rle_v2.rs:135-146)
    if (lds_bits32(wv.lds, o * 8, 16) != (((0x40u | (code << 1) | 1u) << 8) | 0xFFu)) return fail();
    const bool zz = is_signed != 0;
    const uint32_t bit_lane = (o + 2) * 8 + 2u * (uint32_t)wv.lane * w;
    int64_t *wout = out + wbase;
    if (w <= 32) direct_window<true, false>(wv, wout, w, zz, 0, bit_lane, 0, WINDOW);
    else direct_window<true, true>(wv, wout, w, zz, 0, bit_lane, 0, WINDOW);
}

// What a speculative launch came to, where the host can see it: *flag into the `error` field of
// the pinned totals, stamped last (the protocol of block_totals_scan_kernel).  The flag is left
// zero for the next launch.
__global__ void spec_publish_kernel(uint32_t *__restrict__ flag, ScanTotals *host_totals, uint32_t stamp) {
    if (threadIdx.x || blockIdx.x) return;
    volatile ScanTotals *h = host_totals;
    h->error = flag[0];
    h->error_seg = flag[1];  // (what the planning kernels noted beside their verdict)
    flag[0] = 0;
    flag[1] = 0;
    flag[2] = 0;  // (rlev2_selfplan_values_kernel's count of walks)
    __threadfence_system();
    h->pad = stamp;
}

hipError_t launch_rlev2_strided_values(hipStream_t st, const uint8_t *stream, uint64_t len,
                                       const uint64_t *entries, uint32_t n_seg, uint32_t n_win,
                                       uint64_t num_values, int is_signed, int64_t *out, uint32_t *flag,
                                       ScanTotals *host_totals, uint32_t stamp, uint32_t stride_vals) {
    if (n_win == 0 || n_seg == 0 || (((uintptr_t)out) & 15)) return hipErrorInvalidValue;
    const uint32_t blocks = (n_win + VAL_WAVES - 1) / VAL_WAVES;
    uint32_t r_lo = 0, r_hi = 0, runs_hint = 0;
    if (rlev2_regular_plan_shape(n_seg, n_win, &r_lo, &r_hi) && r_lo == r_hi) runs_hint = r_lo;
    if (stride_vals && (stride_vals < WINDOW || stride_vals >= (1u << 23) || n_seg < 2)) stride_vals = 0;
    const uint64_t magic = stride_vals ? ~0ull / stride_vals + 1 : 0;
    rlev2_strided_values_kernel<<<blocks, VAL_WAVES * 64, 0, st>>>(stream, len, entries, n_seg, n_win, runs_hint,
                                                                  stride_vals, magic, num_values, is_signed, out, flag);
    spec_publish_kernel<<<1, 64, 0, st>>>(flag, host_totals, stamp);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K1 + K2 in one pass, without prefix sums, for the column whose every window of 512 values is one
// run of 512 values of any kind (sorted keys and timestamps come as full Delta runs) with
// `runs_per_seg` runs between two entry points.  Run g is then window g, its place in the output is
// 512 g, and the only unknown is where its header lies: run sizes differ by the bytes of their
// varints, so positions are not arithmetic as they are for Direct runs.  One lane per run *guesses*:
// it interpolates between the entry points on both sides of its segment and looks, in the 64 bytes
// around that offset, for the two bytes every header of a 512-value run of the segment's kind and
// width starts with.  One hit is the guess; none or several, and the lane takes the end of the run
// its neighbour guessed, or failing that walks the chain from the segment's entry point (at most
// 64 hops, else the column is not for this path).
// Guesses prove nothing; the links do.  Window g's run must hold 512 values and end exactly where
// lane g + 1 says run g + 1 starts, the first run of a segment starts at its entry point, and so,
// by induction over g, every position is the one a walk of the header chain arrives at
// (rle_v2.rs:441-461, 480-514).  Any link that does not hold raises *flag: the value kernel
// launched behind this one then does nothing and the caller decodes the stream the general way.
// A wavefront takes 63 windows; its last lane computes the first window of the next wavefront a
// second time (the choice is a function of the stream alone) so that every link is checked.
// ---------------------------------------------------------------------------------------------

// 0x80 in every byte of v that is zero (exact: no carries between bytes)
__device__ __forceinline__ uint64_t zero_bytes(uint64_t v) {
    const uint64_t m = 0x7F7F7F7F7F7F7F7Full;
    return ~(((v & m) + m) | v | m);
}

constexpr uint32_t PLAN_MAX_WALK = 64;

// One lane's share of the plan: window g64 of the column (any lane whose window lies past the last
// one to plan computes a copy of a neighbour and is not looked at).  `live`: this lane's window is
// to be checked and delivered; all 64 lanes must be here (the links go through shuffles).
// Returns the window's entry (desc 0: no entry, and *flag has been raised).
__device__ __forceinline__ RunEntry plan_window(
    const uint8_t *__restrict__ s, uint64_t len, const uint64_t *__restrict__ entries, uint32_t n_seg,
    uint32_t n_win, uint32_t r_lo, uint32_t r_hi, uint64_t num_values, int is_signed, uint64_t g64,
    uint32_t lane, bool live, uint32_t *__restrict__ flag) {
    const uint32_t g = g64 < n_win ? (uint32_t)g64 : n_win - 1;
    bool bad = false;
    // Runs per segment: the counts of windows and entry points leave r_lo .. r_hi (one value for
    // any column of many segments); among several, the one nearest to what the stream's average
    // bytes per window make of the first segment.  A wrong choice fails the links.
    uint32_t runs_per_seg = r_lo;
    if (r_lo != r_hi) {
        const double est = (double)entries[1] * (double)n_win / (double)(len ? len : 1) + 0.5;
        runs_per_seg = est < (double)r_lo ? r_lo : (est > (double)r_hi ? r_hi : (uint32_t)est);
        runs_per_seg = __builtin_amdgcn_readfirstlane(runs_per_seg);
    }
    uint32_t seg = g / runs_per_seg;
    if (seg >= n_seg) seg = n_seg - 1;  // the last segment holds whatever is left
    const uint32_t k = g - seg * runs_per_seg;
    const uint64_t e0 = seg ? entries[seg] : 0;
    if (e0 >= len) bad = true;
    // Where to look: 64 stream bytes from a 16-byte boundary, fetched with four 16-byte loads.
    // The first run of a segment sits at the entry point; any other is expected at the offset
    // interpolated between this segment's entry point and the next one's (the last segment has
    // none after it and borrows the spacing of the segment before) and searched for from 16 bytes
    // before that on.  Headers that start in the block's first 48 bytes can be parsed out of it.
    const bool last_seg = seg + 1 == n_seg;
    const uint64_t ea = last_seg ? (seg > 1 ? entries[seg - 1] : 0) : e0;
    const uint64_t eb = last_seg ? e0 : entries[seg + 1];
    uint64_t from = e0;
    if (k != 0) {
        const double span = eb > ea ? (double)(eb - ea) : 0.0;
        const uint64_t mid = e0 + (uint64_t)(span * (double)k * __builtin_amdgcn_rcp((double)runs_per_seg));
        from = mid > e0 + 16 ? mid - 16 : e0 + 1;
    }
    if (from >= len) from = bad ? 0 : e0;
    const uintptr_t a0 = (uintptr_t)(s + from) & ~(uintptr_t)15;
    const uintptr_t end = (uintptr_t)(s + len);
    const uint4 *blk = (const uint4 *)a0;
    uint64_t q[8];
    {
        // a chunk at or past the end is replaced by a re-read of the first (select on the address)
        uint4 c[4];
#pragma unroll
        for (int i = 0; i < 4; i++) c[i] = *((a0 + 16 * i < end) ? blk + i : blk);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            q[2 * i] = ((uint64_t)c[i].y << 32) | c[i].x;
            q[2 * i + 1] = ((uint64_t)c[i].w << 32) | c[i].z;
        }
    }
    uint32_t at = (uint32_t)((uintptr_t)(s + from) - a0);  // k == 0: the entry point itself
    bool found = k == 0;
    if (k != 0 && !bad) {
        // the two bytes a run of 512 values of this segment's kind and width starts with
        const uint64_t first = 0x0101010101010101ull * (uint64_t)s[e0];
        uint32_t hits = 0;
        uint64_t ff_next = zero_bytes(~q[0]);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const uint64_t is_h0 = zero_bytes(q[i] ^ first);
            const uint64_t ff = ff_next;
            ff_next = zero_bytes(~q[i + 1]);
            const uint64_t m = is_h0 & ((ff >> 8) | (ff_next << 56));  // byte j is h0 and byte j + 1 is 0xFF
            if (m) {
                hits += __builtin_popcountll(m);
                at = 8 * i + (__builtin_ctzll(m) >> 3);
            }
        }
        // (chunks read in place of ones past the end may add hits that are not there: more than
        // one hit, or one outside the stream, only sends the lane on the walk)
        const uint64_t cand = (uint64_t)(a0 - (uintptr_t)s) + at;
        found = hits == 1 && cand > e0 && cand < len && len - cand >= 2;
    }
    uint64_t pos = (uint64_t)(a0 - (uintptr_t)s) + at;
    Window16 w;
    {
        // the 16 bytes at `at` out of the block (at <= 47)
        const uint32_t wi = at >> 3, sh = (at & 7) * 8;
        uint64_t x0 = q[0], x1 = q[1], x2 = q[2];
#pragma unroll
        for (int i = 1; i < 6; i++)
            if (wi == (uint32_t)i) {
                x0 = q[i];
                x1 = q[i + 1];
                x2 = q[i + 2];
            }
        w.lo = sh ? (x0 >> sh) | (x1 << (64 - sh)) : x0;
        w.hi = sh ? (x1 >> sh) | (x2 << (64 - sh)) : x1;
    }
    ParsedRun r;
    r.run_end = 0;
    r.base = 0;
    r.aux = 0;
    r.kind = 0;
    r.len = 0;
    bool parsed = false;
    if (found && !bad) {
        if (pos >= len || parse_run(w, s, len, pos, is_signed, r) != SEG_OK) bad = true;
        parsed = !bad;
    }
    // No guess, or several: where the lane before says its run ends, if it had one guess (the links
    // below prove the whole of it or fail) ...
    {
        const uint64_t prev_end = shfl_up64(r.run_end, 1);
        const bool prev_parsed = __shfl_up((int)parsed, 1) != 0;
        if (!found && !bad && lane > 0 && prev_parsed && g64 < n_win) {
            pos = prev_end;
            if (pos >= len || parse_run(load_window(s, len, pos), s, len, pos, is_signed, r) != SEG_OK) bad = true;
            parsed = !bad;
            found = true;
        }
    }
    if (!found && !bad) {
        // ... else the chain from the entry point, a run at a time
        if (k > PLAN_MAX_WALK) bad = true;
        pos = e0;
        for (uint32_t j = 0; j < k && !bad; j++) {
            ParsedRun hop;
            if (pos >= len || parse_run(load_window(s, len, pos), s, len, pos, is_signed, hop) != SEG_OK)
                bad = true;
            else
                pos = hop.run_end;
        }
        if (!bad && (pos >= len || parse_run(load_window(s, len, pos), s, len, pos, is_signed, r) != SEG_OK))
            bad = true;
    }
    uint64_t desc = 0;
    if (!bad) {
        desc = make_desc(r.payload_off, r.kind, r.width, r.len, r.delta_neg, r.header_bytes) | DESC_SINGLE;
        const uint64_t wbase = (uint64_t)g << WINDOW_SHIFT;
        if (g + 1 == n_win) {
            // the column's last run: any length that holds the values still wanted
            if (wbase >= num_values || (uint64_t)r.len < num_values - wbase) bad = true;
        } else if (r.len != WINDOW) {
            bad = true;
        }
    }
    // the link: this run ends where the next lane's begins (lanes past the last window mirror it
    // and are not looked at)
    const uint64_t next_pos = shfl64(pos, (int)(lane < 63 ? lane + 1 : 63));
    const bool linked = live && lane < 63 && g + 1 < n_win && !bad;
    if (linked && r.run_end != next_pos) bad = true;
    if (live && bad) raise_flag(flag);
    // flag[1]: some run is not a Direct run, or not of its successor's size -- a column that
    // leaves it zero is one the strided kernel decodes without this pass
    const uint64_t size = r.run_end - pos;
    const uint64_t next_size = shfl64(size, (int)(lane < 63 ? lane + 1 : 63));
    if (linked && (r.kind != KIND_DIRECT || (g + 2 < n_win && size != next_size))) {
        if (__builtin_amdgcn_readfirstlane(*(const volatile uint32_t *)(flag + 1)) == 0) atomicOr(flag + 1, 1u);
    }
    RunEntry e;
    e.desc = (live && lane < 63 && !bad) ? desc : 0;
    e.out_off = (uint64_t)g << WINDOW_SHIFT;
    e.base = r.base;
    e.aux = r.aux;
    return e;
}

__global__ __launch_bounds__(256) void rlev2_regular_plan_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const uint64_t *__restrict__ entries, uint32_t n_seg,
    uint32_t n_win, uint32_t r_lo, uint32_t r_hi, uint64_t num_values, int is_signed,
    RunEntry *__restrict__ group_run, uint32_t *__restrict__ flag) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave_id = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint64_t g64 = wave_id * 63 + lane;
    if (wave_id * 63 >= n_win) return;  // whole wavefront
    if (__builtin_amdgcn_readfirstlane(*flag)) return;  // (through the cache: it only saves time)
    const RunEntry e = plan_window(s, len, entries, n_seg, n_win, r_lo, r_hi, num_values, is_signed, g64, lane,
                                   g64 < n_win, flag);
    if (e.desc != 0) group_run[g64] = e;
}

bool rlev2_regular_plan_shape(uint32_t n_seg, uint32_t n_win, uint32_t *r_lo, uint32_t *r_hi) {
    if (n_win == 0 || n_seg < 2 || n_seg > n_win) return false;
    // n_win = (n_seg - 1) * runs_per_seg + (1 .. runs_per_seg windows of the last segment)
    *r_lo = (n_win + n_seg - 1) / n_seg;
    *r_hi = (n_win - 1) / (n_seg - 1);
    return *r_lo <= *r_hi;
}

// ---------------------------------------------------------------------------------------------
// Plan and expansion in one launch, for the same kind of column -- every 512 values one run of 512
// values, the runs of a segment alike in kind and width -- and for segments of unequal run counts:
// row group k of a file starts at value k * S (S = the row index stride, 10 000 by default, no
// multiple of 512), so its entry point is the header of run floor(k * S / 512) and a segment holds
// 19 or 20 runs.  Window g then lies in the last segment whose first window is not after it, and
// that is arithmetic.
//
// A wavefront does for its window what a lane of rlev2_regular_plan_kernel does, on bytes it must
// stage anyway: it stages the stream from 16 bytes before the interpolated position of its run to
// 18 bytes after the interpolated position of the next, looks in the first 32 positions for the
// two bytes that every 512-value header of the segment's kind and width starts with (lanes 0..31)
// and, in the 32 positions around the next run's, for the same (lanes 32..63).  One hit is the
// choice; the segment's first run is at the entry point and the run after its last at the next
// entry point, no search.  The header is parsed out of LDS, the run must hold 512 values and end
// exactly at the next window's choice -- which the next wavefront arrives at by the same arithmetic
// on the same bytes -- and so, by induction from the entry point, every position used is one a walk
// of the header chain reaches (rle_v2.rs:441-461, 480-514).  No hit or several (two bytes of
// payload can look like a header: about one window in two thousand): the wavefront walks the chain
// from the entry point, which proves its position by itself; it still checks its end against the
// next choice, because the next wavefront trusts that choice.  A next choice that is ambiguous
// needs no check: that wavefront walks.  Windows that are not plain -- the column's last, fields
// wider than 32 bits, PatchedBase, Delta without packed deltas -- go through the general
// single-run code.  Anything that does not hold raises *flag, and so do too many walks (a column
// whose runs are a few bytes long has several headers in every 32 positions: it is not for this
// path); the caller then decodes the stream another way.
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t lds_byte(const uint32_t *lds, uint32_t idx) {
    return (lds[idx >> 2] >> (24u - 8u * (idx & 3u))) & 0xFFu;  // (the words are byte-swapped)
}

// The two bytes at idx, idx + 1 of the slice, first byte in the high half.
__device__ __forceinline__ uint32_t lds_be16(const uint32_t *lds, uint32_t idx) {
    const uint32_t d = idx >> 2;
    const uint64_t x = ((uint64_t)lds[d] << 32) | lds[d + 1];
    return (uint32_t)((x << (8u * (idx & 3u))) >> 48);
}

// The seven bits of each of the eight bytes of x, lowest byte first, packed into 56 bits (the value
// of a varint of eight bytes whose continuation bits are ignored).
__device__ __forceinline__ uint64_t compress7(uint64_t x) {
    x = ((x & 0x7F007F007F007F00ull) >> 1) | (x & 0x007F007F007F007Full);
    x = ((x & 0x3FFF00003FFF0000ull) >> 2) | (x & 0x00003FFF00003FFFull);
    x = ((x & 0x0FFFFFFF00000000ull) >> 4) | (x & 0x000000000FFFFFFFull);
    return x;
}

constexpr uint32_t SELFPLAN_REACH = 16;   // positions searched on either side of the guess
constexpr uint32_t SELFPLAN_MAX_RUN = 4200;  // bytes: no run of 512 values is longer (4 + 8 + 4096 + 62)
constexpr uint32_t STEP_FRAC_BITS = 20;

// What the wavefronts of a segment share, worked out once per segment by a lane of
// rlev2_selfplan_segments_kernel so that a wavefront gets it in one load: where the segment's
// first run is, how far apart its runs are expected (the distance to the next entry point over the
// runs in between, in 2^-20ths of a byte; the last segment borrows the one before's), which
// windows it holds and what its first header byte is.
struct SegPlan {
    uint64_t e0;
    uint64_t step;
    uint32_t first_win;
    uint32_t n_win;
    uint32_t h0;
    uint32_t pad;
};

// first window of segment q when entry q is the header of the run that holds value q * S
__host__ __device__ inline uint64_t seg_first_win(uint64_t q, uint64_t S) { return (q * S) >> WINDOW_SHIFT; }

// Segment table [n_seg + 1] (the last entry a sentinel at the end of the stream); anything that does
// not fit -- entry points out of order or out of the stream, a segment without a window, runs
// that would have to be longer than a run can be -- raises *flag.
__global__ __launch_bounds__(256) void rlev2_selfplan_segments_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const uint64_t *__restrict__ entries, uint32_t n_seg,
    uint32_t n_win, uint64_t S, SegPlan *__restrict__ table, uint32_t *__restrict__ flag) {
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg > n_seg) return;
    SegPlan p = {len, 0, n_win, 0, 0, 0};
    if (seg < n_seg) {
        const bool last = seg + 1 == n_seg;
        const uint64_t fw = seg_first_win(seg, S), fw_next = last ? (uint64_t)n_win : seg_first_win((uint64_t)seg + 1, S);
        const uint64_t e0 = seg ? entries[seg] : 0;
        uint64_t ea = e0, eb = last ? e0 : entries[seg + 1], runs = fw_next - fw;
        if (last) {  // no entry point follows: the spacing of the segment before
            ea = seg > 1 ? entries[seg - 1] : 0;
            runs = fw - seg_first_win(seg ? seg - 1 : 0, S);
        }
        const bool ok = fw < fw_next && fw_next <= n_win && e0 < len && eb > ea && eb <= len && runs > 0 &&
                        eb - ea <= runs * SELFPLAN_MAX_RUN && fw <= 0xFFFFFFFFull;
        if (!ok) {
            raise_flag(flag);
        } else {
            p.e0 = e0;
            p.step = ((eb - ea) << STEP_FRAC_BITS) / runs;
            p.first_win = (uint32_t)fw;
            p.n_win = (uint32_t)(fw_next - fw);
            p.h0 = s[e0];
            // PatchedBase runs are not for the expansion behind this table (it would decode each
            // header with the general parser, patches and all, in every wavefront's critical
            // path): the plan as a pass of its own takes such a column a third faster
            if ((p.h0 >> 6) == KIND_PATCHED_BASE) raise_flag(flag);
        }
    }
    table[seg] = p;
}

__global__ __launch_bounds__(VAL_WAVES * 64, 8) void rlev2_selfplan_values_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const SegPlan *__restrict__ table, uint32_t n_seg,
    uint32_t n_win, uint64_t seg_magic, uint64_t num_values, int is_signed,
    int64_t *__restrict__ out, uint32_t *__restrict__ flag, uint32_t stream_once) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[VAL_WAVES][STAGE_WORDS];
    __shared__ uint32_t patch_tab[VAL_WAVES][PATCH_TAB_WORDS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t wg = blockIdx.x;  // XCD-contiguous remap, as in rlev2_values_kernel
    {
        const uint32_t nwg = gridDim.x, q = nwg / 8, r = nwg % 8, x = wg % 8;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + wg / 8;
    }
    const uint32_t g = wg * VAL_WAVES + wave;
    if (g >= n_win) return;  // whole wavefront; no workgroup barrier below
    // somebody found the column irregular: the rest of this launch is wasted work (read through the
    // cache: it only saves time; looked at further down, when the loads behind it are on their way)
    const uint32_t flag_now = *flag;
    WaveOf<MODE_VALUES> wv;
    wv.s = s;
    wv.lds = stage[wave];
    wv.marks = nullptr;
    wv.patch_tab = patch_tab[wave];
    wv.out = out;
    wv.num_values = num_values;
    wv.len = len;
    wv.lane = threadIdx.x & 63;
    wv.is_signed = is_signed;
    wv.out_aligned = true;  // the launcher insists
    wv.acc = 0;
    wv.saw_patched = 0;
    wv.stream_once = stream_once != 0;
    const uint32_t lane = (uint32_t)wv.lane;
    auto fail = [&]() {
        if (lane == 0) raise_flag(flag);
    };
    // ---- which segment, which run of it ------------------------------------------------------
    // floor(((g + 1) * 512 - 1) / S) by a multiplication with seg_magic = 2^64 / S rounded up: the
    // last segment that starts at or before the window (exact for any S below 2^23; the table's
    // window ranges are checked all the same)
    uint32_t seg = (uint32_t)__umul64hi((((uint64_t)g + 1) << WINDOW_SHIFT) - 1, seg_magic);
    if (seg >= n_seg) seg = n_seg - 1;
    SegPlan sp = table[seg], sp_next = table[seg + 1];
    for (int hops = 0; hops < 2 && g < sp.first_win && seg > 0; hops++) {
        seg--;
        sp_next = sp;
        sp = table[seg];
    }
    for (int hops = 0; hops < 2 && g - sp.first_win >= sp.n_win && seg + 1 < n_seg; hops++) {
        seg++;
        sp = sp_next;
        sp_next = table[seg + 1];
    }
    if (g < sp.first_win || g - sp.first_win >= sp.n_win) return fail();
    const bool last_seg = seg + 1 == n_seg;
    const uint32_t k = g - sp.first_win, R = sp.n_win;
    const uint64_t e0 = readfirstlane64(sp.e0), eb = readfirstlane64(sp_next.e0), step = readfirstlane64(sp.step);
    const bool is_last_win = g + 1 == n_win;
    const bool next_is_entry = !last_seg && k + 1 == R;  // the next run is the next segment's first
    // own: positions A .. A + 31 are searched (k == 0: the entry point, no search); run j of the
    // segment is expected at e0 + j * step, a function of the table alone and so the same for the
    // wavefront of the window and for that of the window before
    uint64_t A = e0;
    if (k != 0) {
        const uint64_t mid = e0 + (((uint64_t)k * step) >> STEP_FRAC_BITS);
        A = mid > e0 + SELFPLAN_REACH ? mid - SELFPLAN_REACH : e0 + 1;
    }
    // next: positions B .. B + 31
    uint64_t B = next_is_entry ? eb : 0;
    if (!next_is_entry && !is_last_win) {
        const uint64_t mid = e0 + ((((uint64_t)k + 1) * step) >> STEP_FRAC_BITS);
        B = mid > e0 + SELFPLAN_REACH ? mid - SELFPLAN_REACH : e0 + 1;
    }
    A = readfirstlane64(A);
    B = readfirstlane64(B);
    const uint32_t h0 = __builtin_amdgcn_readfirstlane(sp.h0);
    // ---- stage [A, hi) ------------------------------------------------------------------------
    uint64_t hi = is_last_win ? A + 64 : (next_is_entry ? B + 2 : B + 2 * SELFPLAN_REACH + 2);
    if (hi > len) hi = len;
    if (A >= len || hi <= A || hi - A > STAGE_BYTES - 16) return fail();
    const uint64_t origin = stage_bytes(wv, A, hi);
    if (__builtin_amdgcn_readfirstlane(flag_now)) return;
    const uint32_t oA = (uint32_t)(A - origin), staged_end = (uint32_t)(hi - origin);  // LDS byte indices
    // ---- the two searches, 32 lanes each --------------------------------------------------------
    uint64_t own = k == 0 ? e0 : ~0ull, next = next_is_entry ? eb : ~0ull;  // ~0: no choice
    bool next_wanted = !is_last_win;
    {
        const uint32_t at = ((lane >> 5) ? (uint32_t)(B - origin) : oA) + (lane & 31u);
        const bool inside = at + 1 < staged_end && (lane >> 5 ? (next_wanted && !next_is_entry) : k != 0);
        const uint32_t two = inside ? lds_be16(wv.lds, at) : 0u;
        const uint64_t hits = __ballot(inside && two == ((h0 << 8) | 0xFFu));
        const uint32_t mine = (uint32_t)hits, theirs = (uint32_t)(hits >> 32);
        if (k != 0 && __builtin_popcount(mine) == 1) own = A + (uint32_t)__builtin_ctz(mine);
        if (next_wanted && !next_is_entry && __builtin_popcount(theirs) == 1) next = B + (uint32_t)__builtin_ctz(theirs);
    }
    const uint64_t wbase = (uint64_t)g << WINDOW_SHIFT;
    // ---- the plain window: header out of LDS ----------------------------------------------------
    bool plain = own != ~0ull && !is_last_win && wbase + WINDOW <= num_values;
    uint32_t kind = 0, w = 0, header_bytes = 2;
    uint64_t base = 0, dbase = 0;
    if (plain) {
        // the 24 bytes from the header on as scalars, lowest address in the lowest byte: seven
        // lanes fetch a word each, and everything after that is scalar arithmetic
        const uint32_t o = (uint32_t)(own - origin);
        const uint32_t wi = (o >> 2) + (lane < 7 ? lane : 0u);
        const uint32_t dw = __builtin_bswap32(wv.lds[wi < STAGE_WORDS ? wi : 0u]);
        const uint64_t q0 = ((uint64_t)readlane32(dw, 1) << 32) | readlane32(dw, 0);
        const uint64_t q1 = ((uint64_t)readlane32(dw, 3) << 32) | readlane32(dw, 2);
        const uint64_t q2 = ((uint64_t)readlane32(dw, 5) << 32) | readlane32(dw, 4);
        const uint64_t q3 = readlane32(dw, 6);
        // (a shift by 64 - sh in two steps, so that sh = 0 needs no case of its own)
        const uint32_t sh = 8u * (o & 3u);
        const uint64_t x0 = (q0 >> sh) | ((q1 << 1) << (63 - sh));
        const uint64_t x1 = (q1 >> sh) | ((q2 << 1) << (63 - sh));
        const uint64_t x2 = (q2 >> sh) | ((q3 << 1) << (63 - sh));
        const uint32_t b0 = (uint32_t)x0 & 0xFFu, b1 = (uint32_t)(x0 >> 8) & 0xFFu;
        kind = b0 >> 6;
        const uint32_t code = (b0 >> 1) & 31u;
        plain = (b0 & 1u) && b1 == 0xFFu && (kind == KIND_DIRECT || (kind == KIND_DELTA && code != 0));  // 512 values
        w = width_from_code(code);
        plain = plain && w <= 32 && o + 24 <= staged_end;
        if (plain && kind == KIND_DELTA) {
            // the two varints (rle_v2.rs:71-89), of at most eight bytes each here: a longer one
            // sends the window to the general code
            const uint64_t tops = 0x8080808080808080ull;
            const uint64_t va = (x0 >> 16) | (x1 << 48);  // header bytes 2 .. 9
            const uint64_t vb = (x1 >> 16) | (x2 << 48);  // 10 .. 17
            const uint64_t e1 = ~va & tops;
            const uint32_t n1 = e1 ? ((uint32_t)__builtin_ctzll(e1) >> 3) + 1u : 9u;
            const uint32_t m1 = n1 <= 8 ? n1 : 8u;  // (shift counts stay in range when there is no end)
            const uint64_t ya = ((va >> (8 * m1 - 1)) >> 1) | (vb << (64 - 8 * m1));
            const uint64_t e2 = ~ya & tops;
            const uint32_t n2 = e2 ? ((uint32_t)__builtin_ctzll(e2) >> 3) + 1u : 9u;
            plain = n1 <= 8 && n2 <= 8;
            if (plain) {
                const uint64_t v1 = compress7(va & (~0ull >> (64 - 8 * n1)));
                const uint64_t v2 = compress7(ya & (~0ull >> (64 - 8 * n2)));
                base = is_signed ? zigzag(v1) : v1;  // rle_v2.rs:350
                dbase = zigzag(v2);                    // rle_v2.rs:100-102, 351
                header_bytes = 2 + n1 + n2;
            }
        }
    }
    if (plain) {
        const uint64_t payload = own + header_bytes;
        const uint32_t fields = kind == KIND_DELTA ? WINDOW - 2 : WINDOW;
        const uint64_t run_end = payload + ((fields * w + 7) >> 3);
        if (run_end - origin > staged_end) plain = false;  // (not all staged: the general code looks again)
        else {
            if (next_wanted && next != ~0ull && run_end != next) return fail();  // the link
            const uint32_t bit0 = (uint32_t)(payload - origin) << 3;
            int64_t *wout = out + wbase;
            if (kind == KIND_DIRECT) {
                direct_window<true, false>(wv, wout, w, is_signed != 0, 0, bit0 + 2u * lane * w, 0, WINDOW);
            } else {
                const bool neg = (int64_t)dbase < 0;
                if (w <= 16) {
                    if (neg) delta_window_full<true, false, uint32_t, true>(wv, wout, w, base, dbase, bit0);
                    else delta_window_full<true, false, uint32_t, false>(wv, wout, w, base, dbase, bit0);
                } else if (w <= 23) {
                    if (neg) delta_window_full<false, false, uint32_t, true>(wv, wout, w, base, dbase, bit0);
                    else delta_window_full<false, false, uint32_t, false>(wv, wout, w, base, dbase, bit0);
                } else {
                    if (neg) delta_window_full<false, false, uint64_t, true>(wv, wout, w, base, dbase, bit0);
                    else delta_window_full<false, false, uint64_t, false>(wv, wout, w, base, dbase, bit0);
                }
            }
            return;
        }
    }
    // ---- everything else: position by the search or by a walk of the chain, then the general code --
    uint64_t pos = own;
    if (pos == ~0ull) {
        // (counted: a column where this is the rule is not for this path)
        if (lane == 0 && atomicAdd(flag + 2, 1u) > (n_win >> 5) + 64u) raise_flag(flag);
        pos = e0;
        for (uint32_t j = 0; j < k; j++) {
            ParsedRun hop;
            if (pos >= len || parse_run(load_window(s, len, pos), s, len, pos, is_signed, hop) != SEG_OK ||
                hop.len != WINDOW)
                return fail();
            pos = readfirstlane64(hop.run_end);
        }
    }
    ParsedRun r;
    if (pos >= len || parse_run(load_window(s, len, pos), s, len, pos, is_signed, r) != SEG_OK) return fail();
    if (is_last_win) {
        // the column's last run: any length that holds the values still wanted
        if (wbase >= num_values || (uint64_t)r.len < num_values - wbase) return fail();
    } else {
        if (r.len != WINDOW) return fail();
        if (next != ~0ull && readfirstlane64(r.run_end) != next) return fail();  // the link
    }
    const uint64_t desc = make_desc(r.payload_off, r.kind, r.width, r.len, r.delta_neg, r.header_bytes);
    decode_single_run(wv, readfirstlane64(desc), wbase, readfirstlane64(r.base), readfirstlane64(r.aux));
}

hipError_t launch_rlev2_selfplan_values(hipStream_t st, const uint8_t *stream, uint64_t len,
                                        const uint64_t *entries, uint32_t n_seg, uint32_t n_win,
                                        uint32_t stride_vals, uint64_t num_values, int is_signed, int64_t *out,
                                        uint32_t *flag, void *seg_table) {
    if (!entries || !seg_table || (((uintptr_t)out) & 15) || !rlev2_selfplan_shape(n_seg, n_win, stride_vals))
        return hipErrorInvalidValue;
    uint64_t S = stride_vals;
    if (S == 0) {  // equally many runs in every segment: the counts leave one possibility
        uint32_t r_lo = 0, r_hi = 0;
        (void)rlev2_regular_plan_shape(n_seg, n_win, &r_lo, &r_hi);
        S = (uint64_t)r_lo << WINDOW_SHIFT;
    }
    const uint64_t magic = ~0ull / S + 1;  // 2^64 / S rounded up (S >= 512 is no divisor that gives 2^64)
    rlev2_selfplan_segments_kernel<<<(n_seg + 1 + 255) / 256, 256, 0, st>>>(stream, len, entries, n_seg, n_win, S,
                                                                            (SegPlan *)seg_table, flag);
    const uint32_t blocks = (n_win + VAL_WAVES - 1) / VAL_WAVES;
    // A stream that the last-level cache (256 MB) cannot hold anyway is loaded with the
    // nontemporal hint: 1.67 -> 1.62 ms on 10^9 values of 10-bit deltas (1.27 GB), while the 130 MB
    // of 10^8 timestamps took 0.206 instead of 0.185 ms with it.  (The hint costs the strided and
    // the planned kernel 2-3 % at every size: their neighbouring windows share more lines.)
    const uint32_t stream_once = len > SELFPLAN_STREAMING_BYTES;
    rlev2_selfplan_values_kernel<<<blocks, VAL_WAVES * 64, 0, st>>>(stream, len, (const SegPlan *)seg_table, n_seg, n_win,
                                                                   magic, num_values, is_signed, out, flag, stream_once);
    return hipGetLastError();
}

size_t rlev2_selfplan_table_bytes(uint32_t n_seg) { return ((size_t)n_seg + 1) * sizeof(SegPlan); }

bool rlev2_selfplan_shape(uint32_t n_seg, uint32_t n_win, uint32_t stride_vals) {
    if (n_seg < 2 || n_win == 0) return false;
    if (stride_vals) return stride_vals >= WINDOW && stride_vals < (1u << 23);
    uint32_t r_lo = 0, r_hi = 0;
    return rlev2_regular_plan_shape(n_seg, n_win, &r_lo, &r_hi) && r_lo == r_hi && r_lo < (1u << 14);
}


hipError_t launch_rlev2_regular_plan(hipStream_t st, const uint8_t *stream, uint64_t len,
                                     const uint64_t *entries, uint32_t n_seg, uint32_t n_win,
                                     uint64_t num_values, int is_signed, RunEntry *group_run,
                                     uint32_t *flag) {
    uint32_t r_lo, r_hi;
    if (!entries || !rlev2_regular_plan_shape(n_seg, n_win, &r_lo, &r_hi)) return hipErrorInvalidValue;
    const uint64_t waves = ((uint64_t)n_win + 62) / 63;
    rlev2_regular_plan_kernel<<<(uint32_t)((waves + 3) / 4), 256, 0, st>>>(
        stream, len, entries, n_seg, n_win, r_lo, r_hi, num_values, is_signed, group_run, flag);
    return hipGetLastError();
}

hipError_t launch_spec_publish(hipStream_t st, uint32_t *flag, ScanTotals *host_totals, uint32_t stamp) {
    spec_publish_kernel<<<1, 64, 0, st>>>(flag, host_totals, stamp);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------

hipError_t launch_rlev2_count(hipStream_t st, const uint8_t *stream, uint64_t len,
                              const uint64_t *entries, const uint64_t *seg_ends, uint32_t n_seg,
                              int is_signed, const SegTable &t, uint64_t *staged, uint32_t staged_cap) {
    // Many small segments: a wavefront stages the bytes of 64 of them in LDS, or of 32 where 64
    // would not fit with room to spare (see rlev2_count_staged_kernel).  One wavefront to a
    // workgroup: three slices fit a compute unit.
    const uint64_t avg = len / n_seg;
    if (n_seg >= 4096 && len < 0xFFFF0000ull) {
        uint32_t group = 0;
        for (uint32_t g : {64u, 32u})
            if (!group && avg * g * 11 / 10 + 64 <= COUNT_SLICE_BYTES) group = g;
        if (group) {
            const uint64_t n_waves = ((uint64_t)n_seg + group - 1) / group;
            rlev2_count_staged_kernel<<<(uint32_t)n_waves, 64, COUNT_SLICE_BYTES + 32, st>>>(
                stream, len, entries, seg_ends, n_seg, is_signed, t, staged, staged_cap, group, COUNT_SLICE_BYTES);
            rlev2_publish_kernel<<<(n_seg + SEG_BLOCK - 1) / SEG_BLOCK, SEG_BLOCK, 0, st>>>(n_seg, t);
            return hipGetLastError();
        }
    }
    const uint32_t blocks = (n_seg + SEG_BLOCK - 1) / SEG_BLOCK;
    rlev2_scan_kernel<false><<<blocks, SEG_BLOCK, 0, st>>>(stream, len, entries, seg_ends, n_seg,
                                                          is_signed, t, nullptr, nullptr, nullptr, 0, 0,
                                                          staged, staged_cap, 0);
    return hipGetLastError();
}

hipError_t launch_rlev2_coop_count(hipStream_t st, const uint8_t *stream, uint64_t len,
                                   const uint64_t *entries, const uint64_t *seg_ends, uint32_t n_seg,
                                   int is_signed, const SegTable &t, uint64_t *staged, uint32_t staged_cap) {
    // the LDS slice: room for a segment twice the average (whatever is longer is parsed from memory),
    // and no more than lets five wavefronts share a compute unit
    const uint64_t want = (2 * (len / n_seg) + 64 + 4095) / 4096 * 4096;
    const uint32_t slice_bytes = (uint32_t)(want < 8192 ? 8192 : want > 32768 ? 32768 : want);
    rlev2_coop_count_kernel<<<n_seg, 64, COOP_RING * sizeof(uint64_t) + slice_bytes + 32, st>>>(
        stream, len, entries, seg_ends, n_seg, is_signed, t, staged, staged_cap, slice_bytes);
    rlev2_publish_kernel<<<(n_seg + SEG_BLOCK - 1) / SEG_BLOCK, SEG_BLOCK, 0, st>>>(n_seg, t);
    return hipGetLastError();
}

hipError_t launch_block_totals_scan(hipStream_t st, const SegTable &t, uint32_t n_blocks,
                                    uint64_t num_values) {
    block_totals_scan_kernel<<<1, SCAN_THREADS, 0, st>>>(t, n_blocks, num_values);
    return hipGetLastError();
}

hipError_t launch_rlev2_emit(hipStream_t st, const uint8_t *stream, uint64_t len,
                             const uint64_t *entries, const uint64_t *seg_ends, uint32_t n_seg,
                             int is_signed, const SegTable &t, uint64_t *runs, uint32_t *group_first,
                             RunEntry *group_run, uint32_t n_group_slots, bool full_table,
                             uint32_t staged_up_to) {
    const uint32_t blocks = (n_seg + SEG_BLOCK - 1) / SEG_BLOCK;
    rlev2_scan_kernel<true><<<blocks, SEG_BLOCK, 0, st>>>(stream, len, entries, seg_ends, n_seg,
                                                         is_signed, t, runs, group_first, group_run,
                                                         n_group_slots, full_table ? 1 : 0, nullptr,
                                                         0, staged_up_to);
    return hipGetLastError();
}

hipError_t launch_rlev2_emit_staged(hipStream_t st, const uint8_t *stream, uint64_t len,
                                    const uint64_t *entries, const uint64_t *seg_ends, uint32_t n_seg,
                                    int is_signed, const SegTable &t, const uint64_t *staged,
                                    uint32_t staged_cap, uint64_t *runs, uint32_t *group_first,
                                    RunEntry *group_run, uint32_t n_group_slots, bool full_table,
                                    uint32_t max_seg_runs) {
    const uint32_t group = staged_group_lanes(max_seg_runs);
    // two sets of 256 / group segments per workgroup once there are wavefronts to spare
    const uint32_t sets = (uint64_t)n_seg * group >= 64ull * 256 * 64 ? 2 : 1;
    const uint64_t blocks = ((uint64_t)n_seg * group + 256 * sets - 1) / (256 * sets);
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    rlev2_emit_staged_kernel<<<(uint32_t)blocks, 256, 0, st>>>(stream, len, entries, seg_ends, n_seg, is_signed, t,
                                                     staged, staged_cap, runs, group_first, group_run,
                                                     n_group_slots, full_table ? 1 : 0, group, sets);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Entry points for a stream that comes without any.  The header chain is serial, but in a long
// column most runs have the same size (512 values of one width), and that can be checked in
// parallel: if the run at p + k * S parses and ends at p + (k + 1) * S for every k < j, and the
// header at p is real, then by induction all of them are real headers.
// ---------------------------------------------------------------------------------------------

// out[0] = size in bytes of the run whose header is at pos (0 if it does not parse)
__global__ void rlev2_probe_kernel(const uint8_t *__restrict__ s, uint64_t len, uint64_t pos,
                                   int is_signed, uint64_t *__restrict__ out) {
    if (blockIdx.x || threadIdx.x) return;
    ParsedRun r;
    uint64_t size = 0;
    if (pos < len && parse_run(load_window(s, len, pos), s, len, pos, is_signed, r) == SEG_OK)
        size = r.run_end - pos;
    out[0] = size;
}

// *first_fail = smallest k < n for which the bytes at p + k * stride are not a run of `stride` bytes
__global__ __launch_bounds__(256) void rlev2_verify_stride_kernel(
    const uint8_t *__restrict__ s, uint64_t len, uint64_t p, uint64_t stride, uint64_t n,
    int is_signed, unsigned long long *__restrict__ first_fail) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint64_t pos = p + k * stride;
    ParsedRun r;
    const bool ok = pos < len && parse_run(load_window(s, len, pos), s, len, pos, is_signed, r) == SEG_OK &&
                    r.run_end - pos == stride;
    if (!ok) atomicMin(first_fail, (unsigned long long)k);
}

hipError_t launch_rlev2_probe(hipStream_t st, const uint8_t *stream, uint64_t len, uint64_t pos,
                              int is_signed, uint64_t *out) {
    rlev2_probe_kernel<<<1, 64, 0, st>>>(stream, len, pos, is_signed, out);
    return hipGetLastError();
}

hipError_t launch_rlev2_verify_stride(hipStream_t st, const uint8_t *stream, uint64_t len, uint64_t p,
                                      uint64_t stride, uint64_t n, int is_signed, uint64_t *first_fail) {
    if (n == 0) return hipSuccess;
    rlev2_verify_stride_kernel<<<(uint32_t)((n + 255) / 256), 256, 0, st>>>(
        stream, len, p, stride, n, is_signed, (unsigned long long *)first_fail);
    return hipGetLastError();
}

// out[i] = number of values before segment first_seg[i] (after the scan of the counts)
__global__ void seg_value_offsets_kernel(SegTable t, const uint32_t *__restrict__ first_seg, uint32_t n,
                                         uint64_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = seg_global_offset(t, first_seg[i]).val_off;
}

// Whether the values before segment first_seg[k] are want_off[k] of them, for every k < n: the
// first k for which they are not (0xFFFFFFFF: none) goes to host_word[0] -- host memory the device
// can write -- and `stamp` to host_word[1] after it, for a host that watches for the stamp.
__global__ __launch_bounds__(256) void seg_offsets_verify_kernel(SegTable t, const uint32_t *__restrict__ first_seg,
                                                                 const uint64_t *__restrict__ want_off, uint32_t n,
                                                                 uint64_t *host_word, uint32_t stamp) {
    __shared__ uint32_t first_bad;
    if (threadIdx.x == 0) first_bad = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n; k += 256)
        if (seg_global_offset(t, first_seg[k]).val_off != want_off[k]) atomicMin(&first_bad, k);
    __syncthreads();
    if (threadIdx.x == 0) {
        volatile uint64_t *h = host_word;
        h[0] = first_bad;
        __threadfence_system();
        h[1] = stamp;
    }
}

hipError_t launch_seg_offsets_verify(hipStream_t st, const SegTable &t, const uint32_t *first_seg,
                                     const uint64_t *want_off, uint32_t n, uint64_t *host_word, uint32_t stamp) {
    seg_offsets_verify_kernel<<<1, 256, 0, st>>>(t, first_seg, want_off, n, host_word, stamp);
    return hipGetLastError();
}

hipError_t launch_seg_value_offsets(hipStream_t st, const SegTable &t, const uint32_t *first_seg,
                                    uint32_t n, uint64_t *out) {
    if (n == 0) return hipSuccess;
    seg_value_offsets_kernel<<<(n + 255) / 256, 256, 0, st>>>(t, first_seg, n, out);
    return hipGetLastError();
}

bool rlev2_values_short_variant(const Rlev2Table &t) { return (uint64_t)t.n_runs > 2ull * t.n_groups; }

template <int MODE>
static hipError_t launch_values_mode(hipStream_t st, const Rlev2Table &t, int64_t *out,
                                     uint64_t *group_sums, uint32_t *flags) {
    if (t.n_groups == 0) return hipSuccess;
    const uint32_t blocks = (t.n_groups + VAL_WAVES - 1) / VAL_WAVES;
    // more than two runs per window on average: the variant for short runs (which, when it
    // writes values, gives a wavefront two windows)
    const uint32_t pairs = (t.n_groups + 1) / 2;
    const uint32_t short_blocks = MODE == MODE_VALUES ? (pairs + VAL_WAVES - 1) / VAL_WAVES : blocks;
    if (rlev2_values_short_variant(t))
        rlev2_values_kernel<MODE, true><<<short_blocks, VAL_WAVES * 64, 0, st>>>(
            t.stream, t.len, t.runs, t.group_first, t.group_run, t.n_groups, t.n_runs, t.num_values,
            t.is_signed, out, group_sums, flags);
    else
        rlev2_values_kernel<MODE, false><<<blocks, VAL_WAVES * 64, 0, st>>>(
            t.stream, t.len, t.runs, t.group_first, t.group_run, t.n_groups, t.n_runs, t.num_values,
            t.is_signed, out, group_sums, flags);
    return hipGetLastError();
}

hipError_t launch_rlev2_resync(hipStream_t st, const uint8_t *stream, uint64_t len, int is_signed,
                               uint64_t spacing, uint32_t n_entries, uint64_t *entries) {
    if (n_entries == 0) return hipSuccess;
    rlev2_resync_kernel<<<(n_entries + 3) / 4, 256, 0, st>>>(stream, len, is_signed, spacing, n_entries, entries);
    return hipGetLastError();
}

hipError_t launch_rlev2_planned_values(hipStream_t st, const Rlev2Table &t, int64_t *out,
                                       const uint32_t *abort_flag) {
    if (t.n_groups == 0) return hipSuccess;
    const uint32_t blocks = (t.n_groups + VAL_WAVES - 1) / VAL_WAVES;
    rlev2_planned_values_kernel<<<blocks, VAL_WAVES * 64, 0, st>>>(t.stream, t.len, t.group_run, t.n_groups,
                                                                  t.num_values, t.is_signed, out, abort_flag);
    return hipGetLastError();
}

hipError_t launch_rlev2_values(hipStream_t st, const Rlev2Table &t, int64_t *out,
                               const uint32_t *abort_flag) {
    return launch_values_mode<MODE_VALUES>(st, t, out, nullptr, const_cast<uint32_t *>(abort_flag));
}

hipError_t launch_rlev2_group_sums(hipStream_t st, const Rlev2Table &t, uint64_t *group_sums,
                                   uint32_t *flags) {
    // tables of long runs: four windows to a wavefront (see rlev2_quad_sums_kernel);
    // ORCD_WAVE_SUMS=1 in the environment: a wavefront per window always (for comparisons)
    static const bool wave_sums = [] {
        const char *e = getenv("ORCD_WAVE_SUMS");
        return e && *e && *e != '0';
    }();
    if (!wave_sums && t.n_groups && t.runs && !rlev2_values_short_variant(t)) {
        const uint64_t waves = ((uint64_t)t.n_groups + SUMS_WINDOWS - 1) / SUMS_WINDOWS;
        rlev2_quad_sums_kernel<<<(uint32_t)((waves + VAL_WAVES - 1) / VAL_WAVES), VAL_WAVES * 64, 0, st>>>(
            t.stream, t.len, t.runs, t.group_run, t.n_groups, t.n_runs, t.num_values, t.is_signed, group_sums, flags);
        return hipGetLastError();
    }
    return launch_values_mode<MODE_SUMS>(st, t, nullptr, group_sums, flags);
}

hipError_t launch_rlev2_offsets(hipStream_t st, const Rlev2Table &t, const uint64_t *group_prefix,
                                int64_t *offsets) {
    return launch_values_mode<MODE_OFFSETS>(st, t, offsets, const_cast<uint64_t *>(group_prefix), nullptr);
}

// ---------------------------------------------------------------------------------------------
// Integer RLE version 1: the DATA and LENGTH streams of files written as ORC 0.11 (ColumnEncoding
// DIRECT / DICTIONARY).  Not in the reference (README.md:23-25 lists it as missing); the layout is
// the ORC v1 specification's: a control byte 0..127 starts a run of control + 3 values -- a signed
// delta byte and the base as a varint, value i = base + i * delta -- and a control byte 128..255
// starts 256 - control literal values, a varint each (zigzag for signed streams).
//
// A wavefront walks a segment (the groups between two entry points) twice: once to count its
// values, then -- after the same prefix sums the other decoders use -- to write them where they
// belong.  The stream passes through LDS in tiles of V1_TILE bytes; a group is decoded only when
// all of it lies in the tile.  Inside a literal group the wavefront looks at 64 bytes at a time,
// a lane per byte: a byte without the continuation bit ends a varint, so one ballot gives every
// varint's start, length and index (the number of ends in front of it), and the lane on a start
// byte assembles and stores that value.  A step takes the bytes up to the last end it sees, or up
// to the group's last value.
// ---------------------------------------------------------------------------------------------

constexpr uint32_t V1_TILE = 4096;  // bytes per wavefront in LDS (a literal group is at most 1281)
constexpr uint32_t V1_WAVES = 4;    // wavefronts (= segments) per workgroup

template <bool EMIT>
__global__ __launch_bounds__(V1_WAVES * 64) void rlev1_walk_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const uint64_t *__restrict__ entries, uint32_t n_seg,
    int is_signed, SegTable t, uint64_t num_values, int64_t *__restrict__ out) {
    __shared__ __align__(16) uint8_t tiles[V1_WAVES][V1_TILE];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t seg = blockIdx.x * V1_WAVES + wave;
    if (seg >= n_seg) return;  // (no workgroup barrier below: a wavefront may leave alone)
    uint8_t *tile = tiles[wave];
    uint64_t pos = (entries && seg > 0) ? entries[seg] : 0;
    uint64_t end = (entries && seg + 1 < n_seg) ? entries[seg + 1] : len;
    const bool last_seg = seg + 1 == n_seg;
    uint32_t n_groups = 0, last_count = 0, status = SEG_OK;
    uint64_t n_vals = 0, out_off = 0;
    if (pos > len) pos = len;
    if (end > len || end < pos) status = SEG_MISALIGNED;
    if (EMIT) {
        out_off = seg_global_offset(t, seg).val_off;
        if (out_off >= num_values) return;  // nothing of this segment is wanted
    }

    // the tile holds the bytes at addresses [tile_at, tile_at + V1_TILE), tile_at a multiple of 16
    const uintptr_t s_addr = (uintptr_t)s, lim_addr = s_addr + len;
    uintptr_t tile_at = 0;
    bool loaded = false;
    auto fill = [&](uint64_t p) {
        tile_at = (s_addr + p) & ~(uintptr_t)15;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (uint32_t k = 0; k < V1_TILE / 1024; k++) {
            const uintptr_t a = tile_at + (uintptr_t)(k * 64 + lane) * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (a < lim_addr) v = *(const uint4 *)a;  // (the 16 bytes around the stream's ends are read whole)
            ((uint4 *)tile)[k * 64 + lane] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        loaded = true;
    };
    // the n bytes at p (p < len), as far as the stream goes, are in the tile afterwards
    auto ensure = [&](uint64_t p, uint32_t n) {
        const uint64_t last = p + n < len ? p + n : len;
        if (!loaded || s_addr + last > tile_at + V1_TILE) fill(p);
    };
    auto in_tile = [&](uint64_t p) { return p < len && s_addr + p < tile_at + V1_TILE; };
    auto at = [&](uint64_t p) -> uint32_t { return tile[(s_addr + p) - tile_at]; };

    if (status == SEG_OK) {
        while (pos < end) {
            ensure(pos, 12);
            const uint32_t control = (uint32_t)__builtin_amdgcn_readfirstlane((int)at(pos));
            if (control < 128) {  // a run: delta byte, base varint (at most 10 bytes, rle_v2.rs:71-89)
                const uint32_t count = control + 3;
                if (pos + 2 >= len) {
                    status = SEG_TRUNCATED;
                    break;
                }
                const uint64_t delta = (uint64_t)(int64_t)(int8_t)at(pos + 1);
                const uint64_t q = pos + 2 + lane;
                const bool in = lane < 10 && q < len;
                const uint32_t b = in ? at(q) : 0x80u;
                const uint64_t ends = __ballot(in && !(b & 0x80));
                if (!ends) {
                    status = SEG_BAD_RUN;
                    break;
                }
                const uint32_t u = (uint32_t)__builtin_ctzll(ends);
                if (EMIT) {
                    uint64_t base = 0;
                    for (uint32_t j = 0; j <= u; j++) base |= (uint64_t)(readlane32(b, (int)j) & 0x7F) << (7 * j);
                    if (is_signed) base = zigzag(base);
                    for (uint32_t i = lane; i < count; i += 64) {
                        const uint64_t idx = out_off + n_vals + i;
                        if (idx < num_values) out[idx] = (int64_t)(base + (uint64_t)i * delta);
                    }
                }
                pos += 3 + u;
                n_vals += count;
                last_count = count;
            } else {  // literals: a varint each
                const uint32_t count = 256 - control;
                ensure(pos, 1 + 10 * count);
                uint64_t p = pos + 1;
                uint32_t done = 0;
                while (done < count) {
                    const uint64_t q = p + lane;
                    const bool in = in_tile(q);
                    const uint32_t b = in ? at(q) : 0x80u;
                    const uint64_t ends = __ballot(in && !(b & 0x80));
                    if (!ends) {  // 64 bytes and no varint ends, or the stream does
                        status = p + 10 > len ? SEG_TRUNCATED : SEG_BAD_RUN;
                        break;
                    }
                    const uint32_t rank = (uint32_t)__popcll(ends & ((1ull << lane) - 1));  // ends in front of this byte
                    const uint32_t seen = (uint32_t)__popcll(ends);
                    uint32_t take, bytes;
                    if (done + seen >= count) {  // the group's last value ends in this step
                        take = count - done;
                        const uint64_t hit = __ballot(((ends >> lane) & 1) && rank == take - 1);
                        bytes = (uint32_t)__builtin_ctzll(hit) + 1;
                    } else {
                        take = seen;
                        bytes = 64 - (uint32_t)__builtin_clzll(ends);
                    }
                    const bool starts = (lane == 0 || ((ends >> (lane - 1)) & 1)) && lane < bytes;
                    const uint32_t my_len = (uint32_t)__builtin_ctzll((ends >> lane) | (1ull << 63)) + 1;
                    if (__ballot(starts && my_len > 10)) {
                        status = SEG_BAD_RUN;
                        break;
                    }
                    if (EMIT && starts) {
                        uint64_t v = 0;
                        for (uint32_t j = 0; j < my_len; j++) v |= (uint64_t)(at(q + j) & 0x7F) << (7 * j);
                        const uint64_t idx = out_off + n_vals + done + rank;
                        if (idx < num_values) out[idx] = (int64_t)(is_signed ? zigzag(v) : v);
                    }
                    p += bytes;
                    done += take;
                }
                if (status != SEG_OK) break;
                pos = p;
                n_vals += count;
                last_count = count;
            }
            n_groups++;
            if (EMIT && out_off + n_vals >= num_values) break;  // the rest is not wanted
        }
    }
    if (EMIT) return;
    if (status == SEG_OK && pos != end && !last_seg) status = SEG_MISALIGNED;
    if (lane == 0) {
        SegCount c;
        c.n_runs = n_groups;
        c.last_len = last_count;
        c.status = status;
        c.pad = 0;
        c.n_vals = n_vals;
        t.seg_count[seg] = c;
    }
}

// The wavefronts' results, a lane per segment again, into the workgroup-local prefix sums.
__global__ __launch_bounds__(SEG_BLOCK) void rlev1_publish_kernel(SegTable t, uint32_t n_seg) {
    const uint32_t seg = blockIdx.x * SEG_BLOCK + threadIdx.x;
    SegCount c = {0, 0, SEG_OK, 0, 0};
    if (seg < n_seg) c = t.seg_count[seg];
    seg_publish(t, seg, n_seg, c.n_runs, c.last_len, c.status, c.n_vals);
}

hipError_t launch_rlev1_count(hipStream_t st, const uint8_t *stream, uint64_t len, const uint64_t *entries,
                              uint32_t n_seg, int is_signed, const SegTable &t) {
    rlev1_walk_kernel<false><<<(n_seg + V1_WAVES - 1) / V1_WAVES, V1_WAVES * 64, 0, st>>>(
        stream, len, entries, n_seg, is_signed, t, 0, nullptr);
    rlev1_publish_kernel<<<(n_seg + SEG_BLOCK - 1) / SEG_BLOCK, SEG_BLOCK, 0, st>>>(t, n_seg);
    return hipGetLastError();
}

hipError_t launch_rlev1_emit(hipStream_t st, const uint8_t *stream, uint64_t len, const uint64_t *entries,
                             uint32_t n_seg, int is_signed, const SegTable &t, uint64_t num_values, int64_t *out) {
    rlev1_walk_kernel<true><<<(n_seg + V1_WAVES - 1) / V1_WAVES, V1_WAVES * 64, 0, st>>>(
        stream, len, entries, n_seg, is_signed, t, num_values, out);
    return hipGetLastError();
}

}  // namespace orcd

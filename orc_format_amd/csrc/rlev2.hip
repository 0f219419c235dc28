// RLE v2 integer decode for gfx950: header scan, segment prefix sums, value expansion.
//
// The math is that of src/read/decode/rle_v2.rs of the orc-format crate (ShortRepeat :283-301,
// Direct :135-175, Delta :195-264/343-408, zigzag :92-98, unpack :105-123, varint :71-89) with the
// full bit-width table and PatchedBase of the ORC v1 specification.  The structure is not the
// crate's: run headers form a serial chain, so they are walked once per *segment* (the bytes
// between two caller-supplied entry points) by one lane each, and the values -- the bulk of the
// bytes -- are then expanded by one wavefront per aligned window of 512 output values.
#include "orcd_internal.h"

namespace orcd {

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------

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
    const uint64_t q0 = p[0];
    const uint64_t q1 = (a0 + 8 < end) ? p[1] : 0;
    const uint64_t q2 = (a0 + 16 < end) ? p[2] : 0;
    Window16 w;
    w.lo = sh ? (q0 >> sh) | (q1 << (64 - sh)) : q0;
    w.hi = sh ? (q1 >> sh) | (q2 << (64 - sh)) : q1;
    return w;
}

struct ParsedRun {
    uint64_t payload_off, run_end, base, aux;
    uint32_t kind, width, len, delta_neg;
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

// Parses the run whose header starts at `pos` (pos < len).  Returns a SEG_* status.
__device__ __forceinline__ uint32_t parse_run(const uint8_t *s, uint64_t len, uint64_t pos,
                                              int is_signed, ParsedRun &r) {
    const Window16 w = load_window(s, len, pos);
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
    if (r.run_end > len) return SEG_TRUNCATED;
    if (r.payload_off > PAYLOAD_MASK) return SEG_TOO_LONG;
    return SEG_OK;
}

// ---------------------------------------------------------------------------------------------
// K1: header scan -- one lane per segment.  COUNT pass: runs and values per segment.
// EMIT pass (after the segment prefix sums): the run table and the first run of every window.
// ---------------------------------------------------------------------------------------------

template <bool EMIT>
__global__ __launch_bounds__(256) void rlev2_scan_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const uint64_t *__restrict__ entries,
    uint32_t n_seg, int is_signed, SegCount *__restrict__ seg_count,
    const SegOffset *__restrict__ seg_off, RunEntry *__restrict__ runs,
    uint32_t *__restrict__ group_first, uint32_t n_group_slots) {
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= n_seg) return;
    // offset 0 is always an entry: segment 0 starts there whatever entries[0] says
    uint64_t pos = (entries && seg > 0) ? entries[seg] : 0;
    const uint64_t end = (entries && seg + 1 < n_seg) ? entries[seg + 1] : len;
    const bool last_seg = seg + 1 == n_seg;
    if (pos > len) pos = len;

    if (!EMIT) {
        uint32_t n_runs = 0, last_len = 0, status = SEG_OK;
        uint64_t n_vals = 0;
        if (end > len || end < pos) status = SEG_MISALIGNED;
        while (status == SEG_OK && pos < end) {
            ParsedRun r;
            status = parse_run(s, len, pos, is_signed, r);
            if (status != SEG_OK) break;
            n_runs++;
            n_vals += r.len;
            last_len = r.len;
            pos = r.run_end;
        }
        if (status == SEG_OK && pos != end && !last_seg) status = SEG_MISALIGNED;
        SegCount c;
        c.n_runs = n_runs;
        c.last_len = last_len;
        c.status = status;
        c.pad = 0;
        c.n_vals = n_vals;
        seg_count[seg] = c;
    } else {
        const uint32_t n_runs = seg_count[seg].n_runs;
        const SegOffset so = seg_off[seg];
        uint64_t out_off = so.val_off;
        uint32_t run_idx = so.run_off;
        uint32_t prev_len = so.prev_len;  // 0 only before the very first run
        for (uint32_t k = 0; k < n_runs; k++) {
            ParsedRun r;
            parse_run(s, len, pos, is_signed, r);  // status was SEG_OK in the counting pass
            RunEntry e;
            e.desc = make_desc(r.payload_off, r.kind, r.width, r.len, r.delta_neg);
            e.out_off = out_off;
            e.base = r.base;
            e.aux = r.aux;
            runs[run_idx] = e;
            // first run that starts inside its 512-value window?
            const uint64_t win = out_off >> WINDOW_SHIFT;
            const bool first = (out_off == 0) || (((out_off - prev_len) >> WINDOW_SHIFT) != win);
            if (first && win < n_group_slots) group_first[win] = run_idx;
            out_off += r.len;
            prev_len = r.len;
            run_idx++;
            pos = r.run_end;
        }
        if (last_seg) {
            // sentinel: one-past-the-end run and the window after the last value
            RunEntry e;
            e.desc = make_desc(pos, KIND_SHORT_REPEAT, 0, 1, 0);
            e.out_off = out_off;
            e.base = 0;
            e.aux = 0;
            runs[run_idx] = e;
            // Windows in which no run starts exist only at the tail: the last run may reach
            // into one more window.  They, and the slot one past the last window, get the
            // sentinel's index (an empty group).
            const uint64_t g_end = (out_off + WINDOW - 1) >> WINDOW_SHIFT;
            uint64_t g = out_off == 0 ? 0 : ((out_off - prev_len) >> WINDOW_SHIFT) + 1;
            for (; g <= g_end && g < n_group_slots; g++) group_first[g] = run_idx;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2: prefix sums over the segments (few: one per ~10k values) -- a single workgroup.
// ---------------------------------------------------------------------------------------------

constexpr int SCAN_THREADS = 1024;

template <typename T, typename Op>
__device__ __forceinline__ T block_inclusive_scan(T v, T identity, Op op, T *wave_tot /*[16]*/,
                                                  T &block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) {
        T up;
        if constexpr (sizeof(T) == 8) up = (T)shfl_up64((uint64_t)v, d);
        else up = (T)__shfl_up((uint32_t)v, d);
        if (lane >= d) v = op(up, v);
    }
    __syncthreads();
    if (lane == 63) wave_tot[wave] = v;
    __syncthreads();
    T prefix = identity, total = identity;
    for (int i = 0; i < SCAN_THREADS / 64; i++) {
        const T t = wave_tot[i];
        if (i < wave) prefix = op(prefix, t);
        total = op(total, t);
    }
    block_total = total;
    return op(prefix, v);
}

__global__ __launch_bounds__(SCAN_THREADS) void segment_scan_kernel(
    const SegCount *__restrict__ seg_count, uint32_t n_seg, uint64_t num_values,
    SegOffset *__restrict__ seg_off, ScanTotals *__restrict__ totals) {
    __shared__ uint64_t wave_tot[SCAN_THREADS / 64];
    __shared__ uint32_t first_bad;
    if (threadIdx.x == 0) first_bad = 0xFFFFFFFFu;
    __syncthreads();
    auto add = [](uint64_t a, uint64_t b) { return a + b; };
    auto mx = [](uint64_t a, uint64_t b) { return a > b ? a : b; };
    uint64_t carry_vals = 0, carry_runs = 0, carry_key = 0;
    for (uint32_t base = 0; base < n_seg; base += SCAN_THREADS) {
        const uint32_t i = base + threadIdx.x;
        SegCount c;
        c.n_runs = 0; c.last_len = 0; c.status = SEG_OK; c.pad = 0; c.n_vals = 0;
        if (i < n_seg) c = seg_count[i];
        // "length of the last run before segment i" = last_len of the most recent non-empty
        // segment j < i: a prefix maximum over keys (j + 1, last_len_j) of the elements shifted
        // right by one, so that the inclusive scan below is exclusive in i.
        uint64_t key = 0;
        if (i > 0 && i < n_seg) {
            const SegCount p = seg_count[i - 1];
            if (p.n_runs) key = ((uint64_t)i << 10) | p.last_len;
        }
        uint64_t tv, tr, tk;
        const uint64_t iv = block_inclusive_scan<uint64_t>(c.n_vals, 0, add, wave_tot, tv);
        const uint64_t ir = block_inclusive_scan<uint64_t>(c.n_runs, 0, add, wave_tot, tr);
        const uint64_t ik = block_inclusive_scan<uint64_t>(key, 0, mx, wave_tot, tk);
        if (i < n_seg) {
            SegOffset o;
            o.val_off = carry_vals + iv - c.n_vals;
            o.run_off = (uint32_t)(carry_runs + ir - c.n_runs);
            o.prev_len = (uint32_t)(mx(carry_key, ik) & 1023);
            seg_off[i] = o;
            // a stopped walk matters only if values past it are still wanted
            if (c.status != SEG_OK && o.val_off + c.n_vals < num_values) atomicMin(&first_bad, i);
        }
        carry_vals += tv;
        carry_runs += tr;
        carry_key = mx(carry_key, tk);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ScanTotals t;
        t.total_vals = carry_vals;
        t.total_runs = carry_runs;
        t.error = ORCD_OK;
        t.error_seg = first_bad;
        t.pad = 0;
        if (first_bad != 0xFFFFFFFFu) {
            t.error = seg_count[first_bad].status == SEG_TOO_LONG ? ORCD_UNSUPPORTED : ORCD_OUT_OF_SPEC;
        } else if (carry_vals < num_values) {
            t.error = ORCD_OUT_OF_SPEC;  // the stream ends before num_values values
        }
        *totals = t;
    }
}

// ---------------------------------------------------------------------------------------------
// K3: value expansion -- one wavefront per group (the runs that start inside one aligned window
// of 512 output values).  Lane i of step j owns output values 128*j + 2*i and 128*j + 2*i + 1 of
// the window, so every store instruction writes 1 KiB of consecutive, 16-byte aligned output.
// ---------------------------------------------------------------------------------------------

constexpr int VAL_WAVES = 4;                   // wavefronts per workgroup
constexpr uint32_t STAGE_BYTES = 2560;         // LDS staging slice per wavefront
constexpr uint32_t STAGE_WORDS = STAGE_BYTES / 4 + 4;  // + padding for the 3-word extraction

// Field of `w` bits (1..64) at bit `bitpos` of the big-endian bit stream formed by `words`.
// GUARD: touch only words that hold wanted bits (global memory: a word past them may be unmapped).
template <bool GUARD>
__device__ __forceinline__ uint64_t extract_bits(const uint32_t *words, uint64_t bitpos, uint32_t w) {
    const uint64_t d = bitpos >> 5;
    const uint32_t sh = (uint32_t)bitpos & 31;
    const uint32_t need = sh + w;
    const uint32_t w0 = __builtin_bswap32(words[d]);
    const uint32_t w1 = (!GUARD || need > 32) ? __builtin_bswap32(words[d + 1]) : 0;
    const uint32_t w2 = (!GUARD || need > 64) ? __builtin_bswap32(words[d + 2]) : 0;
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

struct RunFields {
    uint64_t desc, base, aux;
    uint32_t rel_out;  // out_off relative to the batch's first value
};

__global__ __launch_bounds__(VAL_WAVES * 64) void rlev2_values_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const RunEntry *__restrict__ runs,
    const uint32_t *__restrict__ group_first, uint32_t n_groups, uint32_t n_runs,
    uint64_t num_values, int is_signed, int64_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[VAL_WAVES][STAGE_WORDS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t g = blockIdx.x * VAL_WAVES + wave;
    if (g >= n_groups) return;  // whole wavefront leaves; no workgroup barrier is used below
    uint32_t *lds = stage[wave];
    const bool out_aligned = (((uintptr_t)out) & 15) == 0;

    // the group's runs; clamped to the table so that a bad index can never walk off it
    uint32_t r1 = __builtin_amdgcn_readfirstlane(group_first[g + 1]);
    if (r1 > n_runs) r1 = n_runs;
    uint32_t r0 = __builtin_amdgcn_readfirstlane(group_first[g]);
    if (r0 > r1) r0 = r1;

    for (uint32_t rb = r0; rb < r1; rb += 64) {
        const uint32_t nb = (r1 - rb) < 64u ? (r1 - rb) : 64u;
        // lane k holds run rb + k of the batch
        RunEntry me;
        me.desc = make_desc(0, KIND_SHORT_REPEAT, 0, 1, 0);
        me.out_off = ~0ull;
        me.base = 0;
        me.aux = 0;
        if ((uint32_t)lane < nb) me = runs[rb + lane];
        const uint32_t my_kind = desc_kind(me.desc), my_w = desc_width(me.desc);
        const uint32_t my_len = desc_len(me.desc);
        const uint64_t batch_beg = readlane64(me.out_off, 0);
        uint64_t batch_end = readlane64(me.out_off + my_len, (int)nb - 1);
        if (batch_end > num_values) batch_end = num_values;
        const uint32_t my_rel = (uint32_t)lane < nb ? (uint32_t)(me.out_off - batch_beg) : 0xFFFFFFFFu;
        uint64_t carry = 0;  // running value of a Delta run that continues into the next step

        for (uint64_t wbase = (batch_beg >> WINDOW_SHIFT) << WINDOW_SHIFT; wbase < batch_end;
             wbase += WINDOW) {
            const uint64_t lo = wbase > batch_beg ? wbase : batch_beg;
            const uint64_t hi = (wbase + WINDOW) < batch_end ? (wbase + WINDOW) : batch_end;

            // ---- which payload bytes does this window need? -------------------------------
            uint64_t need_b0 = 0, need_b1 = 0;
            bool needs = false, scans = false;
            if ((uint32_t)lane < nb && me.out_off < hi && me.out_off + my_len > lo &&
                my_kind != KIND_SHORT_REPEAT && my_w != 0) {
                const uint32_t first = me.out_off >= lo ? 0u : (uint32_t)(lo - me.out_off);
                const uint32_t last = (me.out_off + my_len) <= hi ? my_len : (uint32_t)(hi - me.out_off);
                uint64_t b0, b1;
                payload_range(my_kind, my_w, first, last, b0, b1);
                needs = b1 > b0;
                need_b0 = desc_payload(me.desc) + b0;
                need_b1 = desc_payload(me.desc) + b1;
                scans = my_kind == KIND_DELTA;
            }
            const uint64_t need_mask = __ballot(needs);
            const bool any_scan = __ballot(scans) != 0;
            bool use_lds = false;
            uint64_t lds_origin = 0;  // stream offset of LDS byte 0 (may precede the stream by <16)
            if (need_mask) {
                const int k_lo = __ffsll((unsigned long long)need_mask) - 1;
                const int k_hi = 63 - __clzll((unsigned long long)need_mask);
                const uint64_t span_b0 = readlane64(need_b0, k_lo);
                const uint64_t span_b1 = readlane64(need_b1, k_hi);
                const uint32_t skew = (uint32_t)((uintptr_t)(s + span_b0) & 15);
                const uint64_t span = span_b1 - span_b0 + skew;
                if (span <= STAGE_BYTES) {
                    use_lds = true;
                    lds_origin = span_b0 - skew;  // wraps below zero harmlessly: only differences are used
                    const uint4 *src = (const uint4 *)(s + span_b0 - skew);
                    const uint32_t n16 = (uint32_t)((span + 15) >> 4);
                    for (uint32_t c = lane; c < n16; c += 64) ((uint4 *)lds)[c] = src[c];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }

            // ---- decode the window: 4 steps of 128 values ---------------------------------
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t e0 = wbase + 128u * j + 2u * lane;
                uint64_t c[2];
                bool head[2], act[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const uint64_t e = e0 + t;
                    act[t] = e >= lo && e < hi;
                    const uint32_t e_rel = act[t] ? (uint32_t)(e - batch_beg) : (uint32_t)(lo - batch_beg);
                    RunFields f;
                    if (nb == 1) {
                        f.desc = readlane64(me.desc, 0);
                        f.base = readlane64(me.base, 0);
                        f.aux = readlane64(me.aux, 0);
                        f.rel_out = 0;
                    } else {
                        // largest k with rel_out[k] <= e_rel (lanes >= nb hold 0xFFFFFFFF)
                        int k = 0;
#pragma unroll
                        for (int sft = 32; sft >= 1; sft >>= 1) {
                            const uint32_t v = __shfl(my_rel, k + sft);
                            if (v <= e_rel) k += sft;
                        }
                        f.rel_out = __shfl(my_rel, k);
                        f.desc = shfl64(me.desc, k);
                        f.base = shfl64(me.base, k);
                        f.aux = shfl64(me.aux, k);
                    }
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
                            const uint64_t bit = ((desc_payload(f.desc) - lds_origin) << 3) + (uint64_t)fidx * w;
                            m = extract_bits<false>(lds, bit, w);
                        } else {
                            const uint8_t *p = s + desc_payload(f.desc);
                            const uint32_t sk = (uint32_t)((uintptr_t)p & 3);
                            m = extract_bits<true>((const uint32_t *)(p - sk), (uint64_t)sk * 8 + (uint64_t)fidx * w, w);
                        }
                        if (kind == KIND_DIRECT) val = is_signed ? zigzag(m) : m;
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

                int64_t *dst = out + e0;
                if (act[0] && act[1] && out_aligned) {
                    *(ulonglong2 *)dst = make_ulonglong2(v0, v1);
                } else {
                    if (act[0]) dst[0] = (int64_t)v0;
                    if (act[1]) dst[1] = (int64_t)v1;
                }
            }
        }

        // ---- PatchedBase: overwrite the patched positions ------------------------------------
        uint64_t pb_mask = __ballot((uint32_t)lane < nb && my_kind == KIND_PATCHED_BASE);
        if (pb_mask) {
            __builtin_amdgcn_s_waitcnt(0);  // the unpatched values of this batch are written
            while (pb_mask) {
                const int k = __ffsll((unsigned long long)pb_mask) - 1;
                pb_mask &= pb_mask - 1;
                const uint64_t desc = readlane64(me.desc, k), base = readlane64(me.base, k);
                const uint64_t aux = readlane64(me.aux, k), ro = readlane64(me.out_off, k);
                const uint32_t w = desc_width(desc), rlen = desc_len(desc);
                const uint32_t pw = pb_patch_width(aux), pll = pb_list_len(aux), cfb = pb_entry_bits(aux);
                const uint8_t *p = s + desc_payload(desc);
                const uint32_t sk = (uint32_t)((uintptr_t)p & 3);
                const uint32_t *words = (const uint32_t *)(p - sk);
                const uint64_t list_bit = (uint64_t)sk * 8 + (((uint64_t)w * rlen + 7) >> 3) * 8;
                uint64_t gap = 0, patch = 0;
                if ((uint32_t)lane < pll) {
                    const uint64_t entry = extract_bits<true>(words, list_bit + (uint64_t)lane * cfb, cfb);
                    gap = pw >= 64 ? 0 : entry >> pw;
                    patch = pw >= 64 ? entry : entry & ((1ull << pw) - 1);
                }
                uint32_t posn = (uint32_t)gap;  // position = inclusive sum of the gaps
#pragma unroll
                for (int d = 1; d < 32; d <<= 1) {
                    const uint32_t up = __shfl_up(posn, d);
                    if (lane >= d) posn += up;
                }
                const bool filler = gap == 255 && patch == 0;  // only extends the gap
                if ((uint32_t)lane < pll && !filler && posn < rlen && ro + posn < num_values) {
                    uint64_t m = w ? extract_bits<true>(words, (uint64_t)sk * 8 + (uint64_t)posn * w, w) : 0;
                    if (w < 64) m |= patch << w;
                    out[ro + posn] = (int64_t)(base + m);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------

hipError_t launch_rlev2_count(hipStream_t st, const uint8_t *stream, uint64_t len,
                              const uint64_t *entries, uint32_t n_seg, int is_signed,
                              SegCount *seg_count) {
    const uint32_t blocks = (n_seg + 255) / 256;
    rlev2_scan_kernel<false><<<blocks, 256, 0, st>>>(stream, len, entries, n_seg, is_signed,
                                                    seg_count, nullptr, nullptr, nullptr, 0);
    return hipGetLastError();
}

hipError_t launch_segment_scan(hipStream_t st, const SegCount *seg_count, uint32_t n_seg,
                               uint64_t num_values, SegOffset *seg_off, ScanTotals *totals) {
    segment_scan_kernel<<<1, SCAN_THREADS, 0, st>>>(seg_count, n_seg, num_values, seg_off, totals);
    return hipGetLastError();
}

hipError_t launch_rlev2_emit(hipStream_t st, const uint8_t *stream, uint64_t len,
                             const uint64_t *entries, uint32_t n_seg, int is_signed,
                             const SegCount *seg_count, const SegOffset *seg_off, RunEntry *runs,
                             uint32_t *group_first, uint32_t n_group_slots) {
    const uint32_t blocks = (n_seg + 255) / 256;
    rlev2_scan_kernel<true><<<blocks, 256, 0, st>>>(stream, len, entries, n_seg, is_signed,
                                                   const_cast<SegCount *>(seg_count), seg_off,
                                                   runs, group_first, n_group_slots);
    return hipGetLastError();
}

hipError_t launch_rlev2_values(hipStream_t st, const uint8_t *stream, uint64_t len,
                               const RunEntry *runs, const uint32_t *group_first,
                               uint32_t n_groups, uint32_t n_runs, uint64_t num_values,
                               int is_signed, int64_t *out) {
    if (n_groups == 0) return hipSuccess;
    const uint32_t blocks = (n_groups + VAL_WAVES - 1) / VAL_WAVES;
    rlev2_values_kernel<<<blocks, VAL_WAVES * 64, 0, st>>>(stream, len, runs, group_first, n_groups,
                                                          n_runs, num_values, is_signed, out);
    return hipGetLastError();
}

}  // namespace orcd

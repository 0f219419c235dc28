// Byte-RLE and boolean expansion for gfx950 (PRESENT and boolean DATA streams).
//
// Byte-RLE as in src/read/decode/boolean_rle.rs:28-62 of the orc-format crate: a control byte
// c >= 0 announces c + 3 copies of the next byte, c < 0 announces -c literal bytes; a boolean
// stream carries 8 values per byte, most significant bit first (boolean_rle.rs:98,118).  Run and
// literal lengths are exactly those of the ORC specification.  Same two-phase shape as the RLE v2
// decoder: control bytes form a serial chain walked per segment, then the bytes are expanded.
#include <algorithm>

#include "orcd_internal.h"
#include "seg_scan.h"

namespace orcd {

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
constexpr uint64_t BR_IS_RUN = 1ull << 48;
constexpr uint32_t BR_WINDOW_SHIFT = 7;  // the expansion looks runs up per window of 128 bytes
constexpr uint32_t BR_WINDOW = 1u << BR_WINDOW_SHIFT;

size_t byte_rle_window_slots(uint64_t total_bytes) {
    return (size_t)((total_bytes + BR_WINDOW - 1) >> BR_WINDOW_SHIFT) + 1;
}

__device__ __forceinline__ uint64_t readfirst_u64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// COUNT pass: runs and bytes per segment.  EMIT pass: the run table, a sentinel run one past the
// end, and win_first[w] = index of the first run that starts inside or after window w.  A run is
// at most 130 bytes, so it can span at most one whole window in which nothing starts; the lane
// that emits the run fills that slot, and the last lane fills the slots past the last run start.
template <bool EMIT>
__global__ __launch_bounds__(SEG_BLOCK) void byte_rle_scan_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const uint64_t *__restrict__ entries,
    const uint64_t *__restrict__ seg_ends, uint32_t n_seg, SegTable t, ByteRun *__restrict__ runs,
    uint32_t *__restrict__ win_first, uint32_t n_win_slots, uint64_t *__restrict__ staged,
    uint32_t staged_cap, uint32_t staged_up_to) {
    __shared__ uint64_t staged_words[EMIT ? 1 : SEG_BLOCK * SB_STRIDE];
    const uint32_t seg = blockIdx.x * SEG_BLOCK + threadIdx.x;
    const bool valid = seg < n_seg;
    if (EMIT && !valid) return;
    StagedWriter keep = {(!EMIT && staged && valid) ? staged + (uint64_t)seg * staged_cap : nullptr,
                         staged_words + (EMIT ? 0 : threadIdx.x * SB_STRIDE), staged_cap};
    // seg_ends (nullable): the segments are slices of a larger buffer, each a stream of its own
    // that ends where it says (see the RLE v2 header scan)
    uint64_t pos = (valid && entries && (seg > 0 || seg_ends)) ? entries[seg] : 0;
    uint64_t end = (valid && entries && seg + 1 < n_seg) ? entries[seg + 1] : len;
    if (valid && seg_ends) end = seg_ends[seg];
    const uint64_t limit = seg_ends ? (end < len ? end : len) : len;
    const bool last_seg = seg + 1 == n_seg;
    if (pos > len) pos = len;
    uint32_t n_runs = 0, status = SEG_OK, last_len = 0;
    uint64_t n_bytes = 0;
    uint64_t out_off = 0;
    uint32_t run_idx = 0, want_runs = 0xFFFFFFFFu, prev_len = 0;
    if (EMIT) {
        const SegOffset so = seg_global_offset(t, seg);
        out_off = so.val_off;
        run_idx = so.run_off;
        prev_len = so.prev_len;
        want_runs = t.seg_count[seg].n_runs;
        if (staged_up_to && want_runs <= staged_up_to) return;  // emitted from the staged words
    } else if (end > len || end < pos) {
        status = SEG_MISALIGNED;
    }
    while (valid && status == SEG_OK && pos < end && n_runs < want_runs) {
        const uint32_t c = s[pos];
        uint32_t n;
        uint64_t next;
        uint64_t src;
        if (c < 128) {  // run
            n = c + 3;
            next = pos + 2;
            src = (pos + 1) | BR_IS_RUN;
        } else {  // literals
            n = 256 - c;
            next = pos + 1 + n;
            src = pos + 1;
        }
        if (next > limit) { status = SEG_TRUNCATED; break; }
        if (pos + 1 > PAYLOAD_MASK) { status = SEG_TOO_LONG; break; }
        if (!EMIT) keep.put(n_runs, src | ((uint64_t)(n - 1) << 40));
        if (EMIT) {
            ByteRun r;
            r.src = src | ((uint64_t)(n - 1) << 40);
            r.out_off = out_off;
            runs[run_idx] = r;
            const uint64_t win = out_off >> BR_WINDOW_SHIFT;
            const bool first = out_off == 0 || ((out_off - prev_len) >> BR_WINDOW_SHIFT) != win;
            if (first && win < n_win_slots) win_first[win] = run_idx;
            // a window this run covers entirely without starting in it
            const uint64_t cov = win + 1;
            if ((out_off & (BR_WINDOW - 1)) != 0 && ((cov + 1) << BR_WINDOW_SHIFT) <= out_off + n &&
                cov < n_win_slots)
                win_first[cov] = run_idx + 1;
            run_idx++;
            out_off += n;
            prev_len = n;
        }
        n_runs++;
        n_bytes += n;
        last_len = n;
        pos = next;
    }
    if (!EMIT) {
        keep.finish(n_runs);
        if (status == SEG_OK && pos != end && !last_seg) status = SEG_MISALIGNED;
        seg_publish(t, seg, n_seg, n_runs, last_len, status, n_bytes);
    } else if (last_seg) {
        ByteRun r;  // sentinel: an empty literal one past the end
        r.src = 0;
        r.out_off = out_off;
        runs[run_idx] = r;
        uint64_t w = out_off == 0 ? 0 : ((out_off - prev_len) >> BR_WINDOW_SHIFT) + 1;
        const uint64_t w_end = (out_off + BR_WINDOW - 1) >> BR_WINDOW_SHIFT;
        for (; w <= w_end && w < n_win_slots; w++) win_first[w] = run_idx;
    }
}

// COUNT for streams of few segments (a stripe's PRESENT stream with an entry point per row group):
// the lane-per-segment walk above waits for memory at every control byte, about a microsecond a
// hop, with a handful of wavefronts on the whole chip -- a hundred hops are a hundred
// microseconds whatever the stream's size.  Here a wavefront copies the bytes that `group`
// neighbouring segments span into LDS first (loads that write LDS themselves, every line fetched
// once) and `group` of its lanes walk a segment each out of LDS, a hop an LDS read.  A span that
// does not fit the slice is walked from memory as above.
// Writes t.seg_count only; byte_rle_publish_kernel turns that into the scan's inputs.
// One wavefront per workgroup; dynamic LDS: slice_bytes + 32.
__global__ __launch_bounds__(64) void byte_rle_count_staged_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const uint64_t *__restrict__ entries,
    const uint64_t *__restrict__ seg_ends, uint32_t n_seg, SegTable t, uint64_t *__restrict__ staged,
    uint32_t staged_cap, uint32_t group, uint32_t slice_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t br_dyn[];
    const uint32_t lane = threadIdx.x;
    const uint64_t first64 = (uint64_t)blockIdx.x * group;
    if (first64 >= n_seg) return;  // whole wavefront
    const uint32_t seg = (uint32_t)first64 + lane;
    const bool valid = lane < group && seg < n_seg;
    // a segment's bounds: as in byte_rle_scan_kernel
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
        const uint64_t ol = ((uint64_t)__shfl_xor((uint32_t)(lo >> 32), d) << 32) | __shfl_xor((uint32_t)lo, d);
        const uint64_t oh = ((uint64_t)__shfl_xor((uint32_t)(hi >> 32), d) << 32) | __shfl_xor((uint32_t)hi, d);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    lo = readfirst_u64(lo);
    hi = readfirst_u64(hi);
    const uint32_t skew = (uint32_t)((uintptr_t)(s + (hi > lo ? lo : 0)) & 15);
    const bool in_lds = hi > lo && hi - lo + skew <= slice_bytes;
    if (in_lds) {
        const uintptr_t a0 = (uintptr_t)(s + lo) - skew;
        const uint32_t n16 = (uint32_t)((hi - lo + skew + 15) >> 4);  // hi <= len: every chunk starts inside the stream
        for (uint32_t c0 = 0; c0 < n16; c0 += 64) {
            const uint32_t c = c0 + lane;
            if (c < n16)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a0 + 16 * (uintptr_t)c),
                                                 (__attribute__((address_space(3))) void *)(br_dyn + 16 * c0), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t n_runs = 0, last_len = 0;
    uint64_t n_bytes = 0;
    uint64_t *mine = staged + (uint64_t)seg * staged_cap;
    const uint32_t cap_mine = (staged && valid) ? staged_cap : 0;
    if (walking) {
        while (status == SEG_OK && pos < end) {
            const uint32_t c = in_lds ? br_dyn[(uint32_t)(pos - lo) + skew] : s[pos];
            const bool run = c < 128;
            const uint32_t n = run ? c + 3 : 256 - c;
            const uint64_t next = run ? pos + 2 : pos + 1 + n;
            if (next > limit) {
                status = SEG_TRUNCATED;
                break;
            }
            if (pos + 1 > PAYLOAD_MASK) {
                status = SEG_TOO_LONG;
                break;
            }
            if (n_runs < cap_mine) mine[n_runs] = ((pos + 1) | (run ? BR_IS_RUN : 0)) | ((uint64_t)(n - 1) << 40);
            n_runs++;
            n_bytes += n;
            last_len = n;
            pos = next;
        }
        if (status == SEG_OK && pos != end && !last_seg) status = SEG_MISALIGNED;
    }
    if (valid) {
        SegCount c;
        c.n_runs = n_runs;
        c.last_len = last_len;
        c.status = status;
        c.pad = 0;
        c.n_vals = n_bytes;
        t.seg_count[seg] = c;
    }
}

// COUNT for streams of few long segments: a wavefront per segment.  The lanes look at the 64 bytes
// that follow the current control byte at once, each taking its byte for a control byte and
// working out where the next one would lie (2 bytes on for a run, 1 + count on for literals); the
// first of the 64 is a real control byte, and the chain that starts there is picked out of the 64
// guesses by pointer doubling (five rounds of lane permutes: lane i holds the first 2^k links of
// the chain from i; a hop is two bytes at least, so 32 links cover the 64 offsets).  The lanes on
// the chain write their run words straight to the staging area, in order.  The segment's bytes
// are copied to LDS first when they fit (loads that write LDS themselves).  A stripe's PRESENT
// stream, 430 segments of 1 250 bytes and 150 runs each: 18 microseconds, against 48 with a lane
// per segment out of LDS and 103 from memory.
// Writes t.seg_count only; byte_rle_publish_kernel turns that into the scan's inputs.
// One wavefront per workgroup; dynamic LDS: slice_bytes + 32.
__global__ __launch_bounds__(64) void byte_rle_coop_count_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const uint64_t *__restrict__ entries,
    const uint64_t *__restrict__ seg_ends, uint32_t n_seg, SegTable t, uint64_t *__restrict__ staged,
    uint32_t staged_cap, uint32_t slice_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t br_dyn[];
    const uint32_t lane = threadIdx.x;
    const uint32_t seg = blockIdx.x;
    if (seg >= n_seg) return;  // whole wavefront
    // the segment's bounds: as in byte_rle_scan_kernel
    uint64_t pos = (entries && (seg > 0 || seg_ends)) ? entries[seg] : 0;
    uint64_t end = (entries && seg + 1 < n_seg) ? entries[seg + 1] : len;
    if (seg_ends) end = seg_ends[seg];
    const uint64_t limit = seg_ends ? (end < len ? end : len) : len;
    const bool last_seg = seg + 1 == n_seg;
    if (pos > len) pos = len;
    pos = readfirst_u64(pos);
    end = readfirst_u64(end);
    uint32_t status = SEG_OK;
    if (end > len || end < pos) status = SEG_MISALIGNED;
    const uint64_t lo = pos;
    const uint32_t skew = (uint32_t)((uintptr_t)(s + lo) & 15);
    const bool in_lds = status == SEG_OK && end > lo && end - lo + skew <= slice_bytes;
    if (in_lds) {
        const uintptr_t a0 = (uintptr_t)(s + lo) - skew;
        const uint32_t n16 = (uint32_t)((end - lo + skew + 15) >> 4);  // end <= len: every chunk starts inside the stream
        for (uint32_t c0 = 0; c0 < n16; c0 += 64) {
            const uint32_t c = c0 + lane;
            if (c < n16)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a0 + 16 * (uintptr_t)c),
                                                 (__attribute__((address_space(3))) void *)(br_dyn + 16 * c0), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t n_runs = 0, last_len = 0;
    uint64_t n_bytes = 0;
    uint64_t *mine = staged ? staged + (uint64_t)seg * staged_cap : nullptr;
    while (status == SEG_OK && pos < end) {
        const uint64_t p = pos + lane;
        const bool inb = p < end;
        uint32_t c = 0;
        if (inb) c = in_lds ? br_dyn[(uint32_t)(p - lo) + skew] : s[p];
        const bool run = c < 128;
        const uint32_t n = run ? c + 3 : 256 - c;        // bytes the run or the literals decode to
        const uint32_t hop = run ? 2u : 1u + n;          // to the next control byte
        // a control byte whose run does not fit the stream (or the 40 bits of a run word) ends
        // the chain; it is reported once it is the byte at pos
        const bool fits = inb && p + hop <= limit && p + 1 <= PAYLOAD_MASK;
        if (!__builtin_amdgcn_readfirstlane((int)fits)) {
            status = (p + 1 > PAYLOAD_MASK && __builtin_amdgcn_readfirstlane((int)(p + hop <= limit))) ? SEG_TOO_LONG : SEG_TRUNCATED;
            break;
        }
        uint32_t jump = (fits && lane + hop < 64) ? lane + hop : lane;  // (itself: the chain stops here)
        uint64_t links = fits ? 1ull << lane : 0;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const uint32_t l_lo = __shfl((uint32_t)links, (int)jump), l_hi = __shfl((uint32_t)(links >> 32), (int)jump);
            links |= ((uint64_t)l_hi << 32) | l_lo;
            jump = __shfl(jump, (int)jump);
        }
        const uint64_t chain = readfirst_u64(links);  // bit 0 is set: the byte at pos fits
        const bool on = (chain >> lane) & 1;
        const uint32_t rank = (uint32_t)__popcll((unsigned long long)(chain & ((1ull << lane) - 1)));
        if (on && mine && n_runs + rank < staged_cap)
            mine[n_runs + rank] = ((p + 1) | (run ? BR_IS_RUN : 0)) | ((uint64_t)(n - 1) << 40);
        uint32_t sum = on ? n : 0;
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
        const int last = 63 - __builtin_clzll((unsigned long long)chain);
        n_bytes += sum;
        n_runs += (uint32_t)__popcll((unsigned long long)chain);
        last_len = (uint32_t)__builtin_amdgcn_readlane((int)n, last);
        pos += (uint32_t)last + (uint32_t)__builtin_amdgcn_readlane((int)hop, last);
    }
    if (status == SEG_OK && pos != end && !last_seg) status = SEG_MISALIGNED;
    if (lane == 0) {
        SegCount c;
        c.n_runs = n_runs;
        c.last_len = last_len;
        c.status = status;
        c.pad = 0;
        c.n_vals = n_bytes;
        t.seg_count[seg] = c;
    }
}

// What byte_rle_count_staged_kernel left in t.seg_count, through the workgroup-local scan that the
// counting kernel proper ends with.
__global__ __launch_bounds__(SEG_BLOCK) void byte_rle_publish_kernel(uint32_t n_seg, SegTable t) {
    const uint32_t seg = blockIdx.x * SEG_BLOCK + threadIdx.x;
    SegCount c;
    c.n_runs = 0;
    c.last_len = 0;
    c.status = SEG_OK;
    c.n_vals = 0;
    if (seg < n_seg) c = t.seg_count[seg];
    seg_publish(t, seg, n_seg, c.n_runs, c.last_len, c.status, c.n_vals);
}

// EMIT for the segments whose run words the counting pass staged: a group of lanes per segment,
// one lane per run (a segment with more runs than lanes takes them a group at a time); a run's
// place in the output follows from a scan of the lengths.
__global__ __launch_bounds__(256) void byte_rle_emit_staged_kernel(
    uint32_t n_seg, SegTable t, const uint64_t *__restrict__ staged, uint32_t staged_cap,
    ByteRun *__restrict__ runs, uint32_t *__restrict__ win_first, uint32_t n_win_slots, uint32_t group) {
    // `group` lanes per segment (see rlev2_emit_staged_kernel)
    const uint64_t seg64 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / group;
    const uint32_t lane = threadIdx.x & (group - 1);
    if (seg64 >= n_seg) return;
    const uint32_t seg = (uint32_t)seg64;
    // (a staged slot that was never written holds whatever it held; it is not looked at)
    const uint64_t *mine = staged + (uint64_t)seg * staged_cap;
    uint64_t staged_word = mine[lane];
    const uint32_t n_runs = t.seg_count[seg].n_runs;
    const SegOffset so = seg_global_offset(t, seg);
    if (n_runs > staged_cap) return;  // the walking kernel emits this one
    const bool last_seg = seg + 1 == n_seg;
    if (n_runs == 0 && !last_seg) return;
    uint64_t before = so.val_off;
    uint32_t carry_prev = so.prev_len;
    for (uint32_t base = 0; base == 0 || base < n_runs; base += group) {
        if (base) staged_word = mine[base + lane];
        const uint32_t k = base + lane;
        const bool live = k < n_runs;
        const uint64_t src = live ? staged_word : 0;
        const uint32_t n = live ? (uint32_t)((src >> 40) & 0xFF) + 1 : 0;
        uint32_t incl = n;
        for (uint32_t d = 1; d < group; d <<= 1) {
            const uint32_t u = __shfl_up(incl, d, (int)group);
            if (lane >= d) incl += u;
        }
        uint32_t prev_len = __shfl_up(n, 1, (int)group);
        if (lane == 0) prev_len = carry_prev;
        const uint64_t out_off = before + (incl - n);
        const uint32_t run_idx = so.run_off + k;
        if (live) {
            ByteRun r;
            r.src = src;
            r.out_off = out_off;
            runs[run_idx] = r;
            // the same window rules as the walking kernel's
            const uint64_t win = out_off >> BR_WINDOW_SHIFT;
            const bool first = out_off == 0 || ((out_off - prev_len) >> BR_WINDOW_SHIFT) != win;
            if (first && win < n_win_slots) win_first[win] = run_idx;
            const uint64_t cov = win + 1;
            if ((out_off & (BR_WINDOW - 1)) != 0 && ((cov + 1) << BR_WINDOW_SHIFT) <= out_off + n &&
                cov < n_win_slots)
                win_first[cov] = run_idx + 1;
        }
        if (last_seg && k + 1 == (n_runs ? n_runs : 1u)) {
            const uint64_t end_off = out_off + n;
            const uint32_t end_prev = n_runs ? n : so.prev_len;
            const uint32_t end_idx = so.run_off + n_runs;
            ByteRun r;  // sentinel: an empty literal one past the end
            r.src = 0;
            r.out_off = end_off;
            runs[end_idx] = r;
            uint64_t w = end_off == 0 ? 0 : ((end_off - end_prev) >> BR_WINDOW_SHIFT) + 1;
            const uint64_t w_end = (end_off + BR_WINDOW - 1) >> BR_WINDOW_SHIFT;
            for (; w <= w_end && w < n_win_slots; w++) win_first[w] = end_idx;
        }
        before += __shfl(incl, (int)group - 1, (int)group);
        carry_prev = __shfl(n, (int)group - 1, (int)group);
    }
}

__device__ __forceinline__ uint32_t byte_run_len(const ByteRun &r) {
    return (uint32_t)((r.src >> 40) & 0xFF) + 1;
}

// Walks a lane through the run table: find(o) positions it at or shortly before the run that
// holds decoded byte o, byte_at(pos) returns decoded byte pos for ascending pos >= o.
struct RunCursor {
    const uint8_t *s;
    uint64_t len;
    const ByteRun *runs;
    uint32_t n_runs;
    uint32_t r;
    ByteRun run;

    __device__ __forceinline__ void find(const uint32_t *__restrict__ win_first, uint64_t o) {
        // the run before the first one that starts in o's window began before the window; the
        // few runs between it and o are walked by byte_at
        r = win_first[o >> BR_WINDOW_SHIFT];
        if (r > n_runs) r = n_runs;
        r = r > 0 ? r - 1 : 0;
        run = runs[r];
    }
    // Stream bytes are fetched as aligned dwords (a byte-sized load costs the memory pipeline as
    // much per lane as a dword-sized one, and consecutive literals share a dword): the last one
    // is kept.  The aligned dword around a valid byte never leaves the byte's own page.
    uintptr_t word_addr;
    uint32_t word;

    __device__ __forceinline__ uint32_t byte_at(uint64_t pos) {
        // (the scan found all wanted bytes in the runs, so r stays below n_runs)
        while (r < n_runs && pos >= run.out_off + byte_run_len(run)) run = runs[++r];
        const uint64_t src = (run.src & PAYLOAD_MASK) + ((run.src & BR_IS_RUN) ? 0 : pos - run.out_off);
        if (!(r < n_runs && src < len)) return 0u;
        const uintptr_t a = (uintptr_t)(s + src), aw = a & ~(uintptr_t)3;
        if (aw != word_addr) {
            word = *(const uint32_t *)aw;
            word_addr = aw;
        }
        return (word >> (8 * (uint32_t)(a & 3))) & 0xFFu;
    }
};

// Bytes: one lane per 16 consecutive decoded bytes, one 16-byte store each.
__global__ __launch_bounds__(256) void byte_rle_expand_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const ByteRun *__restrict__ runs,
    const uint32_t *__restrict__ win_first, uint32_t n_runs, uint64_t num_bytes,
    uint8_t *__restrict__ out) {
    const uint64_t n_lanes = (num_bytes + 15) / 16;
    RunCursor cur = {s, len, runs, n_runs, 0, {0, 0}, 0, 0};
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_lanes;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t o = t * 16;
        cur.find(win_first, o);
        uint32_t d[4] = {0, 0, 0, 0};
#pragma unroll
        for (uint32_t k = 0; k < 16; k++)
            if (o + k < num_bytes) d[k >> 2] |= cur.byte_at(o + k) << (8 * (k & 3));
        uint8_t *dst = out + o;
        if (o + 16 <= num_bytes && (((uintptr_t)dst) & 15) == 0) {
            *(uint4 *)dst = make_uint4(d[0], d[1], d[2], d[3]);
        } else {
            for (uint32_t k = 0; k < 16 && o + k < num_bytes; k++) dst[k] = (uint8_t)(d[k >> 2] >> (8 * (k & 3)));
        }
    }
}

// Bytes, tile by tile: a wavefront takes BX_TILE decoded bytes; lane i of step j owns the 16
// bytes at 1024*j + 16*i of the tile, so every store instruction writes 1 KiB of consecutive
// output.  The runs that overlap the tile (at most 64, one per lane) go into a small LDS table,
// the stream bytes they span -- one contiguous stretch -- into LDS with 16-byte loads; each lane
// then finds the run of its first byte by binary search in the table and assembles its four
// dwords from LDS: a dword inside one run is an unaligned LDS dword (literals) or a replicated
// byte (run), a dword across two runs is a byte-wise mix of the two, anything finer goes byte by
// byte.  Tiles with more runs or a longer stretch than fit fall back to per-lane walks.
constexpr uint32_t BX_STEPS = 4;
constexpr uint32_t BX_TILE = 1024 * BX_STEPS;
constexpr uint32_t BX_RUNS = 64;
constexpr uint32_t BX_SPAN = BX_RUNS * 132;  // stream bytes of 64 runs, control bytes between them included
constexpr uint32_t BX_PAD = 16;              // LDS bytes in front of the stretch (never loaded, never used)

struct __attribute__((aligned(16))) BxWave {
    uint32_t bytes[(BX_PAD + BX_SPAN + 32) / 4 + 2];
    int32_t rel[BX_RUNS + 2];    // start of run k relative to the tile; INT_MAX past the last
    uint32_t srel[BX_RUNS + 2];  // its source relative to the first run's; bit 31: a run, not literals
};

struct BxCursor {  // runs k and k + 1 of the tile's table, and where k + 2 starts
    const BxWave &L;
    uint32_t base;  // LDS byte index of the first run's source
    uint32_t k;
    int32_t relA, relB, relC;
    uint32_t srelA, srelB;
    __device__ __forceinline__ void load(uint32_t k_) {
        k = k_;
        relA = L.rel[k];
        srelA = L.srel[k];
        relB = L.rel[k + 1];
        srelB = L.srel[k + 1];
        relC = k + 2 <= BX_RUNS + 1 ? L.rel[k + 2] : 0x7FFFFFFF;
    }
    __device__ __forceinline__ void advance() {
        if (k + 1 > BX_RUNS) return;  // (sentinels keep q below relB before this can happen)
        load(k + 1);
    }
    // the 4 decoded bytes at tile position q as if all of them belonged to the run (rel, srel)
    __device__ __forceinline__ uint32_t dword_of(int32_t rel, uint32_t srel, int32_t q) const {
        const uint32_t off = srel & 0x7FFFFFFFu;
        if (srel >> 31) {
            const uint32_t a = base + off;
            return ((L.bytes[a >> 2] >> (8 * (a & 3))) & 0xFFu) * 0x01010101u;
        }
        const uint32_t a = base + off + (uint32_t)(q - rel);  // q - rel >= -3, base >= BX_PAD
        return __builtin_amdgcn_alignbyte(L.bytes[(a >> 2) + 1], L.bytes[a >> 2], a & 3);
    }
    __device__ __forceinline__ uint32_t dword_at(int32_t q) {
        while (q >= relB) advance();
        if (q + 4 <= relB) return dword_of(relA, srelA, q);
        if (q + 4 <= relC) {  // bytes below relB from run k, the rest from run k + 1
            const uint32_t keep = 0xFFFFFFFFu >> (8 * (4 - (uint32_t)(relB - q)));
            return (dword_of(relA, srelA, q) & keep) | (dword_of(relB, srelB, q) & ~keep);
        }
        uint32_t w = 0;
        for (int32_t j = 0; j < 4; j++) {
            while (q + j >= relB) advance();
            w |= (dword_of(relA, srelA, q + j) & 0xFFu) << (8 * j);
        }
        return w;
    }
};

__global__ __launch_bounds__(256) void byte_rle_tile_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const ByteRun *__restrict__ runs,
    const uint32_t *__restrict__ win_first, uint32_t n_win_slots, uint32_t n_runs,
    uint64_t num_bytes, uint8_t *__restrict__ out) {
    __shared__ BxWave lds[4];
    BxWave &L = lds[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t n_tiles = (num_bytes + BX_TILE - 1) / BX_TILE;
    for (uint64_t tile = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < n_tiles;
         tile += (uint64_t)gridDim.x * 4) {
        const uint64_t tile_base = tile * BX_TILE;
        // runs [r_lo, r_hi) overlap the tile (see bool_expand_kernel)
        const uint64_t w0 = tile_base >> BR_WINDOW_SHIFT;
        const uint64_t w1 = (w0 + BX_TILE / BR_WINDOW < n_win_slots) ? w0 + BX_TILE / BR_WINDOW : n_win_slots - 1;
        uint32_t r_lo = __builtin_amdgcn_readfirstlane(win_first[w0]);
        uint32_t r_hi = __builtin_amdgcn_readfirstlane(win_first[w1]);
        if (r_hi > n_runs) r_hi = n_runs;
        r_lo = r_lo > 0 ? r_lo - 1 : 0;
        if (r_lo > r_hi) r_lo = r_hi;
        const uint32_t nr = r_hi - r_lo;
        bool fast = nr >= 1 && nr <= BX_RUNS;
        int32_t my_rel = 0x7FFFFFFF;
        uint64_t my_src = 0;
        uint64_t src0 = 0, span = 0;
        if (fast) {
            if (lane < nr) {
                const ByteRun run = runs[r_lo + lane];
                my_rel = (int32_t)(int64_t)(run.out_off - tile_base);
                my_src = run.src;
            }
            src0 = readfirst_u64(my_src) & PAYLOAD_MASK;
            const uint64_t last = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_src >> 32), (int)nr - 1) << 32) |
                                  (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_src, (int)nr - 1);
            const uint64_t end = (last & PAYLOAD_MASK) + ((last & BR_IS_RUN) ? 1 : ((last >> 40) & 0xFF) + 1);
            span = end - src0;
            // the first run of the tile must begin at or before it, and the stretch must fit
            fast = end > src0 && span <= BX_SPAN && end <= len && __builtin_amdgcn_readfirstlane(my_rel) <= 0;
        }
        if (fast) {
            const uint8_t *sb = s + src0;
            const uint32_t skew = (uint32_t)((uintptr_t)sb & 15);
            const uint4 *src16 = (const uint4 *)(sb - skew);  // aligned chunks that each hold a byte of the stretch
            const uint32_t n16 = (uint32_t)((span + skew + 15) >> 4);
            uint4 *dst16 = (uint4 *)(L.bytes + BX_PAD / 4);
            // loads that write LDS themselves, 1 KiB per instruction (lane i's 16 bytes land at the
            // instruction's LDS address + 16 i): all of the stretch is asked for at once -- a loop
            // that stores each KiB before it asks for the next pays a memory round trip per KiB,
            // and a tile's stretch is four of them
            for (uint32_t c0 = 0; c0 < n16; c0 += 64)
                if (c0 + lane < n16)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src16 + c0 + lane),
                                                     (__attribute__((address_space(3))) void *)(dst16 + c0), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            L.rel[lane] = my_rel;
            L.srel[lane] = (uint32_t)((my_src & PAYLOAD_MASK) - src0) | ((my_src & BR_IS_RUN) ? 0x80000000u : 0u);
            if (lane < 2) {
                L.rel[BX_RUNS + lane] = 0x7FFFFFFF;
                L.srel[BX_RUNS + lane] = 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            BxCursor cur = {L, BX_PAD + skew, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
            for (uint32_t step = 0; step < BX_STEPS; step++) {
                const int32_t p = (int32_t)(1024 * step + 16 * lane);
                const uint64_t o = tile_base + (uint32_t)p;
                if (o >= num_bytes) continue;
                // largest k with rel[k] <= p (rel[0] <= 0; entries past the last run hold INT_MAX)
                uint32_t k = 0;
#pragma unroll
                for (uint32_t sft = 32; sft >= 1; sft >>= 1)
                    if (L.rel[k + sft] <= p) k += sft;
                cur.load(k);
                uint32_t d[4];
#pragma unroll
                for (int32_t i = 0; i < 4; i++) d[i] = cur.dword_at(p + 4 * i);
                uint8_t *dst = out + o;
                if (o + 16 <= num_bytes) {
                    *(uint4 *)dst = make_uint4(d[0], d[1], d[2], d[3]);
                } else {
                    for (uint32_t j = 0; j < 16 && o + j < num_bytes; j++) dst[j] = (uint8_t)(d[j >> 2] >> (8 * (j & 3)));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // the next tile's LDS writes come after these reads
        } else {
            RunCursor cur = {s, len, runs, n_runs, 0, {0, 0}, 0, 0};
            for (uint32_t step = 0; step < BX_STEPS; step++) {
                const uint64_t o = tile_base + 1024 * step + 16 * lane;
                if (o >= num_bytes) continue;
                cur.find(win_first, o);
                uint32_t d[4] = {0, 0, 0, 0};
#pragma unroll
                for (uint32_t j = 0; j < 16; j++)
                    if (o + j < num_bytes) d[j >> 2] |= cur.byte_at(o + j) << (8 * (j & 3));
                uint8_t *dst = out + o;
                if (o + 16 <= num_bytes) {
                    *(uint4 *)dst = make_uint4(d[0], d[1], d[2], d[3]);
                } else {
                    for (uint32_t j = 0; j < 16 && o + j < num_bytes; j++) dst[j] = (uint8_t)(d[j >> 2] >> (8 * (j & 3)));
                }
            }
        }
    }
}

// Stream byte at offset src (src < len), fetched as the aligned dword around it.
__device__ __forceinline__ uint32_t stream_byte(const uint8_t *s, uint64_t src, uintptr_t &word_addr,
                                                uint32_t &word) {
    const uintptr_t a = (uintptr_t)(s + src), aw = a & ~(uintptr_t)3;
    if (aw != word_addr) {
        word = *(const uint32_t *)aw;
        word_addr = aw;
    }
    return (word >> (8 * (uint32_t)(a & 3))) & 0xFFu;
}

// Booleans: every decoded byte is 8 values, most significant bit first (boolean_rle.rs:98,118).
// A wavefront takes a tile of BOOL_TILE decoded bytes in BOOL_STEPS steps (a tile costs three
// dependent memory round trips -- window index, runs, stream bytes -- whatever its size; tiles of
// half and of twice the size both measured slower); in a step lane i owns bytes 128*step + 2*i
// and the next one, i.e. 16 values, so that each store instruction writes 1 KiB of consecutive
// values.  The runs that overlap the tile are loaded once, one per lane, the stream bytes between
// the first run's source and the last one's are staged in LDS with one or two 16-byte loads per
// lane (loads that write LDS themselves), and every lane then finds the runs of its bytes with register shuffles and its bytes in
// LDS (a tile overlapped by more than 64 runs falls back to per-lane walks of the run table).
// out_bytes: one byte per value; out_bitmap: the same bits least-significant-first; *set_bits:
// the number of ones.
constexpr uint32_t BOOL_STEPS = 8;
constexpr uint32_t BOOL_TILE = 128 * BOOL_STEPS;
constexpr uint32_t BOOL_SLICE = BOOL_TILE + 384;  // stream bytes of a tile that a wavefront stages in LDS
constexpr uint32_t BOOL_COUNT_SLOTS = 64;  // counters of ones, BOOL_COUNT_PITCH words apart
constexpr uint32_t BOOL_COUNT_PITCH = BOOL_COUNT_BYTES / 8 / BOOL_COUNT_SLOTS;

__global__ __launch_bounds__(256) void bool_expand_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const ByteRun *__restrict__ runs,
    const uint32_t *__restrict__ win_first, uint32_t n_win_slots, uint32_t n_runs,
    uint64_t num_bits, uint8_t *__restrict__ out_bytes, uint8_t *__restrict__ out_bitmap,
    unsigned long long *__restrict__ set_bits) {
    const uint64_t num_bytes = (num_bits + 7) >> 3;  // decoded bytes wanted
    const uint64_t n_tiles = (num_bytes + BOOL_TILE - 1) / BOOL_TILE;
    const uint32_t lane = threadIdx.x & 63;
    // A decoded byte's eight values, one per byte, most significant bit first: a table look-up in
    // LDS (the kernel is bound by the instructions it issues, and the LDS pipe has nothing else to do).
    __shared__ uint64_t spread[256];
    __shared__ __attribute__((aligned(16))) uint8_t slice_all[4][BOOL_SLICE + 16];
    uint8_t *slice = slice_all[threadIdx.x >> 6];
    {
        const uint32_t b = threadIdx.x & 255;
        // replicate the byte, keep one bit per copy, turn "non-zero" into 1
        const uint32_t rep = b * 0x01010101u;
        const uint32_t lo4 = (((rep & 0x10204080u) + 0x7F7F7F7Fu) >> 7) & 0x01010101u;
        const uint32_t hi4 = (((rep & 0x01020408u) + 0x7F7F7F7Fu) >> 7) & 0x01010101u;
        spread[b] = ((uint64_t)hi4 << 32) | lo4;
        __syncthreads();
    }
    const bool bytes_aligned = (((uintptr_t)out_bytes) & 15) == 0;
    const bool bitmap_aligned = (((uintptr_t)out_bitmap) & 1) == 0;
    uint32_t ones = 0;
    // Grid-stride over tiles: the count of ones costs one atomic per wavefront.  (Fetching the
    // window index and the runs of the following tiles ahead of time measured no faster, before
    // and after the stream bytes went through LDS: the kernel is bound by what it issues and
    // stores, not by the three dependent loads of a tile.)
    for (uint64_t tile = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < n_tiles;
         tile += (uint64_t)gridDim.x * 4) {
        const uint64_t tile_base = tile * BOOL_TILE;
        // runs [r_lo, r_hi) overlap the tile: from the one that reaches into its first window to
        // the last one that starts before its end
        const uint64_t w0 = tile_base >> BR_WINDOW_SHIFT;
        const uint64_t w1 = (w0 + BOOL_STEPS < n_win_slots) ? w0 + BOOL_STEPS : n_win_slots - 1;
        uint32_t r_lo = __builtin_amdgcn_readfirstlane(win_first[w0]);
        uint32_t r_hi = __builtin_amdgcn_readfirstlane(win_first[w1]);
        if (r_hi > n_runs) r_hi = n_runs;
        r_lo = r_lo > 0 ? r_lo - 1 : 0;
        if (r_lo > r_hi) r_lo = r_hi;
        // a tile whose values are all wanted needs no masking of a partial last byte
        const bool full = (tile_base + BOOL_TILE) * 8 <= num_bits;
        // the values of one step: this lane's two decoded bytes (the first in bits 0..7) counted,
        // turned into bitmap bits and into one byte per value, and stored
        auto emit = [&](uint32_t step, uint32_t two_in) {
            uint32_t two = two_in;
            const uint64_t o = tile_base + 128 * step + 2 * lane;
            if (!full) {
                if (o >= num_bytes) return;
                // clear the bytes and bits past the last value
                const uint64_t first = o << 3;
                const uint64_t left = num_bits - first;  // > 0: o < num_bytes
                if (left < 16) two &= left <= 8 ? ((0xFF00u >> left) & 0xFFu) : (0xFFu | (((0xFF00u >> (left - 8)) & 0xFFu) << 8));
            }
            ones += __popc(two);
            if (out_bitmap) {
                const uint32_t lsb = __brev(two) >> 16;          // both bytes bit-reversed, swapped
                const uint32_t bm = ((lsb >> 8) | (lsb << 8)) & 0xFFFFu;
                uint8_t *dst = out_bitmap + o;
                if ((full || o + 2 <= num_bytes) && bitmap_aligned) {
                    *(uint16_t *)dst = (uint16_t)bm;
                } else {
                    dst[0] = (uint8_t)bm;
                    if (o + 1 < num_bytes) dst[1] = (uint8_t)(bm >> 8);
                }
            }
            if (out_bytes) {
                // byte j of a word = bit (7 - j) of the decoded byte
                const uint64_t word[2] = {spread[two & 0xFFu], spread[(two >> 8) & 0xFFu]};
                const uint64_t first = o << 3;
                uint8_t *dst = out_bytes + first;
                if ((full || first + 16 <= num_bits) && bytes_aligned) {
                    // written once, not read back here: keep it out of the way of the stream bytes
                    u64x2 v;
                    v.x = word[0];
                    v.y = word[1];
                    __builtin_nontemporal_store(v, (u64x2 *)dst);
                } else {
                    for (uint32_t j = 0; j < 16 && first + j < num_bits; j++)
                        dst[j] = (uint8_t)(word[j >> 3] >> (8 * (j & 7)));
                }
            }
        };
        if (r_hi - r_lo <= 64) {
            // lane k holds run r_lo + k: its start relative to the tile (negative for a run that
            // began before it) and its source relative to the first run's
            int32_t my_rel = 0x7FFFFFFF;
            uint64_t my_src = 0;
            if (lane < r_hi - r_lo) {
                const ByteRun run = runs[r_lo + lane];
                my_rel = (int32_t)(int64_t)(run.out_off - tile_base);
                my_src = run.src;
            }
            const uint64_t src0 = readfirst_u64(my_src) & PAYLOAD_MASK;
            // offset from src0 (runs of one tile lie within 64 * 131 stream bytes), run flag on top
            const uint32_t my_srel = (uint32_t)((my_src & PAYLOAD_MASK) - src0) |
                                     ((my_src & BR_IS_RUN) ? 0x80000000u : 0u);
            // all addressing below is 32-bit, relative to s + src0
            const uint8_t *sb = s + src0;
            const uint64_t avail = len > src0 ? len - src0 : 0;  // stream bytes from src0 on
            // The stream bytes the tile needs lie between the first run's source and the last
            // run's (a literal's payload is at most 128 bytes, a run's one): they are fetched once,
            // 16 bytes per lane, into this wavefront's slice of LDS, and every byte is then picked
            // from there.  (Two dword loads per lane and step -- 16 a tile, for 2 bytes each --
            // kept the memory pipeline busier than the output the tile writes.)  64 runs of a tile
            // span at most BOOL_TILE + 127 + 128 + 64 bytes.
            const uint32_t n_live = r_hi - r_lo;
            const uint32_t srel_last = __builtin_amdgcn_readlane(my_srel, n_live ? (int)n_live - 1 : 0);
            uint64_t span = n_live ? (uint64_t)(srel_last & 0x7FFFFFFFu) + ((srel_last >> 31) ? 1u : 128u) : 0;
            if (span > avail) span = avail;
            const uint32_t skew = (uint32_t)((uintptr_t)sb & 15);
            if (span + skew <= BOOL_SLICE) {
                const uint4 *src16 = (const uint4 *)(sb - skew);
                const uint32_t n16 = (uint32_t)((span + skew + 15) >> 4);  // <= BOOL_SLICE / 16 + 1
                constexpr uint32_t ROUNDS = (BOOL_SLICE / 16 + 1 + 63) / 64;
#pragma unroll
                for (uint32_t r = 0; r < ROUNDS; r++)
                    if (lane + 64 * r < n16)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src16 + lane + 64 * r),
                                                         (__attribute__((address_space(3))) void *)(slice + 1024 * r), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // the last offset a look-up may use: every byte that is wanted lies inside a run,
                // and the scan has checked every run against the stream's length, so only bytes
                // past the last value (which are masked out below) can point further -- those are
                // clamped
                const uint32_t last = (span ? (uint32_t)span - 1u : 0u) + skew;
                // A byte at tile position q inside run k comes from offset base_k + (q & mask_k):
                // a literal's bytes follow one another (base = source - start, mask = all ones), a
                // run's all come from its one value byte (base = source, mask = 0).
                const uint32_t my_mask = (my_srel >> 31) ? 0u : 0xFFFFFFFFu;
                const uint32_t my_base = (my_srel & 0x7FFFFFFFu) + skew - ((uint32_t)my_rel & my_mask);
#pragma unroll
                for (uint32_t step = 0; step < BOOL_STEPS; step++) {
                    const int32_t p = (int32_t)(128 * step + 2 * lane);
                    // The runs that hold the step's first and last byte: the starts ascend over the
                    // lanes, so the number of starts at or below a position is a ballot away.
                    const int32_t p_first = (int32_t)(128 * step);
                    const uint32_t kf = (uint32_t)__popcll(__ballot(my_rel <= p_first)) - 1u;
                    const uint32_t kl = (uint32_t)__popcll(__ballot(my_rel <= p_first + 127)) - 1u;
                    int32_t rel_n;
                    uint32_t base_k, base_n, mask_k, mask_n;
                    if (kl <= kf + 1) {
                        // at most two runs under the step (runs of a hundred bytes and more are the
                        // common case): their fields are wave-uniform, no lane has to search
                        const int a = __builtin_amdgcn_readfirstlane((int)kf), b = (a + 1) & 63;
                        rel_n = a == 63 ? 0x7FFFFFFF : __builtin_amdgcn_readlane(my_rel, b);
                        base_k = __builtin_amdgcn_readlane(my_base, a);
                        base_n = __builtin_amdgcn_readlane(my_base, b);
                        mask_k = __builtin_amdgcn_readlane(my_mask, a);
                        mask_n = __builtin_amdgcn_readlane(my_mask, b);
                    } else {
                        // largest k with rel[k] <= p (lanes past the last run hold INT_MAX; rel[0] <= 0)
                        int k = 0;
#pragma unroll
                        for (int sft = 32; sft >= 1; sft >>= 1) {
                            const int32_t v = __shfl(my_rel, k + sft);
                            if (v <= p) k += sft;
                        }
                        rel_n = __shfl(my_rel, (k + 1) & 63);
                        if (k == 63) rel_n = 0x7FFFFFFF;  // no run after the 64th
                        base_k = __shfl(my_base, k);
                        base_n = __shfl(my_base, (k + 1) & 63);
                        mask_k = __shfl(my_mask, k);
                        mask_n = __shfl(my_mask, (k + 1) & 63);
                    }
                    uint32_t two = 0;
#pragma unroll
                    for (uint32_t h = 0; h < 2; h++) {
                        const int32_t q = p + (int32_t)h;
                        const bool next = q >= rel_n;  // runs are not empty: the next byte is in k or k + 1
                        const uint32_t off = (next ? base_n : base_k) + ((uint32_t)q & (next ? mask_n : mask_k));
                        two |= (uint32_t)slice[off < last ? off : last] << (8 * h);
                    }
                    emit(step, two);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();  // the next tile's bytes come after these look-ups
                continue;
            }
        }
        {
            RunCursor cur = {s, len, runs, n_runs, 0, {0, 0}, 0, 0};
            for (uint32_t step = 0; step < BOOL_STEPS; step++) {
                const uint64_t o = tile_base + 128 * step + 2 * lane;
                uint32_t two = 0;
                if (o < num_bytes) {
                    cur.find(win_first, o);
                    two = cur.byte_at(o);
                    if (o + 1 < num_bytes) two |= cur.byte_at(o + 1) << 8;
                }
                emit(step, two);
            }
        }
    }
    if (set_bits) {
        // every wavefront of the workgroup is here, lanes without work carrying 0: one atomic per
        // workgroup (they all arrive at the end of the kernel and queue up on the one address)
        __shared__ uint32_t wave_ones[4];
        for (int d = 32; d >= 1; d >>= 1) ones += __shfl_down(ones, d);
        if (lane == 0) wave_ones[threadIdx.x >> 6] = ones;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long sum = (unsigned long long)wave_ones[0] + wave_ones[1] + wave_ones[2] + wave_ones[3];
            // ... spread over BOOL_COUNT_SLOTS counters, a cache line apart
            if (sum) atomicAdd(set_bits + (blockIdx.x % BOOL_COUNT_SLOTS) * BOOL_COUNT_PITCH, sum);
        }
    }
}

// set_bits[0] = sum of the counters of bool_expand_kernel
__global__ __launch_bounds__(64) void bool_count_sum_kernel(unsigned long long *set_bits) {
    unsigned long long v = threadIdx.x < BOOL_COUNT_SLOTS ? set_bits[threadIdx.x * BOOL_COUNT_PITCH] : 0;
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
    if (threadIdx.x == 0) set_bits[0] = v;
}

// ---------------------------------------------------------------------------------------------
// Entry points for a stream that came without any.
//
// Where the control bytes lie follows only from the start of the stream, one run after the other
// (boolean_rle.rs:40-62) -- but a run is at most 129 stream bytes (a control byte and 128
// literals), so the first control byte at or after any position p lies at one of p .. p + 128.
// The stream is cut into segments of BC_SEG bytes; for segment k and each of those 129 candidate
// offsets c, byte_rle_exits_kernel walks the chain from k * BC_SEG + c to where it first leaves
// the segment: exit[k][c], again 0..128, relative to the next segment.  The true offset of segment
// k + 1 is exit[k][true offset of segment k], and the true offset of segment 0 is 0: a chain of
// table look-ups, which is a prefix "sum" under composition of 129-entry tables and is taken in
// the usual two sweeps, BC_FAN tables at a time (compose_up builds the table of a group of
// tables, compose_down walks a group from its known start).  Nothing here is a guess.
// ---------------------------------------------------------------------------------------------

constexpr uint32_t BC_CAND = 129;
constexpr uint32_t BC_STAGE = BC_SEG + 160;  // a segment, the 128 bytes a walk may run past it, the load skew

// tab[k * BC_PITCH + c], k < n_tab: every segment but the last, whose exits nobody needs.
// A walk is a chain of dependent LDS reads, so the bytes are first turned into hop lengths in
// place (2 for a run, 1 + n for n literals) and a second array takes the length of two hops in a
// row (less 4, to fit a byte): a walk then moves two runs per step while that keeps it inside the
// segment.
__global__ __launch_bounds__(256) void byte_rle_exits_kernel(const uint8_t *__restrict__ s, uint64_t len,
                                                             uint32_t n_tab, uint8_t *__restrict__ tab) {
    __shared__ __attribute__((aligned(16))) uint8_t hop_all[4][BC_STAGE];
    __shared__ __attribute__((aligned(16))) uint8_t hop2_all[4][BC_SEG + 32];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t k = blockIdx.x * 4 + wave;
    if (k >= n_tab) return;  // whole wavefront; no workgroup barrier below
    uint8_t *hop = hop_all[wave], *hop2 = hop2_all[wave];
    const uint64_t base = (uint64_t)k * BC_SEG;
    // whole aligned 16-byte chunks from the one that holds byte `base` on; a chunk that starts
    // at or after the end of the stream is not loaded (its bytes read as zero: never on the chain
    // of a stream that is in order, and harmless on any other)
    const uint32_t skew = (uint32_t)((uintptr_t)(s + base) & 15);
    const uint8_t *src = s + base - skew;
    uint4 loaded[3];  // (all loads in flight before the first is used)
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const uint32_t c = lane + 64u * r;
        loaded[r] = make_uint4(0, 0, 0, 0);
        if (c < BC_STAGE / 16 && base + 16ull * c < len + skew) loaded[r] = ((const uint4 *)src)[c];
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const uint32_t c = lane + 64u * r;
        if (c >= BC_STAGE / 16) break;
        const uint4 v = loaded[r];
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t h = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t ctl = (w[q] >> (8 * e)) & 255u;
                h |= (ctl < 128 ? 2u : 257u - ctl) << (8 * e);
            }
            w[q] = h;
        }
        ((uint4 *)hop)[c] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t limit = skew + BC_SEG;  // (at most 15 + 2048: hop2 has room for it)
    for (uint32_t p4 = 4 * lane; p4 < limit; p4 += 256) {
        const uint32_t h = *(const uint32_t *)(hop + p4);
        uint32_t two = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t h1 = (h >> (8 * e)) & 255u;
            two |= (h1 + hop[p4 + e + h1] - 4u) << (8 * e);
        }
        *(uint32_t *)(hop2 + p4) = two;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // three walks per lane: candidates lane, lane + 64 and (lane 0 only) 128
    uint32_t p0 = skew + lane, p1 = skew + lane + 64, p2 = lane == 0 ? skew + 128 : limit;
    auto step = [&](uint32_t &p) {
        if (p < limit) {
            const uint32_t one = hop[p], two = hop2[p] + 4u;
            p += p + two < limit ? two : one;
        }
    };
    while (__ballot(p0 < limit || p1 < limit || p2 < limit) != 0) {
        step(p0);
        step(p1);
        step(p2);
    }
    uint8_t *row = tab + (uint64_t)k * BC_PITCH;
    row[lane] = (uint8_t)(p0 - limit);
    row[lane + 64] = (uint8_t)(p1 - limit);
    if (lane == 0) row[128] = (uint8_t)(p2 - limit);
}

// out[o] = in[o * BC_FAN + BC_FAN - 1] o ... o in[o * BC_FAN]: a workgroup per group, which brings
// the group's tables into LDS, then a thread per candidate walks them.
__global__ __launch_bounds__(256) void byte_rle_compose_up_kernel(const uint8_t *__restrict__ in, uint32_t n_in,
                                                                  uint8_t *__restrict__ out, uint32_t n_out) {
    __shared__ __attribute__((aligned(16))) uint8_t tabs[BC_FAN * BC_PITCH];
    const uint64_t o = blockIdx.x;
    const uint64_t first = o * BC_FAN;
    const uint32_t m = (uint32_t)(n_in - first < BC_FAN ? n_in - first : BC_FAN);
    const uint32_t *src = (const uint32_t *)(in + first * BC_PITCH);  // (BC_PITCH is a multiple of 4)
    for (uint32_t i = threadIdx.x; i < m * (BC_PITCH / 4); i += 256) ((uint32_t *)tabs)[i] = src[i];
    __syncthreads();
    if (threadIdx.x >= BC_CAND) return;
    uint32_t x = threadIdx.x;
    for (uint32_t i = 0; i < m; i++) x = tabs[i * BC_PITCH + x];
    out[o * BC_PITCH + threadIdx.x] = (uint8_t)x;
}

// A wavefront per group of BC_FAN tables: from the group's true offset (start_in[o]; null: 0, the
// one group at the top) to the true offset of each of its members: start_out[idx] (nullable), and
// at the level of the segments themselves entries[idx] = idx * BC_SEG + offset, with entries[n]
// from the last table's exit.  The tables come into LDS, lane 0 walks them, all lanes write.
__global__ __launch_bounds__(64) void byte_rle_compose_down_kernel(const uint8_t *__restrict__ tab, uint32_t n,
                                                                   const uint8_t *__restrict__ start_in,
                                                                   uint8_t *__restrict__ start_out,
                                                                   uint64_t *__restrict__ entries, uint64_t len) {
    __shared__ __attribute__((aligned(16))) uint8_t tabs[BC_FAN * BC_PITCH];
    __shared__ uint8_t offs[BC_FAN + 1];
    const uint64_t o = blockIdx.x;
    const uint64_t first = o * BC_FAN;
    const uint32_t m = (uint32_t)(n - first < BC_FAN ? n - first : BC_FAN);
    const uint32_t *src = (const uint32_t *)(tab + first * BC_PITCH);
    for (uint32_t i = threadIdx.x; i < m * (BC_PITCH / 4); i += 64) ((uint32_t *)tabs)[i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t x = start_in ? start_in[o] : 0u;
        for (uint32_t i = 0; i < m; i++) {
            offs[i] = (uint8_t)x;
            x = tabs[i * BC_PITCH + x];
        }
        offs[m] = (uint8_t)x;
    }
    __syncthreads();
    const uint32_t i = threadIdx.x;
    if (i < m) {
        if (start_out) start_out[first + i] = offs[i];
        if (entries) {
            const uint64_t e = (first + i) * BC_SEG + offs[i];
            entries[first + i] = e < len ? e : len;
        }
    }
    if (entries && i == 0 && first + m == n) {
        const uint64_t e = (uint64_t)n * BC_SEG + offs[m];
        entries[n] = e < len ? e : len;
    }
}

size_t byte_rle_compose_bytes(uint64_t len, uint32_t *n_entries) {
    const uint64_t k = (len + BC_SEG - 1) / BC_SEG;
    *n_entries = (uint32_t)k;
    size_t total = 0;
    for (uint64_t n = k - 1; n > 0; n = n > BC_FAN ? (n + BC_FAN - 1) / BC_FAN : 0)
        total += (size_t)((n * BC_PITCH + n + 255) / 256 * 256 + 256);  // a level's tables and offsets
    return total;
}

hipError_t launch_byte_rle_compose_entries(hipStream_t st, const uint8_t *stream, uint64_t len, uint8_t *work,
                                           uint64_t *entries) {
    const uint64_t k = (len + BC_SEG - 1) / BC_SEG;
    if (k < 2) return hipErrorInvalidValue;
    // the levels: tables of n[l] elements at tab[l], their true offsets at off[l]
    uint32_t n[8];
    uint8_t *tab[8], *off[8];
    int levels = 0;
    uint8_t *p = work;
    for (uint64_t m = k - 1; m > 0 && levels < 8; m = m > BC_FAN ? (m + BC_FAN - 1) / BC_FAN : 0) {
        n[levels] = (uint32_t)m;
        tab[levels] = p;
        off[levels] = p + (size_t)m * BC_PITCH;
        p += (size_t)((m * BC_PITCH + m + 255) / 256 * 256 + 256);
        levels++;
    }
    byte_rle_exits_kernel<<<(n[0] + 3) / 4, 256, 0, st>>>(stream, len, n[0], tab[0]);
    for (int l = 1; l < levels; l++)
        byte_rle_compose_up_kernel<<<n[l], 256, 0, st>>>(tab[l - 1], n[l - 1], tab[l], n[l]);
    for (int l = levels - 1; l >= 0; l--) {
        const uint64_t groups = ((uint64_t)n[l] + BC_FAN - 1) / BC_FAN;
        byte_rle_compose_down_kernel<<<(uint32_t)groups, 64, 0, st>>>(
            tab[l], n[l], l + 1 < levels ? off[l + 1] : nullptr, l > 0 ? off[l] : nullptr,
            l == 0 ? entries : nullptr, len);
    }
    return hipGetLastError();
}

constexpr uint64_t BR_MAX_BLOCKS = 256 * 16;  // workgroups of the expansion: 16 per CU
// The boolean expansion: 256 per CU, i.e. a wavefront takes about two tiles of the largest columns
// (measured on 4*10^9 values: 1.05 ms with 16 per CU, 0.98 with 128, 0.945 with 256, 0.953 with a
// workgroup per four tiles; one atomic per wavefront instead of per workgroup, all on one counter,
// made the larger grids three to six times slower).
constexpr uint64_t BOOL_MAX_BLOCKS = 256 * 256;

constexpr uint32_t BYTE_RLE_STAGED_MAX_SEGMENTS = 16384;
constexpr uint32_t BYTE_RLE_SLICE_BYTES = 48 << 10;
constexpr uint64_t BYTE_RLE_COOP_MIN_BYTES = 256;  // average segment from which a wavefront per segment pays

hipError_t launch_byte_rle_count(hipStream_t st, const uint8_t *stream, uint64_t len,
                                 const uint64_t *entries, const uint64_t *seg_ends, uint32_t n_seg,
                                 const SegTable &t, uint64_t *staged, uint32_t staged_cap) {
    // Few segments: what the pass costs is the longest walk, not the stream's bytes -- out of LDS
    // (see byte_rle_count_staged_kernel), as many segments to a wavefront as fit its slice.  Many:
    // the lane-per-segment walk already runs at the rate memory delivers the lines it touches.
    if (n_seg <= BYTE_RLE_STAGED_MAX_SEGMENTS) {
        const uint64_t avg = len / n_seg;
        if (avg >= BYTE_RLE_COOP_MIN_BYTES) {  // long segments: a wavefront each (see byte_rle_coop_count_kernel)
            const uint64_t want = (avg * 2 + 256 + 4095) / 4096 * 4096;
            const uint32_t slice_bytes = (uint32_t)(want > BYTE_RLE_SLICE_BYTES ? BYTE_RLE_SLICE_BYTES : want);
            byte_rle_coop_count_kernel<<<n_seg, 64, slice_bytes + 32, st>>>(stream, len, entries, seg_ends, n_seg, t, staged,
                                                                            staged_cap, slice_bytes);
            byte_rle_publish_kernel<<<(n_seg + SEG_BLOCK - 1) / SEG_BLOCK, SEG_BLOCK, 0, st>>>(n_seg, t);
            return hipGetLastError();
        }
        uint32_t group = 1;
        for (uint32_t g : {64u, 32u, 16u, 8u, 4u, 2u})
            if (group == 1 && avg * g * 5 / 4 + 256 <= BYTE_RLE_SLICE_BYTES) group = g;
        const uint64_t want = (avg * group * 5 / 4 + 256 + 4095) / 4096 * 4096;
        const uint32_t slice_bytes = (uint32_t)(want > BYTE_RLE_SLICE_BYTES ? BYTE_RLE_SLICE_BYTES : want);
        byte_rle_count_staged_kernel<<<(n_seg + group - 1) / group, 64, slice_bytes + 32, st>>>(
            stream, len, entries, seg_ends, n_seg, t, staged, staged_cap, group, slice_bytes);
        byte_rle_publish_kernel<<<(n_seg + SEG_BLOCK - 1) / SEG_BLOCK, SEG_BLOCK, 0, st>>>(n_seg, t);
        return hipGetLastError();
    }
    byte_rle_scan_kernel<false><<<(n_seg + SEG_BLOCK - 1) / SEG_BLOCK, SEG_BLOCK, 0, st>>>(
        stream, len, entries, seg_ends, n_seg, t, nullptr, nullptr, 0, staged, staged_cap, 0);
    return hipGetLastError();
}

hipError_t launch_byte_rle_emit_staged(hipStream_t st, uint32_t n_seg, const SegTable &t,
                                       const uint64_t *staged, uint32_t staged_cap, ByteRun *runs,
                                       uint32_t *win_first, uint32_t n_win_slots, uint32_t max_seg_runs) {
    const uint32_t group = staged_group_lanes(max_seg_runs);
    const uint64_t blocks = ((uint64_t)n_seg * group + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    byte_rle_emit_staged_kernel<<<(uint32_t)blocks, 256, 0, st>>>(n_seg, t, staged, staged_cap, runs, win_first, n_win_slots, group);
    return hipGetLastError();
}

hipError_t launch_byte_rle_emit(hipStream_t st, const uint8_t *stream, uint64_t len,
                                const uint64_t *entries, const uint64_t *seg_ends, uint32_t n_seg,
                                const SegTable &t, ByteRun *runs, uint32_t *win_first,
                                uint32_t n_win_slots, uint32_t staged_up_to) {
    byte_rle_scan_kernel<true><<<(n_seg + SEG_BLOCK - 1) / SEG_BLOCK, SEG_BLOCK, 0, st>>>(
        stream, len, entries, seg_ends, n_seg, t, runs, win_first, n_win_slots, nullptr, 0, staged_up_to);
    return hipGetLastError();
}

// out[i] = number of non-zero bytes among bytes[start[i] .. start[i] + count[i]) (the set bits of
// every column of a batch of boolean streams).  A slice is shared out over gridDim.y workgroups,
// 16 bytes per lane and load, each adding its part to the slice's count (zeroed by the launcher):
// one workgroup per slice walked a two-megabyte column in two thousand dependent steps, 0.8 ms for
// what is 50 microseconds of memory traffic.
__global__ __launch_bounds__(256) void count_nonzero_slices_kernel(
    const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ start,
    const uint64_t *__restrict__ count, unsigned long long *__restrict__ out) {
    __shared__ uint64_t partial[4];
    const uint8_t *p = bytes + start[blockIdx.x];
    const uint64_t n = count[blockIdx.x];
    // this workgroup's share: whole 16-byte chunks of the aligned middle; the ragged ends go to
    // the first workgroup
    const uint64_t head = std::min<uint64_t>(n, (16 - ((uintptr_t)p & 15)) & 15);
    const uint64_t chunks = (n - head) / 16;
    const uint64_t per = (chunks + gridDim.y - 1) / gridDim.y;
    const uint64_t c0 = std::min<uint64_t>(chunks, per * blockIdx.y), c1 = std::min<uint64_t>(chunks, c0 + per);
    uint64_t c = 0;
    const uint4 *q = (const uint4 *)(p + head);
    auto nz = [](uint32_t d) { return (uint32_t)__popc(((d | ((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu)) >> 7) & 0x01010101u); };
    for (uint64_t i = c0 + threadIdx.x; i < c1; i += blockDim.x) {
        const uint4 v = q[i];
        c += nz(v.x) + nz(v.y) + nz(v.z) + nz(v.w);
    }
    if (blockIdx.y == 0) {
        for (uint64_t i = threadIdx.x; i < head; i += blockDim.x) c += p[i] != 0;
        for (uint64_t i = head + chunks * 16 + threadIdx.x; i < n; i += blockDim.x) c += p[i] != 0;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = __shfl_down((uint32_t)c, d), hi = __shfl_down((uint32_t)(c >> 32), d);
        c += ((uint64_t)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint64_t t = partial[0] + partial[1] + partial[2] + partial[3];
        if (t) atomicAdd(&out[blockIdx.x], (unsigned long long)t);
    }
}

hipError_t launch_count_nonzero_slices(hipStream_t st, const uint8_t *bytes, const uint64_t *start,
                                       const uint64_t *count, uint32_t n_slices, uint64_t *out,
                                       uint64_t largest) {
    if (n_slices == 0) return hipSuccess;
    const hipError_t e = hipMemsetAsync(out, 0, (size_t)n_slices * sizeof(uint64_t), st);
    if (e != hipSuccess) return e;
    // (a workgroup per 64 KiB of the largest slice, at most 256 per slice)
    const uint32_t parts = (uint32_t)std::min<uint64_t>(256, std::max<uint64_t>(1, largest >> 16));
    count_nonzero_slices_kernel<<<dim3(n_slices, parts), 256, 0, st>>>(bytes, start, count, (unsigned long long *)out);
    return hipGetLastError();
}

hipError_t launch_byte_rle_expand(hipStream_t st, const uint8_t *stream, uint64_t len,
                                  const ByteRun *runs, const uint32_t *win_first, uint32_t n_win_slots,
                                  uint32_t n_runs, uint64_t num_bytes, uint8_t *out) {
    if (num_bytes == 0) return hipSuccess;
    if ((((uintptr_t)out) & 15) == 0 && n_win_slots > 0) {
        const uint64_t tiles = (num_bytes + BX_TILE - 1) / BX_TILE;  // one per wavefront, four per workgroup
        byte_rle_tile_kernel<<<(uint32_t)std::min<uint64_t>((tiles + 3) / 4, BR_MAX_BLOCKS), 256, 0, st>>>(
            stream, len, runs, win_first, n_win_slots, n_runs, num_bytes, out);
        return hipGetLastError();
    }
    const uint64_t lanes = (num_bytes + 15) / 16;
    byte_rle_expand_kernel<<<(uint32_t)std::min<uint64_t>((lanes + 255) / 256, BR_MAX_BLOCKS), 256, 0, st>>>(
        stream, len, runs, win_first, n_runs, num_bytes, out);
    return hipGetLastError();
}

hipError_t launch_bool_expand(hipStream_t st, const uint8_t *stream, uint64_t len,
                              const ByteRun *runs, const uint32_t *win_first, uint32_t n_win_slots,
                              uint32_t n_runs, uint64_t num_bits, uint8_t *out_bytes,
                              uint8_t *out_bitmap, uint64_t *set_bits) {
    const uint64_t n_bytes = (num_bits + 7) >> 3;
    if (n_bytes == 0) return hipSuccess;
    const uint64_t tiles = (n_bytes + BOOL_TILE - 1) / BOOL_TILE;  // one per wavefront, four per workgroup
    if (set_bits) {
        const hipError_t e = hipMemsetAsync(set_bits, 0, BOOL_COUNT_BYTES, st);
        if (e != hipSuccess) return e;
    }
    bool_expand_kernel<<<(uint32_t)std::min<uint64_t>((tiles + 3) / 4, BOOL_MAX_BLOCKS), 256, 0, st>>>(
        stream, len, runs, win_first, n_win_slots, n_runs, num_bits, out_bytes, out_bitmap,
        (unsigned long long *)set_bits);
    if (set_bits) bool_count_sum_kernel<<<1, 64, 0, st>>>((unsigned long long *)set_bits);
    return hipGetLastError();
}

}  // namespace orcd

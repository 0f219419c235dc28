// Byte-RLE and boolean expansion for gfx950 (PRESENT and boolean DATA streams).
//
// Byte-RLE as in src/read/decode/boolean_rle.rs:28-62 of the orc-format crate: a control byte
// c >= 0 announces c + 3 copies of the next byte, c < 0 announces -c literal bytes; a boolean
// stream carries 8 values per byte, most significant bit first (boolean_rle.rs:98,118).  Run and
// literal lengths are exactly those of the ORC specification.  Same two-phase shape as the RLE v2
// decoder: control bytes form a serial chain walked per segment, then the bytes are expanded.
#include "orcd_internal.h"
#include "seg_scan.h"

namespace orcd {

constexpr uint64_t BR_IS_RUN = 1ull << 48;

template <bool EMIT>
__global__ __launch_bounds__(SEG_BLOCK) void byte_rle_scan_kernel(
    const uint8_t *__restrict__ s, uint64_t len, const uint64_t *__restrict__ entries,
    uint32_t n_seg, SegTable t, ByteRun *__restrict__ runs) {
    const uint32_t seg = blockIdx.x * SEG_BLOCK + threadIdx.x;
    const bool valid = seg < n_seg;
    if (EMIT && !valid) return;
    uint64_t pos = (valid && entries && seg > 0) ? entries[seg] : 0;
    const uint64_t end = (valid && entries && seg + 1 < n_seg) ? entries[seg + 1] : len;
    const bool last_seg = seg + 1 == n_seg;
    if (pos > len) pos = len;
    uint32_t n_runs = 0, status = SEG_OK, last_len = 0;
    uint64_t n_bytes = 0;
    uint64_t out_off = 0;
    uint32_t run_idx = 0, want_runs = 0xFFFFFFFFu;
    if (EMIT) {
        const SegOffset so = seg_global_offset(t, seg);
        out_off = so.val_off;
        run_idx = so.run_off;
        want_runs = t.seg_count[seg].n_runs;
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
        if (next > len) { status = SEG_TRUNCATED; break; }
        if (pos + 1 > PAYLOAD_MASK) { status = SEG_TOO_LONG; break; }
        if (EMIT) {
            ByteRun r;
            r.src = src | ((uint64_t)(n - 1) << 40);
            r.out_off = out_off;
            runs[run_idx++] = r;
            out_off += n;
        }
        n_runs++;
        n_bytes += n;
        last_len = n;
        pos = next;
    }
    if (!EMIT) {
        if (status == SEG_OK && pos != end && !last_seg) status = SEG_MISALIGNED;
        seg_publish(t, seg, n_seg, n_runs, last_len, status, n_bytes);
    }
}

// One lane per run copies the run's bytes (at most 130).
__global__ __launch_bounds__(256) void byte_rle_expand_kernel(const uint8_t *__restrict__ s,
                                                              const ByteRun *__restrict__ runs,
                                                              uint64_t n_runs, uint64_t num_bytes,
                                                              uint8_t *__restrict__ out) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_runs) return;
    const ByteRun run = runs[r];
    if (run.out_off >= num_bytes) return;
    const uint64_t src = run.src & PAYLOAD_MASK;
    uint32_t n = (uint32_t)((run.src >> 40) & 0xFF) + 1;
    if (run.out_off + n > num_bytes) n = (uint32_t)(num_bytes - run.out_off);
    uint8_t *dst = out + run.out_off;
    if (run.src & BR_IS_RUN) {
        const uint8_t v = s[src];
        for (uint32_t i = 0; i < n; i++) dst[i] = v;
    } else {
        for (uint32_t i = 0; i < n; i++) dst[i] = s[src + i];
    }
}

// Packed bytes -> one byte per value and/or an LSB-first bitmap, plus the number of ones.
__global__ __launch_bounds__(256) void bool_expand_kernel(const uint8_t *__restrict__ packed,
                                                          uint64_t num_bits,
                                                          uint8_t *__restrict__ out_bytes,
                                                          uint8_t *__restrict__ out_bitmap,
                                                          unsigned long long *__restrict__ set_bits) {
    const uint64_t n_bytes = (num_bits + 7) >> 3;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t ones = 0;
    if (i < n_bytes) {
        uint32_t b = packed[i];
        const uint64_t first = i << 3;
        const uint32_t valid = (num_bits - first) >= 8 ? 8u : (uint32_t)(num_bits - first);
        b &= (0xFF00u >> valid) & 0xFFu;  // keep the `valid` most significant bits
        ones = __popc(b);
        if (out_bitmap) out_bitmap[i] = (uint8_t)(__brev(b) >> 24);
        if (out_bytes) {
            // byte k of the word = bit (7 - k) of b
            uint64_t word = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) word |= (uint64_t)((b >> (7 - k)) & 1u) << (8 * k);
            uint8_t *dst = out_bytes + first;
            if (valid == 8 && (((uintptr_t)dst) & 7) == 0) {
                *(uint64_t *)dst = word;
            } else {
                for (uint32_t k = 0; k < valid; k++) dst[k] = (uint8_t)(word >> (8 * k));
            }
        }
    }
    if (set_bits) {
        for (int d = 32; d >= 1; d >>= 1) ones += __shfl_down(ones, d);
        if ((threadIdx.x & 63) == 0 && ones) atomicAdd(set_bits, (unsigned long long)ones);
    }
}

hipError_t launch_byte_rle_count(hipStream_t st, const uint8_t *stream, uint64_t len,
                                 const uint64_t *entries, uint32_t n_seg, const SegTable &t) {
    byte_rle_scan_kernel<false><<<(n_seg + SEG_BLOCK - 1) / SEG_BLOCK, SEG_BLOCK, 0, st>>>(
        stream, len, entries, n_seg, t, nullptr);
    return hipGetLastError();
}

hipError_t launch_byte_rle_emit(hipStream_t st, const uint8_t *stream, uint64_t len,
                                const uint64_t *entries, uint32_t n_seg, const SegTable &t,
                                ByteRun *runs) {
    byte_rle_scan_kernel<true><<<(n_seg + SEG_BLOCK - 1) / SEG_BLOCK, SEG_BLOCK, 0, st>>>(
        stream, len, entries, n_seg, t, runs);
    return hipGetLastError();
}

hipError_t launch_byte_rle_expand(hipStream_t st, const uint8_t *stream, const ByteRun *runs,
                                  uint64_t n_runs, uint64_t num_bytes, uint8_t *out) {
    if (n_runs == 0) return hipSuccess;
    byte_rle_expand_kernel<<<(uint32_t)((n_runs + 255) / 256), 256, 0, st>>>(stream, runs, n_runs,
                                                                            num_bytes, out);
    return hipGetLastError();
}

hipError_t launch_bool_expand(hipStream_t st, const uint8_t *packed, uint64_t num_bits,
                              uint8_t *out_bytes, uint8_t *out_bitmap, uint64_t *set_bits) {
    const uint64_t n_bytes = (num_bits + 7) >> 3;
    if (n_bytes == 0) return hipSuccess;
    bool_expand_kernel<<<(uint32_t)((n_bytes + 255) / 256), 256, 0, st>>>(
        packed, num_bits, out_bytes, out_bitmap, (unsigned long long *)set_bits);
    return hipGetLastError();
}

}  // namespace orcd

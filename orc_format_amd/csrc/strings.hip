// String column support for gfx950: device-wide prefix sum of LENGTH values into offsets,
// UTF-8 validation, dictionary gather.
//
// Values::next(len) of the orc-format crate (src/read/decode/variable_length.rs:15-23) hands out
// consecutive slices of the DATA stream, i.e. value i starts at the exclusive prefix sum of the
// lengths; tests/it/deserialize.rs:106-140 builds one String per value and validates UTF-8.  Here
// the column is produced as offsets[n + 1] plus the contiguous bytes.
#include <algorithm>

#include "orcd_internal.h"

namespace orcd {

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));


// ---------------------------------------------------------------------------------------------
// exclusive scan of 64-bit values, in place: reduce per tile, scan the tile sums, rescan tiles
// ---------------------------------------------------------------------------------------------

constexpr int SCAN_WAVES = 4;
constexpr int ITEMS_PER_LANE = 16;                       // 8 steps of 2
constexpr uint64_t WAVE_ITEMS = 64 * ITEMS_PER_LANE;     // 1024 consecutive values per wavefront
constexpr uint64_t TILE_ITEMS = SCAN_WAVES * WAVE_ITEMS; // 4096 per workgroup

size_t scan_block_sums_len(uint64_t n) { return (size_t)((n + TILE_ITEMS - 1) / TILE_ITEMS) + 1; }

__device__ __forceinline__ uint64_t wave_sum(uint64_t v) {
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, d), hi = __shfl_xor((uint32_t)(v >> 32), d);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

__device__ __forceinline__ uint64_t lane63_u64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, 63);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), 63);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t sat_add(uint64_t a, uint64_t b) {
    const uint64_t s = a + b;
    return s < a ? ~0ull : s;
}

// inclusive wave scan under saturating addition (associative on unsigned values)
__device__ __forceinline__ uint64_t wave_inclusive_sat_sum(uint64_t v, int lane) {
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t lo = __shfl_up((uint32_t)v, d), hi = __shfl_up((uint32_t)(v >> 32), d);
        if (lane >= d) v = sat_add(v, ((uint64_t)hi << 32) | lo);
    }
    return v;
}

__device__ __forceinline__ uint64_t wave_inclusive_sum(uint64_t v, int lane) {
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t lo = __shfl_up((uint32_t)v, d), hi = __shfl_up((uint32_t)(v >> 32), d);
        if (lane >= d) v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

__global__ __launch_bounds__(SCAN_WAVES * 64) void scan_reduce_kernel(
    const uint64_t *__restrict__ data, uint64_t n, uint64_t *__restrict__ tile_sums,
    uint64_t limit, uint32_t *__restrict__ over_flag) {
    __shared__ uint64_t wsum[SCAN_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * TILE_ITEMS + (uint64_t)wave * WAVE_ITEMS;
    uint64_t acc = 0, biggest = 0;
#pragma unroll
    for (int j = 0; j < ITEMS_PER_LANE / 2; j++) {
        const uint64_t i = base + 128u * j + 2u * lane;
        if (i + 1 < n) {
            const ulonglong2 v = *(const ulonglong2 *)(data + i);
            acc += v.x + v.y;
            biggest = max(biggest, max(v.x, v.y));
        } else if (i < n) {
            const uint64_t x = data[i];
            acc += x;
            biggest = max(biggest, x);
        }
    }
    if (biggest > limit) atomicOr(over_flag, 1u);
    acc = wave_sum(acc);
    if (lane == 0) wsum[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < SCAN_WAVES; w++) t += wsum[w];
        tile_sums[blockIdx.x] = t;
    }
}

// exclusive scan of the tile sums by one workgroup; tile_sums[n_tiles] receives the grand total.
// Sums saturate, so a column whose lengths add up past 2^64 reports UINT64_MAX, not a small total.
__global__ __launch_bounds__(1024) void scan_tile_sums_kernel(uint64_t *__restrict__ tile_sums,
                                                              uint64_t n_tiles) {
    __shared__ uint64_t wtot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t carry = 0;
    for (uint64_t base = 0; base < n_tiles; base += 1024) {
        const uint64_t i = base + threadIdx.x;
        const uint64_t v = i < n_tiles ? tile_sums[i] : 0;
        const uint64_t inc = wave_inclusive_sat_sum(v, lane);
        uint64_t exc = __shfl_up((uint32_t)inc, 1) | ((uint64_t)__shfl_up((uint32_t)(inc >> 32), 1) << 32);
        if (lane == 0) exc = 0;
        __syncthreads();
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint64_t prefix = 0, total = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wave) prefix = sat_add(prefix, wtot[w]);
            total = sat_add(total, wtot[w]);
        }
        if (i < n_tiles) tile_sums[i] = sat_add(sat_add(carry, prefix), exc);
        carry = sat_add(carry, total);
    }
    if (threadIdx.x == 0) tile_sums[n_tiles] = carry;
}

__global__ __launch_bounds__(SCAN_WAVES * 64) void scan_apply_kernel(
    uint64_t *__restrict__ data, uint64_t n, const uint64_t *__restrict__ tile_sums,
    uint64_t n_tiles) {
    __shared__ uint64_t wsum[SCAN_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * TILE_ITEMS + (uint64_t)wave * WAVE_ITEMS;
    uint64_t a[ITEMS_PER_LANE / 2], b[ITEMS_PER_LANE / 2];
    uint64_t carry = 0;
    // wave-local exclusive prefix sums, kept in registers
#pragma unroll
    for (int j = 0; j < ITEMS_PER_LANE / 2; j++) {
        const uint64_t i = base + 128u * j + 2u * lane;
        uint64_t x = 0, y = 0;
        if (i + 1 < n) {
            const ulonglong2 v = *(const ulonglong2 *)(data + i);
            x = v.x;
            y = v.y;
        } else if (i < n) {
            x = data[i];
        }
        const uint64_t inc = wave_inclusive_sum(x + y, lane);
        a[j] = carry + inc - (x + y);
        b[j] = a[j] + x;
        carry += lane63_u64(inc);
    }
    if (lane == 0) wsum[wave] = carry;
    __syncthreads();
    uint64_t prefix = tile_sums[blockIdx.x];
    for (int w = 0; w < wave; w++) prefix += wsum[w];
#pragma unroll
    for (int j = 0; j < ITEMS_PER_LANE / 2; j++) {
        const uint64_t i = base + 128u * j + 2u * lane;
        if (i + 1 < n) {
            *(ulonglong2 *)(data + i) = make_ulonglong2(prefix + a[j], prefix + b[j]);
        } else if (i < n) {
            data[i] = prefix + a[j];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) data[n] = tile_sums[n_tiles];
}

hipError_t launch_exclusive_scan_u64(hipStream_t st, uint64_t *data, uint64_t n,
                                     uint64_t *block_sums, uint64_t limit, uint32_t *over_flag) {
    const uint64_t n_tiles = (n + TILE_ITEMS - 1) / TILE_ITEMS;
    if (n_tiles == 0) {
        hipError_t e = hipMemsetAsync(data, 0, sizeof(uint64_t), st);
        return e;
    }
    scan_reduce_kernel<<<(uint32_t)n_tiles, SCAN_WAVES * 64, 0, st>>>(data, n, block_sums, limit,
                                                                            over_flag);
    scan_tile_sums_kernel<<<1, 1024, 0, st>>>(block_sums, n_tiles);
    scan_apply_kernel<<<(uint32_t)n_tiles, SCAN_WAVES * 64, 0, st>>>(data, n, block_sums, n_tiles);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// String bytes: copy and UTF-8 validation in one pass
// (std::str::from_utf8 acceptance, tests/it/deserialize.rs:116-118)
//
// Every value is well formed iff the concatenated bytes are well formed and no value starts on a
// continuation byte.  Well-formedness of the concatenation is decided position by position from a
// byte and its three predecessors:
//   * a byte is a continuation byte (10xxxxxx) exactly when one is owed -- the previous byte is a
//     lead (>= C0), or the one before is a 3/4-byte lead (>= E0), or the one before that a 4-byte
//     lead (>= F0);
//   * C0, C1 and F5..FF never occur;
//   * after E0 the next byte is >= A0, after ED < A0, after F0 >= 90, after F4 < 90.
// One more position just past the last byte, holding a non-continuation byte, catches a sequence
// cut short by the end of the data.
//
// A workgroup takes a tile of 64 KiB of 16-byte chunks (aligned to the source address), each lane
// a chunk at a time as four little-endian dwords examined with byte-parallel arithmetic; then the
// workgroup looks up which values start inside its tile (two 64-ary searches of the offsets) and
// tests their first bytes, which it has just pulled into the cache.
// ---------------------------------------------------------------------------------------------

constexpr uint32_t SB_THREADS = 256;
constexpr uint32_t SB_CHUNKS_PER_THREAD = 16;
constexpr uint32_t SB_TILE_CHUNKS = SB_THREADS * SB_CHUNKS_PER_THREAD;  // 64 KiB
constexpr uint32_t HI = 0x80808080u;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// per byte: x >= c, for c >= 0x80; result in bit 7 of each byte
__device__ __forceinline__ uint32_t bytes_ge(uint32_t x, uint32_t c) {
    return x & ((x | HI) - ((c & 0x7Fu) * 0x01010101u)) & HI;
}

// Errors (bit 7 of the offending bytes) among the 4 bytes of `d`, whose predecessors are the
// top bytes of `before` (the dword in front of it).
__device__ __forceinline__ uint32_t utf8_errors(uint32_t d, uint32_t before) {
    const uint32_t p1 = __builtin_amdgcn_alignbyte(d, before, 3);
    const uint32_t p2 = __builtin_amdgcn_alignbyte(d, before, 2);
    const uint32_t p3 = __builtin_amdgcn_alignbyte(d, before, 1);
    const uint32_t cont = d & ~(d << 1) & HI;
    const uint32_t owed = (p1 & (p1 << 1)) | (p2 & (p2 << 1) & (p2 << 2)) |
                          (p3 & (p3 << 1) & (p3 << 2) & (p3 << 3));
    uint32_t err = (owed ^ cont) & HI;
    err |= (d & (d << 1) & HI) & ~bytes_ge(d, 0xC2);  // C0, C1
    if ((d | p1) & ((d | p1) << 1) & ((d | p1) << 2) & HI) {  // some byte or predecessor >= E0
        err |= bytes_ge(d, 0xF5);
        const uint32_t b5 = d << 2, b54 = (d << 2) | (d << 3);  // bit 5, bit 5 or 4 -> bit 7
        const uint32_t e0 = bytes_ge(p1, 0xE0) & ~bytes_ge(p1, 0xE1);
        const uint32_t ed = bytes_ge(p1, 0xED) & ~bytes_ge(p1, 0xEE);
        const uint32_t f0 = bytes_ge(p1, 0xF0) & ~bytes_ge(p1, 0xF1);
        const uint32_t f4 = bytes_ge(p1, 0xF4) & ~bytes_ge(p1, 0xF5);
        err |= (e0 & ~b5) | (ed & b5) | (f0 & ~b54) | (f4 & b54);
    }
    return err;
}

// dword whose bytes k (data index idx0 + k) outside [0, total) are cleared
__device__ __forceinline__ uint32_t clip_dword(uint32_t d, int64_t idx0, uint64_t total) {
    uint32_t m = 0;
    for (int k = 0; k < 4; k++)
        if (idx0 + k >= 0 && (uint64_t)(idx0 + k) < total) m |= 0xFFu << (8 * k);
    return d & m;
}

// first i in [0, n_off) with offsets[i] >= key (n_off if none); all 64 lanes take part
__device__ __forceinline__ uint64_t wave_lower_bound(const int64_t *__restrict__ offsets,
                                                     uint64_t n_off, uint64_t key, int lane) {
    uint64_t a = 0, b = n_off;  // offsets[i] < key for i < a; offsets[i] >= key for i >= b
    while (a < b) {
        const uint64_t step = (b - a + 63) / 64;
        const uint64_t p = a + (uint64_t)lane * step;
        const bool below = p < b && (uint64_t)offsets[p] < key;
        const uint32_t k = (uint32_t)__popcll(__ballot(below));  // the lanes below form a prefix
        if (k == 0) {
            b = a;
        } else {
            const uint64_t last_below = a + (uint64_t)(k - 1) * step;
            const uint64_t next_probe = last_below + step;
            b = (k < 64 && next_probe < b) ? next_probe : b;
            a = last_below + 1;
            if (b < a) b = a;
        }
    }
    return a;
}

template <bool COPY, bool CHECK>
__global__ __launch_bounds__(SB_THREADS) void string_bytes_kernel(
    const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const int64_t *__restrict__ offsets,
    uint64_t n, uint64_t total, uint32_t *__restrict__ invalid) {
    __shared__ uint64_t bounds[2];
    const uint32_t skew = (uint32_t)((uintptr_t)src & 15);
    const uint4 *chunks = (const uint4 *)(src - skew);  // chunk c holds data bytes 16c - skew ...
    uint4 *out_chunks = COPY ? (uint4 *)(dst - skew) : nullptr;  // dst has the same skew
    // workgroups that share an XCD (equal id modulo 8) walk one contiguous eighth of the tiles
    uint32_t tile = blockIdx.x;
    {
        const uint32_t nwg = gridDim.x, q = nwg / 8, r = nwg % 8, x = tile % 8;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + tile / 8;
    }
    const uint64_t c_begin = (uint64_t)tile * SB_TILE_CHUNKS;
    uint32_t err = 0, high = 0;
#pragma unroll 8
    for (uint32_t k = 0; k < SB_CHUNKS_PER_THREAD; k++) {
        const uint64_t c = c_begin + (uint64_t)k * SB_THREADS + threadIdx.x;
        const int64_t idx0 = (int64_t)(c * 16) - (int64_t)skew;  // data index of the chunk's byte 0
        // the chunk that holds index `total` is the last one with anything to do
        const bool live = idx0 <= (int64_t)total;
        const bool inside = idx0 >= 0 && (uint64_t)idx0 + 16 <= total;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (inside || (live && idx0 + 16 > 0 && (uint64_t)(idx0 < 0 ? 0 : idx0) < total)) {
            v = chunks[c];  // (a nontemporal load measured no faster)
        }
        if (COPY) {
            if (inside) {
                u32x4 t;
                t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
                __builtin_nontemporal_store(t, (u32x4 *)(out_chunks + c));
            } else if (live) {
                const uint32_t d[4] = {v.x, v.y, v.z, v.w};
                for (int j = 0; j < 16; j++)
                    if (idx0 + j >= 0 && (uint64_t)(idx0 + j) < total)
                        dst[idx0 + j] = (uint8_t)(d[j >> 2] >> (8 * (j & 3)));
            }
        }
        if (CHECK) {
            if (!inside) {  // bytes that are not data read as zero
                v.x = clip_dword(v.x, idx0, total);
                v.y = clip_dword(v.y, idx0 + 4, total);
                v.z = clip_dword(v.z, idx0 + 8, total);
                v.w = clip_dword(v.w, idx0 + 12, total);
            }
            // The dword in front of the chunk supplies the three predecessors of its first byte:
            // the neighbouring lane holds it, except for the first lane of the wavefront.
            uint32_t before = __shfl_up(v.w, 1);
            if ((threadIdx.x & 63) == 0) {
                before = 0;
                if (live && idx0 > 0)
                    before = clip_dword(((const uint32_t *)chunks)[c * 4 - 1], idx0 - 4, total);
            }
            high |= v.x | v.y | v.z | v.w;
            if (((v.x | v.y | v.z | v.w) | (before & 0xFFFFFF00u)) & HI) {  // not plain ASCII
                err |= utf8_errors(v.x, before) | utf8_errors(v.y, v.x) | utf8_errors(v.z, v.y) |
                       utf8_errors(v.w, v.z);
            }
        }
    }
    if (!CHECK) return;
    // Values that start inside this tile must not start on a continuation byte; a tile of plain
    // ASCII has none.
    if (!__syncthreads_or((high & HI) != 0)) {
        if (err) atomicOr(invalid, 1u);
        return;
    }
    const int64_t t0s = (int64_t)(c_begin * 16) - (int64_t)skew;
    const uint64_t t0 = t0s < 0 ? 0 : (uint64_t)t0s;
    uint64_t t1 = (uint64_t)(t0s + (int64_t)SB_TILE_CHUNKS * 16);
    if (t1 > total) t1 = total;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < 2) {
        const uint64_t r = wave_lower_bound(offsets, n + 1, wave == 0 ? t0 : t1, lane);
        if (lane == 0) bounds[wave] = r;
    }
    __syncthreads();
    const uint64_t lo = bounds[0], hi = bounds[1] < n ? bounds[1] : n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += SB_THREADS) {
        const uint64_t o = (uint64_t)offsets[i], e = (uint64_t)offsets[i + 1];
        if (e > o && o < total && (src[o] & 0xC0) == 0x80) err |= HI;
    }
    if (err) atomicOr(invalid, 1u);
}

hipError_t launch_string_bytes(hipStream_t st, const uint8_t *src, uint8_t *dst,
                               const int64_t *offsets, uint64_t n, uint64_t total,
                               uint32_t *invalid_flag) {
    if (total == 0 || (!dst && !invalid_flag)) return hipSuccess;
    if (dst == src) dst = nullptr;
    // the copy shares the source's 16-byte chunking: other placements go through the runtime's copy
    if (dst && (((uintptr_t)dst ^ (uintptr_t)src) & 15)) {
        hipError_t e = hipMemcpyAsync(dst, src, total, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return e;
        dst = nullptr;
        if (!invalid_flag) return hipSuccess;
    }
    const uint64_t skew = (uintptr_t)src & 15;
    const uint64_t n_chunks = (total + skew) / 16 + 1;  // up to the chunk that holds index `total`
    const uint64_t tiles = (n_chunks + SB_TILE_CHUNKS - 1) / SB_TILE_CHUNKS;
    if (tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (dst && invalid_flag)
        string_bytes_kernel<true, true><<<(uint32_t)tiles, SB_THREADS, 0, st>>>(src, dst, offsets, n, total, invalid_flag);
    else if (dst)
        string_bytes_kernel<true, false><<<(uint32_t)tiles, SB_THREADS, 0, st>>>(src, dst, offsets, n, total, invalid_flag);
    else
        string_bytes_kernel<false, true><<<(uint32_t)tiles, SB_THREADS, 0, st>>>(src, dst, offsets, n, total, invalid_flag);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// dictionary gather (tests/it/deserialize.rs:142-185)
// ---------------------------------------------------------------------------------------------

// Inclusive prefix sums over the 64 lanes with DPP row shifts and row broadcasts: shifts by 1, 2,
// 4, 8 scan each row of 16 lanes, then lane 15 of rows 0 and 2 is added to rows 1 and 3, then lane
// 31 to rows 2 and 3.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_inclusive_dpp(uint32_t v) {
    v += dpp_or_zero<0x111, 0xF>(v);
    v += dpp_or_zero<0x112, 0xF>(v);
    v += dpp_or_zero<0x114, 0xF>(v);
    v += dpp_or_zero<0x118, 0xF>(v);
    v += dpp_or_zero<0x142, 0xA>(v);
    v += dpp_or_zero<0x143, 0xC>(v);
    return v;
}

// chunk_sums[c] = bytes that values [64c, 64c + 64) hold together (saturating); the indices and
// the dictionary offsets are checked on the way.  The lengths themselves are never stored: the
// gather recomputes them and places every value by a wave scan on top of the scanned chunk sums.
__global__ __launch_bounds__(256) void dictionary_chunk_sums_kernel(
    const int64_t *__restrict__ dict_offsets, uint64_t dict_size, uint64_t dict_total,
    const int64_t *__restrict__ indices, uint64_t n, uint64_t *__restrict__ chunk_sums,
    uint32_t *__restrict__ error_flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    if ((i & ~(uint64_t)63) >= n) return;  // whole wavefront
    uint64_t l = 0;
    if (i < n) {
        const uint64_t idx = (uint64_t)indices[i];
        if (idx >= dict_size) {
            atomicOr(error_flag, 1u);  // deserialize.rs:162 -> OutOfSpec
        } else {
            const uint64_t b = (uint64_t)dict_offsets[idx], e = (uint64_t)dict_offsets[idx + 1];
            if (e < b || e > dict_total) atomicOr(error_flag, 2u);  // not offsets of dict_bytes
            else l = e - b;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)l, d), hi = __shfl_xor((uint32_t)(l >> 32), d);
        l = sat_add(l, ((uint64_t)hi << 32) | lo);
    }
    if (lane == 0) chunk_sums[i >> 6] = l;
}

// The same with the entries' lengths in LDS, for a dictionary whose offsets fit there: the two
// look-ups per value at random places of the offsets array otherwise keep the texture unit busy
// longer than the indices take to stream in.  Every workgroup builds the table once (checking the
// offsets as it does) and then takes chunks of 64 values in turn.
__global__ __launch_bounds__(256) void dictionary_chunk_sums_lds_kernel(
    const int64_t *__restrict__ dict_offsets, uint32_t dict_size, uint64_t dict_total,
    const int64_t *__restrict__ indices, uint64_t n, uint64_t *__restrict__ chunk_sums,
    uint32_t *__restrict__ error_flag) {
    extern __shared__ uint32_t l_len[];  // [dict_size]
    for (uint32_t k = threadIdx.x; k < dict_size; k += blockDim.x) {
        const uint64_t b = (uint64_t)dict_offsets[k], e = (uint64_t)dict_offsets[k + 1];
        uint32_t l = 0;
        if (e < b || e > dict_total) atomicOr(error_flag, 2u);  // not offsets of dict_bytes
        else l = (uint32_t)(e - b);  // dict_total is below 2^32 here
        l_len[k] = l;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t n_chunks = (n + 63) / 64;
    // A wavefront takes four pairs of chunks at a time, lane i values 2i and 2i + 1 of each pair:
    // one 16-byte load per lane and pair, four in flight, and a pair's two sums by one DPP scan
    // that stops short of the step across the wavefront's halves -- lane 31 then holds the first
    // chunk's sum, lane 63 the second's.  (64 lengths below 2^32 each: the sum is done in two
    // 16-bit halves that cannot wrap.)
    const uint64_t n_pairs = (n_chunks + 1) / 2;
    const uint64_t n_quads = (n_pairs + 3) / 4;
    const bool aligned = (((uintptr_t)indices) & 15) == 0;
    // (... and the next four are asked for before these are looked up)
    auto load4 = [&](uint64_t quad, u64x2 (&idx)[4]) {
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint64_t i = (quad * 4 + j) * 128 + 2 * lane;
            u64x2 v = {0, 0};
            if (quad < n_quads) {
                if (aligned && i + 1 < n) {
                    v = *(const u64x2 *)(indices + i);
                } else {
                    if (i < n) v.x = (uint64_t)indices[i];
                    if (i + 1 < n) v.y = (uint64_t)indices[i + 1];
                }
            }
            idx[j] = v;
        }
    };
    auto half_scan = [](uint32_t v) {  // inclusive over lanes 0..31 and over lanes 32..63
        v += dpp_or_zero<0x111, 0xF>(v);
        v += dpp_or_zero<0x112, 0xF>(v);
        v += dpp_or_zero<0x114, 0xF>(v);
        v += dpp_or_zero<0x118, 0xF>(v);
        v += dpp_or_zero<0x142, 0xA>(v);
        return v;
    };
    const uint64_t quad_step = (uint64_t)gridDim.x * 4;
    uint64_t quad = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    u64x2 idx_next[4];
    load4(quad, idx_next);
    for (; quad < n_quads; quad += quad_step) {
        u64x2 idx[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) idx[j] = idx_next[j];
        load4(quad + quad_step, idx_next);
        bool bad = false;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint64_t pair = quad * 4 + j;
            const uint64_t i = pair * 128 + 2 * lane;
            uint32_t l0 = 0, l1 = 0;
            if (i < n) {
                if (idx[j].x >= dict_size) bad = true;  // deserialize.rs:162 -> OutOfSpec
                else l0 = l_len[idx[j].x];
            }
            if (i + 1 < n) {
                if (idx[j].y >= dict_size) bad = true;
                else l1 = l_len[idx[j].y];
            }
            const uint32_t lo = half_scan((l0 & 0xFFFFu) + (l1 & 0xFFFFu)), hi = half_scan((l0 >> 16) + (l1 >> 16));
            const uint64_t chunk = 2 * pair + (lane >> 5);
            if ((lane & 31) == 31 && chunk < n_chunks) chunk_sums[chunk] = ((uint64_t)hi << 16) + lo;
        }
        if (bad) atomicOr(error_flag, 1u);
    }
}

// len bytes from src to dst, both of any alignment: single bytes up to a 4-byte boundary of the
// destination, then whole dwords -- each put together from the two aligned source dwords it
// straddles -- then the last bytes.  No source dword past `last_word` (the aligned dword that
// holds the dictionary's last byte) is read.
__device__ __forceinline__ void copy_value(uint8_t *dst, const uint8_t *src, uint64_t len, uintptr_t last_word) {
    while (len && ((uintptr_t)dst & 3)) {
        *dst++ = *src++;
        len--;
    }
    if (len >= 4) {
        const uint32_t sh = (uint32_t)((uintptr_t)src & 3);
        const uint32_t *sw = (const uint32_t *)(src - sh);
        uint32_t lo = *sw;
        while (len >= 4) {
            sw++;
            // the next aligned source dword is needed (and is inside the dictionary) unless the
            // source is itself aligned
            const uint32_t hi = (sh && (uintptr_t)sw <= last_word) ? *sw : 0u;
            const uint32_t v = sh ? __builtin_amdgcn_alignbyte(hi, lo, sh) : lo;
            *(uint32_t *)dst = v;
            lo = sh ? hi : ((uintptr_t)sw <= last_word && len >= 8 ? *sw : 0u);
            dst += 4;
            src += 4;
            len -= 4;
        }
    }
    while (len) {
        *dst++ = *src++;
        len--;
    }
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

// A wavefront takes 64 consecutive values: their bytes form one stretch of the output.  The
// stretch is put together in LDS, laid out at its own 16-byte phase, and then written with 16-byte
// stores, 1 KiB per instruction; the first and last 16 bytes, which neighbouring wavefronts share,
// go byte by byte.  Into the image each lane moves its value a destination dword at a time: the
// dword is cut out of the two aligned dictionary dwords it straddles (v_alignbyte), the bytes
// outside the value are masked off, and what is left is OR-ed into the zeroed image -- so two
// values that share a dword need not know of each other, and no lane ever moves a single byte or
// waits for a load before it may store.  The wavefront makes as many such steps as its longest
// value has dwords.  A stretch longer than the image takes it several times, a value joining
// in each pass with the dwords that fall into it.
// SHORT: no entry is longer than DG_SHORT_LEN bytes, so the stretch fits the image and the steps
// are unrolled; the two cases are two kernels because together they need more registers than eight
// wavefronts per SIMD leave, and the spills land in the common case's loop.
// LDS_DICT: the dictionary itself (32-bit offsets + bytes) is small enough to live in LDS; every
// workgroup copies it there once and then takes chunks of values in turn, so that the per-value
// reads -- two offsets and a few dwords at a random place -- never leave the compute unit.
constexpr uint32_t DG_IMAGE = 2048;        // bytes of a stretch the LDS image takes at a time
constexpr uint32_t DG_IMAGE_LONG = 4096;   // ... in the kernel for dictionaries with long entries
constexpr uint32_t DG_DICT_LDS = 48 << 10;  // dictionary offsets + bytes that may live in LDS
constexpr uint32_t DG_SHORT_LEN = 32;        // entries up to this long: DG_IMAGE / 64
constexpr uint32_t DG_THREADS = 512;
constexpr uint32_t DG_WAVES = DG_THREADS / 64;

template <bool LDS_DICT, bool SHORT>
__global__ __launch_bounds__(DG_THREADS, SHORT ? 8 : 6) void dictionary_gather_kernel(
    const int64_t *__restrict__ dict_offsets, const uint8_t *__restrict__ dict_bytes,
    uint64_t dict_total, uint64_t dict_size, const int64_t *__restrict__ indices, uint64_t n,
    const uint64_t *__restrict__ chunk_prefix, int64_t *__restrict__ offsets_out,
    uint8_t *__restrict__ bytes_out, uint64_t capacity) {
    constexpr uint32_t IMG = SHORT ? DG_IMAGE : DG_IMAGE_LONG;
    __shared__ __attribute__((aligned(16))) uint8_t image_all[DG_WAVES][IMG + 32];
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
    uint32_t *l_off = (uint32_t *)dyn;                                  // [dict_size + 1]
    uint8_t *l_bytes = dyn + (((size_t)dict_size + 1) * 4 + 15) / 16 * 16;  // dictionary bytes at their own 4-byte phase
    const uint32_t phase = (uint32_t)((uintptr_t)dict_bytes & 3);
    // the aligned dwords that each hold a dictionary byte: [0, n_words) of dict_bytes - phase
    const uint32_t *g_words = (const uint32_t *)(dict_bytes - phase);
    const uint64_t n_words = (dict_total + phase + 3) / 4;
    if (LDS_DICT) {
        for (uint64_t k = threadIdx.x; k <= dict_size; k += blockDim.x) l_off[k] = (uint32_t)dict_offsets[k];
        for (uint64_t k = threadIdx.x; k < n_words; k += blockDim.x) ((uint32_t *)l_bytes)[k] = g_words[k];
        __syncthreads();
    }
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t *image = (uint32_t *)image_all[wave];
    const uintptr_t last_word = ((uintptr_t)(dict_bytes + dict_total - 1)) & ~(uintptr_t)3;
    const uint64_t n_chunks = (n + 63) / 64;
    const uint64_t chunk_step = (uint64_t)gridDim.x * DG_WAVES;
    // the next chunk's index and bounds are asked for a chunk ahead: the round trip to memory
    // would otherwise stand in front of everything a wavefront does for a chunk
    uint64_t chunk = (uint64_t)blockIdx.x * DG_WAVES + wave;
    uint64_t idx_next = 0, o0_next = 0, o1_next = 0;
    if (chunk < n_chunks) {
        if (chunk * 64 + lane < n) idx_next = (uint64_t)indices[chunk * 64 + lane];
        o0_next = chunk_prefix[chunk];
        o1_next = chunk_prefix[chunk + 1];
    }
    for (; chunk < n_chunks; chunk += chunk_step) {
        const uint64_t i0 = chunk * 64;
        const uint64_t i = i0 + lane;
        const bool live = i < n;
        const uint64_t idx = idx_next;
        const uint64_t o0 = o0_next, o1 = o1_next;  // wave-uniform: where the chunk's bytes start and end
        if (chunk + chunk_step < n_chunks) {
            const uint64_t nc = chunk + chunk_step;
            idx_next = nc * 64 + lane < n ? (uint64_t)indices[nc * 64 + lane] : 0;
            o0_next = chunk_prefix[nc];
            o1_next = chunk_prefix[nc + 1];
        }
        uint64_t src0 = 0, len = 0;
        if (live) {
            if (idx < dict_size) {  // checked before: always
                if (LDS_DICT) {
                    src0 = l_off[idx];
                    len = (uint64_t)l_off[idx + 1] - src0;
                } else {
                    src0 = (uint64_t)dict_offsets[idx];
                    len = (uint64_t)dict_offsets[idx + 1] - src0;
                }
            }
            if (src0 > dict_total || len > dict_total - src0) len = 0;  // checked before: never taken
        }
        // this value's place among the chunk's bytes
        uint64_t incl = len;
        if (dict_total < (1u << 25)) {  // 64 lengths add up in 32 bits
            incl = wave_inclusive_dpp((uint32_t)len);
        } else {
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t lo = __shfl_up((uint32_t)incl, d), hi = __shfl_up((uint32_t)(incl >> 32), d);
                if (lane >= (uint32_t)d) incl += ((uint64_t)hi << 32) | lo;
            }
        }
        const uint64_t off = o0 + incl - len;
        if (live) {
            offsets_out[i] = (int64_t)off;
            if (i + 1 == n) offsets_out[n] = (int64_t)(off + len);
        }
        // offsets only; or bytes_out holds `capacity` bytes and the column has turned out longer
        // (the host learns that from the total, after this launch)
        if (!bytes_out || o1 > capacity) continue;
        const uint64_t total = o1 - o0;
        // (the scanned sums and the wave scan agree unless the sums were not of these values)
        // (... and the image's addressing is 32-bit: 2 GiB and more go the slow way)
        if (o1 < o0 || total > 0x7FFFFF00ull || readlane_u64(incl, 63) != total || dict_total > 0x7FFFFFF0ull) {
            if (len && o1 >= o0 && off + len <= o1) copy_value(bytes_out + off, dict_bytes + src0, len, last_word);
            continue;
        }
        if (total == 0) continue;
        uint8_t *stretch = bytes_out + o0;
        const uint32_t lead = (uint32_t)((uintptr_t)stretch & 15);
        const uint32_t end = lead + (uint32_t)total;  // the stretch is bytes [lead, end) of the aligned pieces
        uint8_t *aligned = stretch - lead;
        const uint8_t *image8 = (const uint8_t *)image;
        const uint32_t l = (uint32_t)len;
        const uint32_t dst = lead + (uint32_t)(incl - len);      // byte of the value's first, counted like `end`
        const uint32_t dsk = dst & 3;                            // ... and its place in its dword
        // destination dword k of the value holds its bytes 4k - dsk .. 4k - dsk + 3, which are the
        // dictionary's bytes from `from` + 4k on (counted from the first aligned dword; up to 3
        // below zero for a value at the very start: masked off, never loaded)
        const int32_t from = (int32_t)((uint32_t)src0 + phase) - (int32_t)dsk;
        const uint32_t sh = (uint32_t)from & 3;
        const int32_t q0 = from >> 2;
        const uint32_t n_dw = l ? (dsk + l + 3) >> 2 : 0;
        const uint32_t d0 = dst >> 2;  // the value's dwords are d0 .. d0 + n_dw - 1
        auto word = [&](int32_t q) -> uint32_t {
            if (LDS_DICT) return ((const uint32_t *)l_bytes)[q];  // (padded on both sides)
            return (q >= 0 && (uint64_t)q < n_words) ? g_words[q] : 0u;
        };
        auto put = [&](uint32_t k, uint32_t w_lo, uint32_t w_hi, uint32_t wd0) {
            const uint32_t v = __builtin_amdgcn_alignbyte(w_hi, w_lo, sh);
            const int32_t first = (int32_t)(4 * k) - (int32_t)dsk;  // value byte under the dword's byte 0
            uint32_t m = 0xFFFFFFFFu;
            if (first < 0) m <<= 8 * (uint32_t)(-first);            // (k = 0 only: 1..3 bytes of the neighbour before)
            const int32_t over = first + 4 - (int32_t)l;            // at most 3, as k < n_dw
            if (over > 0) m &= 0xFFFFFFFFu >> (8 * (uint32_t)over);
            // (a dword that is all this value's needs no OR: most of a long value's are)
            if (!SHORT && m == 0xFFFFFFFFu) image[d0 + k - wd0] = v;
            else atomicOr(&image[d0 + k - wd0], v & m);
        };
        // 16-byte pieces of the image, which holds bytes [wb, we): piece c is bytes [wb + 16c, wb + 16c + 16)
        auto write_out = [&](uint32_t wb, uint32_t we) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t c = lane; c * 16 < we - wb; c += 64) {
                const uint32_t b0 = wb + c * 16, b1 = b0 + 16;
                if (b0 >= lead && b1 <= end) {
                    *(uint4 *)(aligned + b0) = *(const uint4 *)(image8 + c * 16);
                } else {
                    // a piece shared with a neighbouring wavefront: its bytes [p, q) in at most
                    // seven naturally aligned stores (1, 2, 4 bytes up to an 8-byte boundary, 8, then
                    // 4, 2, 1), not a loop of up to fifteen single bytes
                    uint32_t p = b0 < lead ? lead : b0;
                    const uint32_t q = b1 < end ? b1 : end;
                    const uint8_t *im = image8 - wb;
                    if ((p & 1) && p + 1 <= q) { aligned[p] = im[p]; p += 1; }
                    if ((p & 2) && p + 2 <= q) { *(uint16_t *)(aligned + p) = *(const uint16_t *)(im + p); p += 2; }
                    if ((p & 4) && p + 4 <= q) { *(uint32_t *)(aligned + p) = *(const uint32_t *)(im + p); p += 4; }
                    if (p + 8 <= q) { *(uint64_t *)(aligned + p) = *(const uint64_t *)(im + p); p += 8; }
                    if (p + 4 <= q) { *(uint32_t *)(aligned + p) = *(const uint32_t *)(im + p); p += 4; }
                    if (p + 2 <= q) { *(uint16_t *)(aligned + p) = *(const uint16_t *)(im + p); p += 2; }
                    if (p + 1 <= q) aligned[p] = im[p];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // the image's next writes come after these reads
        };
        auto clear = [&](uint32_t bytes) {
            for (uint32_t c = lane; c * 16 < bytes; c += 64) ((uint4 *)image)[c] = make_uint4(0, 0, 0, 0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        if constexpr (SHORT) {
            // no entry of the dictionary is longer than DG_SHORT_LEN bytes (the launcher has looked):
            // the stretch fits the image (64 values of 16 bytes fill half of it), a value has at
            // most DG_UNROLLED dwords, and every load goes out before the first store waits for one
            constexpr uint32_t DG_UNROLLED = (DG_SHORT_LEN + 3 + 3) / 4;
            if (end > DG_IMAGE + 15 || __any(n_dw > DG_UNROLLED)) continue;  // (never)
            clear(end);
            uint32_t w[DG_UNROLLED + 1];
#pragma unroll
            for (uint32_t k = 0; k <= DG_UNROLLED; k++) w[k] = k <= n_dw && n_dw ? word(q0 + (int32_t)k) : 0u;
#pragma unroll
            for (uint32_t k = 0; k < DG_UNROLLED; k++)
                if (k < n_dw) put(k, w[k], w[k + 1], 0);
            write_out(0, end);
        } else {
            // The image takes the stretch IMG bytes at a time, a value joining in each pass
            // with the dwords that fall into it, four per step.
            for (uint32_t wb = 0; wb < end; wb += IMG) {
                const uint32_t we = end - wb < IMG ? end : wb + IMG;  // this pass: bytes [wb, we)
                clear(we - wb);
                const uint32_t wd0 = wb >> 2, wd1 = (wb + IMG) >> 2;
                const uint32_t k_lo = wd0 > d0 ? wd0 - d0 : 0;
                const uint32_t k_hi = d0 + n_dw > wd1 ? (wd1 > d0 ? wd1 - d0 : 0) : n_dw;
                uint32_t w0 = k_lo < k_hi ? word(q0 + (int32_t)k_lo) : 0u;
                for (uint32_t k = k_lo; __any(k < k_hi); k += 4) {
                    uint32_t w[5];
                    w[0] = w0;
#pragma unroll
                    for (uint32_t j = 1; j <= 4; j++) w[j] = k + j <= k_hi ? word(q0 + (int32_t)(k + j)) : 0u;
#pragma unroll
                    for (uint32_t j = 0; j < 4; j++)
                        if (k + j < k_hi) put(k + j, w[j], w[j + 1], wd0);
                    w0 = w[4];
                }
                write_out(wb, we);
            }
        }
    }
}

// *max_len = max(*max_len, length of the longest entry); entries whose offsets descend count as 0
// (the sums kernels report them).
__global__ __launch_bounds__(256) void dictionary_max_len_kernel(const int64_t *__restrict__ dict_offsets,
                                                                 uint64_t dict_size, uint32_t *__restrict__ max_len) {
    uint64_t m = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < dict_size; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = (uint64_t)dict_offsets[k], e = (uint64_t)dict_offsets[k + 1];
        if (e > b && e - b > m) m = e - b;
    }
    uint32_t v = m > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)m;
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(v, d);
        v = o > v ? o : v;
    }
    if ((threadIdx.x & 63) == 0 && v) atomicMax(max_len, v);
}

hipError_t launch_dictionary_max_len(hipStream_t st, const int64_t *dict_offsets, uint64_t dict_size, uint32_t *max_len) {
    if (dict_size == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((dict_size + 255) / 256, 64);
    dictionary_max_len_kernel<<<blocks, 256, 0, st>>>(dict_offsets, dict_size, max_len);
    return hipGetLastError();
}

hipError_t launch_dictionary_chunk_sums(hipStream_t st, const int64_t *dict_offsets, uint64_t dict_size,
                                        uint64_t dict_total, const int64_t *indices, uint64_t n,
                                        uint64_t *chunk_sums, uint32_t *error_flag) {
    if (n == 0) return hipSuccess;
    const uint64_t chunks = (n + 63) / 64;
    if (dict_size * 4 <= DG_DICT_LDS && dict_total < 0xFFFFFFFFull && chunks >= 256) {
        // (a workgroup's four wavefronts take 32 chunks per turn; 64 workgroups per compute unit
        // measured 185 us per 10^8 values against 198 with 8 -- the table is 4 KB to build)
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((chunks + 31) / 32, 256 * 64);
        dictionary_chunk_sums_lds_kernel<<<blocks, 256, (size_t)dict_size * 4, st>>>(
            dict_offsets, (uint32_t)dict_size, dict_total, indices, n, chunk_sums, error_flag);
    } else {
        dictionary_chunk_sums_kernel<<<(uint32_t)((n + 255) / 256), 256, 0, st>>>(
            dict_offsets, dict_size, dict_total, indices, n, chunk_sums, error_flag);
    }
    return hipGetLastError();
}

hipError_t launch_dictionary_gather(hipStream_t st, const int64_t *dict_offsets,
                                    const uint8_t *dict_bytes, uint64_t dict_size, uint64_t dict_total,
                                    const int64_t *indices, uint64_t n, const uint64_t *chunk_prefix,
                                    int64_t *offsets_out, uint8_t *bytes_out, uint32_t max_len, uint64_t capacity) {
    if (n == 0) return hipSuccess;
    const uint64_t chunks = (n + 63) / 64;
    // workgroups take chunks in turn; 32 of them per compute unit rather than as many as fit at
    // once (4): 1.10 -> 1.05 ms per 10^8 values of 4..28 bytes, 9.8 -> 8.5 ms with values of up to
    // 200 bytes -- the tail of a grid that only just fills the chip is long when chunks differ
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((chunks + DG_WAVES - 1) / DG_WAVES, 256 * 32);
    const uint64_t lds = ((dict_size + 1) * 4 + 15) / 16 * 16 + dict_total + 16;
    const bool in_lds = lds <= DG_DICT_LDS && dict_total < 0xFFFFFFFFull;
    const bool is_short = max_len <= DG_SHORT_LEN;  // max_len: of the dictionary's entries (dictionary_max_len_kernel)
#define ORCD_DG_LAUNCH(L, S)                                                                       \
    dictionary_gather_kernel<L, S><<<blocks, DG_THREADS, (L) ? (size_t)lds : 0, st>>>(               \
        dict_offsets, dict_bytes, dict_total, dict_size, indices, n, chunk_prefix, offsets_out, bytes_out, capacity)
    if (in_lds && is_short) ORCD_DG_LAUNCH(true, true);
    else if (in_lds) ORCD_DG_LAUNCH(true, false);
    else if (is_short) ORCD_DG_LAUNCH(false, true);
    else ORCD_DG_LAUNCH(false, false);
#undef ORCD_DG_LAUNCH
    return hipGetLastError();
}

}  // namespace orcd

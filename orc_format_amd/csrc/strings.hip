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
    for (uint64_t chunk = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); chunk < n_chunks;
         chunk += (uint64_t)gridDim.x * 4) {
        const uint64_t i = chunk * 64 + lane;
        uint64_t l = 0;
        if (i < n) {
            const uint64_t idx = (uint64_t)indices[i];
            if (idx >= dict_size) atomicOr(error_flag, 1u);  // deserialize.rs:162 -> OutOfSpec
            else l = l_len[idx];
        }
        for (int d = 32; d >= 1; d >>= 1) {  // 64 lengths below 2^32 each: the sum cannot wrap
            const uint32_t lo = __shfl_xor((uint32_t)l, d), hi = __shfl_xor((uint32_t)(l >> 32), d);
            l += ((uint64_t)hi << 32) | lo;
        }
        if (lane == 0) chunk_sums[chunk] = l;
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

// A wavefront takes 64 consecutive values: their bytes form one stretch of the output.  Each lane
// copies its value out of the dictionary into an LDS image of that stretch (laid out at the
// stretch's own 16-byte phase), and the wavefront then writes the image with 16-byte stores, 1 KiB
// per instruction; the first and last 16 bytes, which neighbouring wavefronts share, go byte by
// byte.  A stretch longer than the image is copied value by value straight to the output.
// LDS_DICT: the dictionary itself (32-bit offsets + bytes) is small enough to live in LDS; every
// workgroup copies it there once and then takes chunks of values in turn, so that the per-value
// reads -- two offsets and a few dwords at a random place -- never leave the compute unit.
constexpr uint32_t DG_IMAGE = 4096;        // bytes of output 64 values may hold to take the LDS way
constexpr uint32_t DG_DICT_LDS = 48 << 10;  // dictionary offsets + bytes that may live in LDS

template <bool LDS_DICT>
__global__ __launch_bounds__(256) void dictionary_gather_kernel(
    const int64_t *__restrict__ dict_offsets, const uint8_t *__restrict__ dict_bytes,
    uint64_t dict_total, uint64_t dict_size, const int64_t *__restrict__ indices, uint64_t n,
    const uint64_t *__restrict__ chunk_prefix, int64_t *__restrict__ offsets_out,
    uint8_t *__restrict__ bytes_out) {
    __shared__ __attribute__((aligned(16))) uint8_t image_all[4][DG_IMAGE + 32];
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
    uint32_t *l_off = (uint32_t *)dyn;                                  // [dict_size + 1]
    uint8_t *l_bytes = dyn + (((size_t)dict_size + 1) * 4 + 15) / 16 * 16;  // dictionary bytes at their own 4-byte phase
    const uint32_t phase = (uint32_t)((uintptr_t)dict_bytes & 3);
    if (LDS_DICT) {
        for (uint64_t k = threadIdx.x; k <= dict_size; k += blockDim.x) l_off[k] = (uint32_t)dict_offsets[k];
        const uint32_t *src = (const uint32_t *)(dict_bytes - phase);  // aligned dwords that each hold a dictionary byte
        const uint64_t n_words = (dict_total + phase + 3) / 4;
        for (uint64_t k = threadIdx.x; k < n_words; k += blockDim.x) ((uint32_t *)l_bytes)[k] = src[k];
        __syncthreads();
    }
    uint8_t *image = image_all[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63;
    const uintptr_t last_word = LDS_DICT ? ~(uintptr_t)0  // (the LDS copy is padded: a dword past the end may be read)
                                         : ((uintptr_t)(dict_bytes + dict_total - 1)) & ~(uintptr_t)3;
    const uint64_t n_chunks = (n + 63) / 64;
    for (uint64_t chunk = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); chunk < n_chunks;
         chunk += (uint64_t)gridDim.x * 4) {
        const uint64_t i0 = chunk * 64;
        const uint64_t i = i0 + lane;
        const bool live = i < n;
        uint64_t src0 = 0, len = 0;
        if (live) {
            const uint64_t idx = (uint64_t)indices[i];
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
        const uint8_t *src = LDS_DICT ? l_bytes + phase + src0 : dict_bytes + src0;
        // where the chunk's bytes start and end, and this value's place among them
        const uint64_t o0 = chunk_prefix[chunk], o1 = chunk_prefix[chunk + 1];  // wave-uniform
        uint64_t incl = len;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t lo = __shfl_up((uint32_t)incl, d), hi = __shfl_up((uint32_t)(incl >> 32), d);
            if (lane >= (uint32_t)d) incl += ((uint64_t)hi << 32) | lo;
        }
        const uint64_t off = o0 + incl - len;
        if (live) {
            offsets_out[i] = (int64_t)off;
            if (i + 1 == n) offsets_out[n] = (int64_t)(off + len);
        }
        if (!bytes_out) continue;  // offsets only
        const uint64_t total = o1 - o0;
        if (o1 < o0 || total > DG_IMAGE) {
            if (len) copy_value(bytes_out + off, src, len, last_word);
            continue;
        }
        uint8_t *stretch = bytes_out + o0;
        const uint32_t lead = (uint32_t)((uintptr_t)stretch & 15);
        if (len && off >= o0 && off - o0 + len <= total)
            copy_value(image + lead + (uint32_t)(off - o0), src, len, last_word);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 16-byte pieces of the image [lead, lead + total): piece c holds image bytes [16c, 16c + 16)
        const uint32_t end = lead + (uint32_t)total;
        uint8_t *aligned = stretch - lead;
        for (uint32_t c = lane; c * 16 < end; c += 64) {
            const uint32_t b0 = c * 16, b1 = b0 + 16;
            if (b0 >= lead && b1 <= end) {
                *(uint4 *)(aligned + b0) = *(const uint4 *)(image + b0);
            } else {
                for (uint32_t b = b0 < lead ? lead : b0; b < (b1 < end ? b1 : end); b++) aligned[b] = image[b];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the next chunk's image writes come after these reads
    }
}

hipError_t launch_dictionary_chunk_sums(hipStream_t st, const int64_t *dict_offsets, uint64_t dict_size,
                                        uint64_t dict_total, const int64_t *indices, uint64_t n,
                                        uint64_t *chunk_sums, uint32_t *error_flag) {
    if (n == 0) return hipSuccess;
    const uint64_t chunks = (n + 63) / 64;
    if (dict_size * 4 <= DG_DICT_LDS && dict_total < 0xFFFFFFFFull && chunks >= 256) {
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((chunks + 3) / 4, 256 * 8);
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
                                    int64_t *offsets_out, uint8_t *bytes_out) {
    if (n == 0) return hipSuccess;
    const uint64_t chunks = (n + 63) / 64;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((chunks + 3) / 4, 256 * 8);  // workgroups take chunks in turn
    const uint64_t lds = ((dict_size + 1) * 4 + 15) / 16 * 16 + dict_total + 16;
    if (lds <= DG_DICT_LDS && dict_total < 0xFFFFFFFFull) {
        dictionary_gather_kernel<true><<<blocks, 256, (size_t)lds, st>>>(
            dict_offsets, dict_bytes, dict_total, dict_size, indices, n, chunk_prefix, offsets_out, bytes_out);
    } else {
        dictionary_gather_kernel<false><<<blocks, 256, 0, st>>>(
            dict_offsets, dict_bytes, dict_total, dict_size, indices, n, chunk_prefix, offsets_out, bytes_out);
    }
    return hipGetLastError();
}

}  // namespace orcd

// String column support for gfx950: device-wide prefix sum of LENGTH values into offsets,
// UTF-8 validation, dictionary gather.
//
// Values::next(len) of the orc-format crate (src/read/decode/variable_length.rs:15-23) hands out
// consecutive slices of the DATA stream, i.e. value i starts at the exclusive prefix sum of the
// lengths; tests/it/deserialize.rs:106-140 builds one String per value and validates UTF-8.  Here
// the column is produced as offsets[n + 1] plus the contiguous bytes.
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
// UTF-8 validation (std::str::from_utf8 acceptance, tests/it/deserialize.rs:116-118)
//
// Every value is well formed iff the concatenated bytes are well formed and no value starts on a
// continuation byte, so bytes are checked position-parallel and offsets separately.
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ bool is_cont(uint32_t b) { return (b & 0xC0) == 0x80; }

__global__ __launch_bounds__(256) void utf8_bytes_kernel(const uint8_t *__restrict__ bytes,
                                                         uint64_t n, uint32_t *__restrict__ invalid) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = bytes[i];
    bool ok = true;
    if (c < 0x80) {
        ok = true;
    } else if (is_cont(c)) {
        // must be claimed by a lead byte at most 3 positions back
        const uint32_t p1 = i >= 1 ? bytes[i - 1] : 0, p2 = i >= 2 ? bytes[i - 2] : 0,
                       p3 = i >= 3 ? bytes[i - 3] : 0;
        ok = (p1 >= 0xC2 && p1 <= 0xF4) || (is_cont(p1) && p2 >= 0xE0 && p2 <= 0xF4) ||
             (is_cont(p1) && is_cont(p2) && p3 >= 0xF0 && p3 <= 0xF4);
    } else {
        const uint32_t need = c >= 0xF0 ? 3 : (c >= 0xE0 ? 2 : 1);
        if (c < 0xC2 || c > 0xF4 || i + need > n - 1) {
            ok = false;
        } else {
            const uint32_t c1 = bytes[i + 1];
            ok = is_cont(c1);
            if (c == 0xE0) ok = ok && c1 >= 0xA0;
            if (c == 0xED) ok = ok && c1 <= 0x9F;
            if (c == 0xF0) ok = ok && c1 >= 0x90;
            if (c == 0xF4) ok = ok && c1 <= 0x8F;
            if (need >= 2) ok = ok && is_cont(bytes[i + 2]);
            if (need >= 3) ok = ok && is_cont(bytes[i + 3]);
        }
    }
    if (!ok) atomicOr(invalid, 1u);
}

__global__ __launch_bounds__(256) void utf8_offsets_kernel(const uint8_t *__restrict__ bytes,
                                                           const int64_t *__restrict__ offsets,
                                                           uint64_t n, uint64_t total,
                                                           uint32_t *__restrict__ invalid) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t o = (uint64_t)offsets[i], e = (uint64_t)offsets[i + 1];
    if (e > o && o < total && is_cont(bytes[o])) atomicOr(invalid, 1u);
}

hipError_t launch_utf8_validate(hipStream_t st, const uint8_t *bytes, const int64_t *offsets,
                                uint64_t n, uint32_t *invalid_flag, uint64_t total_bytes) {
    if (total_bytes)
        utf8_bytes_kernel<<<(uint32_t)((total_bytes + 255) / 256), 256, 0, st>>>(bytes, total_bytes,
                                                                                invalid_flag);
    if (n)
        utf8_offsets_kernel<<<(uint32_t)((n + 255) / 256), 256, 0, st>>>(bytes, offsets, n,
                                                                        total_bytes, invalid_flag);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// dictionary gather (tests/it/deserialize.rs:142-185)
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void dictionary_lengths_kernel(
    const int64_t *__restrict__ dict_offsets, uint64_t dict_size, uint64_t dict_total,
    const int64_t *__restrict__ indices, uint64_t n, uint64_t *__restrict__ lengths_out,
    uint32_t *__restrict__ error_flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t idx = (uint64_t)indices[i];
    uint64_t l = 0;
    if (idx >= dict_size) {
        atomicOr(error_flag, 1u);  // deserialize.rs:162 -> OutOfSpec
    } else {
        const uint64_t b = (uint64_t)dict_offsets[idx], e = (uint64_t)dict_offsets[idx + 1];
        if (e < b || e > dict_total) atomicOr(error_flag, 2u);  // not offsets of dict_bytes
        else l = e - b;
    }
    lengths_out[i] = l;
}

// one 16-lane group per value copies the entry's bytes
__global__ __launch_bounds__(256) void dictionary_gather_kernel(
    const int64_t *__restrict__ dict_offsets, const uint8_t *__restrict__ dict_bytes,
    const int64_t *__restrict__ indices, uint64_t n, const int64_t *__restrict__ offsets_out,
    uint8_t *__restrict__ bytes_out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = t >> 4;
    const uint32_t sub = (uint32_t)t & 15;
    if (i >= n) return;
    const uint64_t idx = (uint64_t)indices[i];
    const int64_t src = dict_offsets[idx];
    const int64_t l = dict_offsets[idx + 1] - src;
    const int64_t dst = offsets_out[i];
    for (int64_t k = sub; k < l; k += 16) bytes_out[dst + k] = dict_bytes[src + k];
}

hipError_t launch_dictionary_lengths(hipStream_t st, const int64_t *dict_offsets, uint64_t dict_size,
                                     uint64_t dict_total, const int64_t *indices, uint64_t n,
                                     uint64_t *lengths_out, uint32_t *error_flag) {
    if (n == 0) return hipSuccess;
    dictionary_lengths_kernel<<<(uint32_t)((n + 255) / 256), 256, 0, st>>>(
        dict_offsets, dict_size, dict_total, indices, n, lengths_out, error_flag);
    return hipGetLastError();
}

hipError_t launch_dictionary_gather(hipStream_t st, const int64_t *dict_offsets,
                                    const uint8_t *dict_bytes, const int64_t *indices, uint64_t n,
                                    const int64_t *offsets_out, uint8_t *bytes_out) {
    if (n == 0) return hipSuccess;
    const uint64_t threads = n * 16;
    dictionary_gather_kernel<<<(uint32_t)((threads + 255) / 256), 256, 0, st>>>(
        dict_offsets, dict_bytes, indices, n, offsets_out, bytes_out);
    return hipGetLastError();
}

}  // namespace orcd

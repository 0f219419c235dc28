// Ablation lab for the RLE v2 value kernel (development aid, not part of the library).
// Builds a width-17 Direct stream of n_runs runs of 512 values on the device, then times
// variants of the expansion kernel in one process, interleaved, to see where the time goes.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o k3_lab k3_lab.hip && ./k3_lab [n_runs]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr uint32_t RUN_BYTES = 1090;

struct Entry { uint64_t payload, out_off, base, aux; };

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// one thread per output byte of the payload
__global__ void gen_stream(uint8_t *s, Entry *tab, uint32_t n_runs) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = (uint64_t)n_runs * RUN_BYTES;
    if (i >= total) return;
    const uint32_t run = i / RUN_BYTES, b = i % RUN_BYTES;
    if (b == 0) { s[i] = 0x61; tab[run] = Entry{(uint64_t)run * RUN_BYTES + 2, (uint64_t)run * 512, 0, 0}; return; }
    if (b == 1) { s[i] = 0xFF; return; }
    const uint32_t bit = (b - 2) * 8;
    uint32_t v = 0;
    for (int k = 0; k < 8; k++) {
        const uint32_t pos = bit + k, idx = pos / 17, off = pos % 17;
        const uint64_t u = (mix((uint64_t)run * 512 + idx + 1) & 0xFFFF) | 0x10000;
        v = (v << 1) | ((u >> (16 - off)) & 1);
    }
    s[i] = (uint8_t)v;
}

__global__ void check_out(const int64_t *out, uint64_t n, unsigned long long *bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t u = (mix(i + 1) & 0xFFFF) | 0x10000;
    const int64_t v = (int64_t)((u >> 1) ^ (0 - (u & 1)));
    if (out[i] != v) atomicAdd(bad, 1ull);
}

__device__ __forceinline__ uint32_t lds_bits32(const uint32_t *lds, uint32_t bit, uint32_t w) {
    const uint32_t d = bit >> 5, sh = bit & 31;
    const uint64_t x = ((uint64_t)lds[d] << 32) | lds[d + 1];
    return (uint32_t)((x << sh) >> (64 - w));
}

// MODE bits: 1 = read table entry, 2 = stage stream bytes into LDS, 4 = decode from LDS (else
// synthesise values), 8 = store.  PERSIST: grid-stride loop over runs instead of one run per wave.
template <int MODE, bool PERSIST, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void lab_kernel(const uint8_t *__restrict__ s,
                                                        const Entry *__restrict__ tab,
                                                        uint32_t n_runs, int64_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[WAVES][1088 / 4 + 8 + 4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *lds = stage[wave];
    const uint32_t stride = gridDim.x * WAVES;
    for (uint32_t g = blockIdx.x * WAVES + wave; g < n_runs; g += stride) {
        uint64_t payload = (uint64_t)g * RUN_BYTES + 2, out_off = (uint64_t)g * 512;
        if (MODE & 1) {
            const Entry e = tab[g];
            payload = __builtin_amdgcn_readfirstlane((uint32_t)e.payload) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(e.payload >> 32)) << 32);
            out_off = __builtin_amdgcn_readfirstlane((uint32_t)e.out_off) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(e.out_off >> 32)) << 32);
        }
        uint32_t bit0 = 0;
        if (MODE & 2) {
            const uint32_t skew = (uint32_t)((uintptr_t)(s + payload) & 15);
            const uint4 *src = (const uint4 *)(s + payload - skew);
            const uint32_t n16 = (1088 + skew + 15) >> 4;
            for (uint32_t c = lane; c < n16; c += 64) {
                uint4 v = src[c];
                v.x = __builtin_bswap32(v.x); v.y = __builtin_bswap32(v.y);
                v.z = __builtin_bswap32(v.z); v.w = __builtin_bswap32(v.w);
                ((uint4 *)lds)[c] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            bit0 = skew * 8;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t i0 = 128 * j + 2 * lane;
            uint64_t v0, v1;
            if (MODE & 4) {
                const uint32_t m0 = lds_bits32(lds, bit0 + i0 * 17, 17), m1 = lds_bits32(lds, bit0 + (i0 + 1) * 17, 17);
                v0 = (uint64_t)(int64_t)(int32_t)((m0 >> 1) ^ (0u - (m0 & 1u)));
                v1 = (uint64_t)(int64_t)(int32_t)((m1 >> 1) ^ (0u - (m1 & 1u)));
            } else {
                v0 = out_off + i0 + bit0;
                v1 = v0 + 1;
            }
            if (MODE & 8) *(ulonglong2 *)(out + out_off + i0) = make_ulonglong2(v0, v1);
            else if (v0 == 0x123456789abcdefull) out[0] = (int64_t)v1;  // keep the values live
        }
        if (!PERSIST) break;
    }
}

// Software-pipelined persistent variant: the next run's table entry and bytes are loaded into
// registers while the current run is decoded from LDS.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void lab_pipelined(const uint8_t *__restrict__ s,
                                                           const Entry *__restrict__ tab,
                                                           uint32_t n_runs, int64_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[WAVES][1088 / 4 + 8 + 4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *lds = stage[wave];
    const uint32_t stride = gridDim.x * WAVES;
    uint32_t g = blockIdx.x * WAVES + wave;
    if (g >= n_runs) return;
    // prologue: entry and bytes of the first run, entry of the second
    Entry e_cur = tab[g];
    Entry e_nxt = tab[g + stride < n_runs ? g + stride : g];
    uint4 r0, r1;
    {
        const uint32_t skew = (uint32_t)((uintptr_t)(s + e_cur.payload) & 15);
        const uint4 *src = (const uint4 *)(s + e_cur.payload - skew);
        r0 = src[lane];
        r1 = lane < 6 ? src[64 + lane] : make_uint4(0, 0, 0, 0);
    }
    for (; g < n_runs; g += stride) {
        const uint64_t payload = e_cur.payload, out_off = e_cur.out_off;
        const uint32_t skew = (uint32_t)((uintptr_t)(s + payload) & 15);
        // bytes of the current run: registers -> LDS
        {
            uint4 v = r0;
            v.x = __builtin_bswap32(v.x); v.y = __builtin_bswap32(v.y);
            v.z = __builtin_bswap32(v.z); v.w = __builtin_bswap32(v.w);
            ((uint4 *)lds)[lane] = v;
            if (lane < 6) {
                uint4 u = r1;
                u.x = __builtin_bswap32(u.x); u.y = __builtin_bswap32(u.y);
                u.z = __builtin_bswap32(u.z); u.w = __builtin_bswap32(u.w);
                ((uint4 *)lds)[64 + lane] = u;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // issue the loads of the next run (bytes) and of the run after it (entry)
        const uint32_t gn = g + stride, gnn = g + 2 * stride;
        e_cur = e_nxt;
        if (gn < n_runs) {
            const uint32_t sk2 = (uint32_t)((uintptr_t)(s + e_cur.payload) & 15);
            const uint4 *src = (const uint4 *)(s + e_cur.payload - sk2);
            r0 = src[lane];
            if (lane < 6) r1 = src[64 + lane];
        }
        if (gnn < n_runs) e_nxt = tab[gnn];
        const uint32_t bit0 = skew * 8;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t i0 = 128 * j + 2 * lane;
            const uint32_t m0 = lds_bits32(lds, bit0 + i0 * 17, 17), m1 = lds_bits32(lds, bit0 + (i0 + 1) * 17, 17);
            const uint64_t v0 = (uint64_t)(int64_t)(int32_t)((m0 >> 1) ^ (0u - (m0 & 1u)));
            const uint64_t v1 = (uint64_t)(int64_t)(int32_t)((m1 >> 1) ^ (0u - (m1 & 1u)));
            *(ulonglong2 *)(out + out_off + i0) = make_ulonglong2(v0, v1);
        }
    }
}


// WPW windows per wavefront: all loads first, then decode and store each window.
// NT: nontemporal stores.  XCD: remap workgroups so that each XCD (workgroup id mod 8) walks one
// contiguous eighth of the column.
template <int WAVES, int WPW, bool NT, bool XCD>
__global__ __launch_bounds__(WAVES * 64) void lab_multi(const uint8_t *__restrict__ s,
                                                       const Entry *__restrict__ tab,
                                                       uint32_t n_runs, int64_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[WAVES][WPW][1088 / 4 + 8 + 4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t b = blockIdx.x;
    if (XCD) {
        const uint32_t nb = gridDim.x, q = nb / 8, r = nb % 8, x = b % 8, i = b / 8;
        b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const uint32_t g0 = (b * WAVES + wave) * WPW;
    uint32_t skews[WPW];
    uint64_t outs[WPW];
#pragma unroll
    for (int k = 0; k < WPW; k++) {
        const uint32_t g = g0 + k;
        skews[k] = 0;
        outs[k] = 0;
        if (g >= n_runs) continue;
        const Entry e = tab[g];
        const uint64_t payload = __builtin_amdgcn_readfirstlane((uint32_t)e.payload) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(e.payload >> 32)) << 32);
        outs[k] = __builtin_amdgcn_readfirstlane((uint32_t)e.out_off) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(e.out_off >> 32)) << 32);
        const uint32_t skew = (uint32_t)((uintptr_t)(s + payload) & 15);
        skews[k] = skew;
        const uint4 *src = (const uint4 *)(s + payload - skew);
        const uint32_t n16 = (1088 + skew + 15) >> 4;
        for (uint32_t c = lane; c < n16; c += 64) {
            uint4 v = src[c];
            v.x = __builtin_bswap32(v.x); v.y = __builtin_bswap32(v.y);
            v.z = __builtin_bswap32(v.z); v.w = __builtin_bswap32(v.w);
            ((uint4 *)stage[wave][k])[c] = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < WPW; k++) {
        if (g0 + k >= n_runs) continue;
        const uint32_t *lds = stage[wave][k];
        const uint32_t bit0 = skews[k] * 8;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t i0 = 128 * j + 2 * lane;
            const uint32_t m0 = lds_bits32(lds, bit0 + i0 * 17, 17), m1 = lds_bits32(lds, bit0 + (i0 + 1) * 17, 17);
            const uint64_t v0 = (uint64_t)(int64_t)(int32_t)((m0 >> 1) ^ (0u - (m0 & 1u)));
            const uint64_t v1 = (uint64_t)(int64_t)(int32_t)((m1 >> 1) ^ (0u - (m1 & 1u)));
            int64_t *dst = out + outs[k] + i0;
            if (NT) {
                __builtin_nontemporal_store((int64_t)v0, dst);
                __builtin_nontemporal_store((int64_t)v1, dst + 1);
            } else {
                *(ulonglong2 *)dst = make_ulonglong2(v0, v1);
            }
        }
    }
}

template <typename F>
float time_ms(F launch, int reps = 5) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < reps; i++) {
        CHECK(hipEventRecord(a));
        launch();
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char **argv) {
    const uint32_t n_runs = argc > 1 ? atoi(argv[1]) : 1953125;
    const uint64_t n = (uint64_t)n_runs * 512, sbytes = (uint64_t)n_runs * RUN_BYTES;
    uint8_t *s;
    Entry *tab;
    int64_t *out;
    unsigned long long *bad;
    CHECK(hipMalloc(&s, sbytes + 64));
    CHECK(hipMalloc(&tab, (size_t)(n_runs + 1) * sizeof(Entry)));
    CHECK(hipMalloc(&out, n * 8));
    CHECK(hipMalloc(&bad, 8));
    gen_stream<<<(uint32_t)((sbytes + 255) / 256), 256>>>(s, tab, n_runs);
    CHECK(hipDeviceSynchronize());
    const double gb = (n * 8.0 + sbytes) / 1e9;
    auto report = [&](const char *name, float ms, bool verify) {
        unsigned long long h = 0;
        if (verify) {
            CHECK(hipMemset(bad, 0, 8));
            check_out<<<(uint32_t)((n + 255) / 256), 256>>>(out, n, bad);
            CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        }
        printf("%-44s %8.3f ms  %7.0f GB/s (in+out)%s\n", name, ms, gb / ms * 1e3,
               verify ? (h ? "  MISMATCH" : "  verified") : "");
        fflush(stdout);
    };
    const uint32_t blocks4 = (n_runs + 3) / 4;
#define RUN1(MODE, NAME, VERIFY) report(NAME, time_ms([&] { lab_kernel<MODE, false, 4><<<blocks4, 256>>>(s, tab, n_runs, out); }), VERIFY)
    RUN1(8, "store only (synthetic values)", false);
    RUN1(9, "entry + store", false);
    RUN1(11, "entry + stage + store", false);
    RUN1(15, "entry + stage + decode + store (full)", true);
    RUN1(7, "entry + stage + decode, no store", false);
    RUN1(14, "no entry: stage + decode + store", true);
#define RUNM(WAVES, WPW, NT, XCD, NAME) report(NAME, time_ms([&] { lab_multi<WAVES, WPW, NT, XCD><<<(n_runs + WAVES * WPW - 1) / (WAVES * WPW), WAVES * 64>>>(s, tab, n_runs, out); }), true)
    RUNM(4, 1, false, false, "multi: 4 waves x 1 window");
    RUNM(4, 1, true, false, "multi: 4 waves x 1 window, nontemporal");
    RUNM(4, 1, false, true, "multi: 4 waves x 1 window, XCD remap");
    RUNM(4, 2, false, false, "multi: 4 waves x 2 windows");
    RUNM(4, 2, false, true, "multi: 4 waves x 2 windows, XCD remap");
    RUNM(4, 4, false, false, "multi: 4 waves x 4 windows");
    RUNM(8, 1, false, false, "multi: 8 waves x 1 window");
    RUNM(8, 2, false, false, "multi: 8 waves x 2 windows");
    RUNM(2, 1, false, false, "multi: 2 waves x 1 window");
    RUNM(1, 1, false, false, "multi: 1 wave x 1 window");
    RUNM(16, 1, false, false, "multi: 16 waves x 1 window");
    for (int bpc : {8}) {
        char name[96];
        const uint32_t pb = 256 * bpc;
        snprintf(name, sizeof name, "persistent full, %d blocks/CU x4 waves", bpc);
        report(name, time_ms([&] { lab_kernel<15, true, 4><<<pb, 256>>>(s, tab, n_runs, out); }), true);
        snprintf(name, sizeof name, "persistent pipelined, %d blocks/CU x4 waves", bpc);
        report(name, time_ms([&] { lab_pipelined<4><<<pb, 256>>>(s, tab, n_runs, out); }), true);
    }
    for (int bpc : {2, 4}) {
        char name[96];
        snprintf(name, sizeof name, "persistent pipelined, %d blocks/CU x8 waves", bpc);
        report(name, time_ms([&] { lab_pipelined<8><<<256 * bpc, 512>>>(s, tab, n_runs, out); }), true);
    }
    return 0;
}

// Prefix sums over the segments of a run-length stream (device code shared by the RLE v2 and
// byte-RLE header scans).
//
// A segment is the stretch of headers between two entry points, walked by one lane.  The counting
// kernels run 256 lanes per workgroup and finish with a workgroup-local scan (seg_publish); one
// small kernel then scans the per-workgroup totals; the emitting kernels add the two levels back
// together (seg_global_offset).
#pragma once

#include "orcd_internal.h"

namespace orcd {

constexpr int SEG_BLOCK = 256;  // lanes (= segments) per workgroup of the count and emit kernels

struct BlockTotals {
    uint64_t n_vals;
    uint32_t n_runs;
    uint32_t last_len;        // length of the last run in the workgroup's segments (0: no run)
    uint64_t bad_local_end;   // values of the workgroup up to the end of its first stopped walk
    uint32_t bad_seg;         // that segment, or 0xFFFFFFFF
    uint32_t bad_status;
    uint32_t max_seg_runs;    // most runs any one of the workgroup's segments holds
    uint32_t pad;
};

struct BlockOffset {
    uint64_t val_off;
    uint32_t run_off;
    uint32_t prev_len;
};

struct SegTable {
    SegCount *seg_count;    // [n_seg]
    SegOffset *seg_local;   // [n_seg]   workgroup-local exclusive offsets
    BlockTotals *blk;       // [n_blocks]
    BlockOffset *boff;      // [n_blocks]
    ScanTotals *totals;
    // (nullable) the same totals once more, in host memory the device can write, stamped with
    // `host_stamp` in ScanTotals::pad after everything else: the host watches for the stamp
    // instead of queueing a copy and waiting for the stream
    ScanTotals *host_totals;
    uint32_t host_stamp;
};

__device__ __forceinline__ uint64_t seg_shfl_up64(uint64_t v, unsigned d) {
    const uint32_t lo = __shfl_up((uint32_t)v, d), hi = __shfl_up((uint32_t)(v >> 32), d);
    return ((uint64_t)hi << 32) | lo;
}

// Called by every lane of a SEG_BLOCK-wide counting workgroup, valid or not (`seg < n_seg`), with
// the result of its walk.
__device__ __forceinline__ void seg_publish(const SegTable &t, uint32_t seg, uint32_t n_seg,
                                            uint32_t n_runs, uint32_t last_len, uint32_t status,
                                            uint64_t n_vals) {
    __shared__ uint64_t w_vals[SEG_BLOCK / 64];
    __shared__ uint64_t w_key[SEG_BLOCK / 64];
    __shared__ uint32_t w_runs[SEG_BLOCK / 64];
    __shared__ uint32_t w_max[SEG_BLOCK / 64];
    __shared__ uint32_t first_bad;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool valid = seg < n_seg;
    if (!valid) { n_runs = 0; last_len = 0; status = SEG_OK; n_vals = 0; }
    if (threadIdx.x == 0) first_bad = 0xFFFFFFFFu;
    uint32_t most = n_runs;
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(most, d);
        most = most > o ? most : o;
    }

    // inclusive scans inside the wavefront: values, runs, and "most recent non-empty segment"
    // as a running maximum over (lane + 1, last_len)
    uint64_t iv = n_vals;
    uint32_t ir = n_runs;
    uint64_t ik = n_runs ? (((uint64_t)threadIdx.x + 1) << 10) | last_len : 0;
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t uv = seg_shfl_up64(iv, d);
        const uint32_t ur = __shfl_up(ir, d);
        const uint64_t uk = seg_shfl_up64(ik, d);
        if (lane >= d) {
            iv += uv;
            ir += ur;
            ik = ik > uk ? ik : uk;
        }
    }
    if (lane == 63) { w_vals[wave] = iv; w_runs[wave] = ir; w_key[wave] = ik; w_max[wave] = most; }
    __syncthreads();
    uint64_t pv = 0, pk = 0, tv = 0, tk = 0;
    uint32_t pr = 0, tr = 0;
    for (int w = 0; w < SEG_BLOCK / 64; w++) {
        if (w < wave) { pv += w_vals[w]; pr += w_runs[w]; pk = pk > w_key[w] ? pk : w_key[w]; }
        tv += w_vals[w];
        tr += w_runs[w];
        tk = tk > w_key[w] ? tk : w_key[w];
    }
    const uint64_t incl_key = ik > pk ? ik : pk;
    // exclusive value of the key scan: the inclusive value of the previous lane
    uint64_t excl_key = seg_shfl_up64(incl_key, 1);
    if (lane == 0) excl_key = pk;
    const uint64_t val_off = pv + iv - n_vals;
    if (valid) {
        SegCount c;
        c.n_runs = n_runs;
        c.last_len = last_len;
        c.status = status;
        c.pad = 0;
        c.n_vals = n_vals;
        t.seg_count[seg] = c;
        SegOffset o;
        o.val_off = val_off;
        o.run_off = pr + ir - n_runs;
        o.prev_len = (uint32_t)(excl_key & 1023);
        t.seg_local[seg] = o;
        if (status != SEG_OK) atomicMin(&first_bad, threadIdx.x);
    }
    __syncthreads();
    const uint32_t fb = first_bad;
    if (fb != 0xFFFFFFFFu && threadIdx.x == fb) {
        t.blk[blockIdx.x].bad_local_end = val_off + n_vals;
        t.blk[blockIdx.x].bad_seg = seg;
        t.blk[blockIdx.x].bad_status = status;
    }
    if (threadIdx.x == 0) {
        t.blk[blockIdx.x].n_vals = tv;
        t.blk[blockIdx.x].n_runs = tr;
        t.blk[blockIdx.x].last_len = (uint32_t)(tk & 1023);
        uint32_t m = 0;
        for (int w = 0; w < SEG_BLOCK / 64; w++) m = m > w_max[w] ? m : w_max[w];
        t.blk[blockIdx.x].max_seg_runs = m;
        t.blk[blockIdx.x].pad = 0;
        if (fb == 0xFFFFFFFFu) {
            t.blk[blockIdx.x].bad_local_end = 0;
            t.blk[blockIdx.x].bad_seg = 0xFFFFFFFFu;
            t.blk[blockIdx.x].bad_status = SEG_OK;
        }
    }
}

// Global exclusive offsets of a segment, for the emitting kernels (same SEG_BLOCK geometry).
__device__ __forceinline__ SegOffset seg_global_offset(const SegTable &t, uint32_t seg) {
    const SegOffset l = t.seg_local[seg];
    const BlockOffset b = t.boff[seg / SEG_BLOCK];
    SegOffset o;
    o.val_off = b.val_off + l.val_off;
    o.run_off = b.run_off + l.run_off;
    o.prev_len = l.prev_len ? l.prev_len : b.prev_len;
    return o;
}

// The counting pass keeps one 64-bit word per run for the first `cap` runs of a segment in
// staged[seg * cap + k] (cap: a multiple of SB_HELD chosen per call), so that the run table can be written later by one lane per run
// instead of a second walk of the header chain.  The words leave in pieces of SB_HELD, collected
// in a lane-private LDS stretch: all lanes of the grid write at once, each to a place of its own,
// and 8-byte pieces of that many cache lines would each go out to memory as a partial line.
constexpr uint32_t SB_HELD = 8;              // words held per lane
constexpr uint32_t SB_STRIDE = SB_HELD + 1;  // LDS pitch in words (odd: spreads the banks)
typedef unsigned long long seg_u64x2 __attribute__((ext_vector_type(2)));

struct StagedWriter {
    uint64_t *mine;  // this segment's `cap` slots, or null: nothing is staged
    uint64_t *held;  // this lane's SB_STRIDE words of LDS
    uint32_t cap;
    // word of the segment's k-th run; k ascends by one from 0
    __device__ __forceinline__ void put(uint32_t k, uint64_t word) {
        if (!mine || k >= cap) return;
        held[k & (SB_HELD - 1)] = word;
        if ((k & (SB_HELD - 1)) != SB_HELD - 1) return;
        seg_u64x2 *dst = (seg_u64x2 *)(mine + (k - (SB_HELD - 1)));
#pragma unroll
        for (uint32_t i = 0; i < SB_HELD / 2; i++) {
            seg_u64x2 v;
            v.x = held[2 * i];
            v.y = held[2 * i + 1];
            dst[i] = v;
        }
    }
    // after the last put: n runs were seen in all
    __device__ __forceinline__ void finish(uint32_t n) {
        if (!mine || n >= cap) return;  // a full table left with its last piece
        for (uint32_t k = n & ~(SB_HELD - 1); k < n; k++) mine[k] = held[k & (SB_HELD - 1)];
    }
};

hipError_t launch_block_totals_scan(hipStream_t st, const SegTable &t, uint32_t n_blocks,
                                    uint64_t num_values);

}  // namespace orcd

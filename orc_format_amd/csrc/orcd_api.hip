// C ABI of the decoder: context (stream, events, recycled device buffers) and the decode entry
// points, which validate arguments, move host buffers when asked to, and chain the kernels.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <sched.h>
#include <pthread.h>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <thread>
#include <vector>

#include "orcd_internal.h"
#include "seg_scan.h"

using namespace orcd;

// No exception leaves the C ABI: host allocations (std::vector, new) and thread or lock creation
// are the only things here that throw.
#define ORCD_CATCH                                     \
    catch (...) {                                      \
        return ORCD_NO_MEMORY;                         \
    }

namespace {

enum Slot {
    SLOT_IN_A = 0,     // host stream staged on the device
    SLOT_IN_B,         // second host input (string DATA, dictionary bytes)
    SLOT_IN_ENTRIES,   // host entry points staged on the device
    SLOT_IN_C,         // third host input (dictionary offsets / indices)
    SLOT_IN_D,
    SLOT_SEG_COUNT,
    SLOT_SEG_OFF,
    SLOT_BLK_TOTALS,
    SLOT_BLK_OFF,
    SLOT_TOTALS,
    SLOT_RUNS,
    SLOT_GROUPS,
    SLOT_GROUP_RUNS,
    SLOT_OUT_A,        // device staging of a host output
    SLOT_OUT_B,
    SLOT_PACKED,       // byte-RLE output feeding the boolean expansion
    SLOT_SCAN_SUMS,
    SLOT_FLAGS,
    SLOT_TMP,
    SLOT_STAGED_RUNS,
    SLOT_SPEC_FLAG,    // the word a speculative launch raises when its guess about the stream is wrong
    SLOT_COMPOSE,      // exit tables of a byte-RLE stream without entry points
    N_SLOTS
};

constexpr int N_READERS = 8;  // threads that read a stripe's bytes, each through a pinned chunk of its own
constexpr int N_EVENTS = 5;  // start, after header count + scan, after emit, after values / offsets, after string bytes

}  // namespace

// Blocks of device or pinned host memory that a context keeps when they are released and hands
// out again (see dev_alloc).
struct BlockCache {
    const bool host;
    std::mutex lock;
    std::multimap<size_t, void *> free_blocks;     // capacity -> block
    std::unordered_map<void *, size_t> live;       // block in use -> capacity
    size_t held = 0;                               // bytes in free_blocks
    explicit BlockCache(bool h) : host(h) {}
    hipError_t raw_alloc(void **p, size_t n) const { return host ? hipHostMalloc(p, n, hipHostMallocDefault) : hipMalloc(p, n); }
    void raw_free(void *p) const {
        if (host) (void)hipHostFree(p);
        else (void)hipFree(p);
    }
    hipError_t take(size_t bytes, void **p) {
        *p = nullptr;
        const size_t want = (std::max<size_t>(bytes, 16) + 255) & ~(size_t)255;
        std::lock_guard<std::mutex> g(lock);
        auto it = free_blocks.lower_bound(want);
        if (it != free_blocks.end() && it->first <= want + want / 2 + 4096) {  // a block that fits without much waste
            *p = it->second;
            held -= it->first;
            live[*p] = it->first;
            free_blocks.erase(it);
            return hipSuccess;
        }
        const hipError_t e = raw_alloc(p, want);
        if (e == hipSuccess) live[*p] = want;
        return e;
    }
    // false: not a block of this cache
    bool give_back(void *p, size_t max_held) {
        std::lock_guard<std::mutex> g(lock);
        auto it = live.find(p);
        if (it == live.end()) return false;
        const size_t cap = it->second;
        live.erase(it);
        if (held + cap > max_held) {
            raw_free(p);
        } else {
            free_blocks.emplace(cap, p);
            held += cap;
        }
        return true;
    }
    void release_all() {
        std::lock_guard<std::mutex> g(lock);
        for (auto &kv : free_blocks) raw_free(kv.second);
        free_blocks.clear();
        held = 0;
    }
};

// What one party needs to move large transfers: pinned chunks and an event per chunk (the threads
// are the process's helpers, orcd_internal_parallel_run).
// The calling thread of a context has one set; the background reader of orcd_stripe_prefetch has
// another, with a stream of its own.
struct StageSet {
    uint8_t *h_stage[N_READERS] = {};  // pinned chunks, allocated on first use
    hipEvent_t ev_stage[N_READERS] = {};
    bool stage_busy[N_READERS] = {};   // a copy out of the chunk was queued and not yet waited for
    hipStream_t stream = nullptr;      // where its copies are queued (null: the context's stream)
    void release() {
        for (int i = 0; i < N_READERS; i++) {
            if (h_stage[i]) (void)hipHostFree(h_stage[i]);
            if (ev_stage[i]) (void)hipEventDestroy(ev_stage[i]);
            h_stage[i] = nullptr;
            ev_stage[i] = nullptr;
        }
    }
};

// One stripe's bytes on their way to the device ahead of the decode call that wants them.
struct PrefetchJob {
    orcd_file *file = nullptr;
    size_t stripe = 0;
    uint64_t file_off = 0, len = 0;  // the region of the file
    void *d_buf = nullptr;           // a block of the context's device cache
    // a compressed file: the bytes stay in host memory as well (the decode call walks the chunk
    // headers there)
    bool keep_host = false;
    void *host = nullptr;  // pinned, from the prefetcher's pool
    size_t host_cap = 0;
    int status = ORCD_OK;
    bool done = false;
};

// The background reader of a context: one thread that takes the queued regions in order and moves
// each through its own stage set (so up to N_READERS threads read and copy side by side while
// the context's own thread decodes the stripe before).
struct Prefetcher {
    static constexpr size_t MAX_JOBS = 2;
    std::thread worker;
    std::mutex m;
    std::condition_variable wake, finished;
    std::deque<PrefetchJob *> jobs;  // queued, being read, or read and not yet claimed -- oldest first
    bool stop = false;
    StageSet stage;
    // pinned host blocks for the stripes of compressed files (read into by several threads, copied
    // to the device in one piece, kept for the decode call): handed out and taken back under `m`
    std::vector<std::pair<void *, size_t>> pinned;
    static constexpr size_t MAX_PINNED = 3;
};

// ---- device memory of orcd_rlev2_decode_noalloc ------------------------------------------------
// That entry point may not allocate, so everything its kernels keep beside the caller's input and
// output -- segment counts and offsets, workgroup totals, the run table, the per-window tables, the
// flag of the speculative launch -- lives in this array, which every GPU the library runs on holds
// from the moment the code object is loaded.  Its name is a host-side stand-in, not a device
// address: the address on a device comes from hipGetSymbolAddress with that device current
// (noalloc_scratch), and only that pointer is handed to kernels, memsets and copies.
namespace {
constexpr uint32_t NOALLOC_MAX_SEGMENTS = ORCD_NOALLOC_MAX_ENTRIES;
constexpr uint64_t NOALLOC_MAX_VALUES = ORCD_NOALLOC_MAX_VALUES;   // values asked for, and values the stream holds
constexpr uint64_t NOALLOC_MAX_RUNS = ORCD_NOALLOC_MAX_RUNS;       // runs the stream holds
constexpr uint64_t NOALLOC_MAX_WINDOWS = (NOALLOC_MAX_VALUES >> WINDOW_SHIFT) + 2;
constexpr size_t up256(size_t n) { return (n + 255) & ~(size_t)255; }
constexpr size_t NA_SEG_COUNT = 0;
constexpr size_t NA_SEG_OFF = NA_SEG_COUNT + up256(NOALLOC_MAX_SEGMENTS * sizeof(SegCount));
constexpr size_t NA_BLK = NA_SEG_OFF + up256(NOALLOC_MAX_SEGMENTS * sizeof(SegOffset));
constexpr size_t NA_BOFF = NA_BLK + up256(NOALLOC_MAX_SEGMENTS / SEG_BLOCK * sizeof(BlockTotals));
constexpr size_t NA_TOTALS = NA_BOFF + up256(NOALLOC_MAX_SEGMENTS / SEG_BLOCK * sizeof(BlockOffset));
constexpr size_t NA_FLAG = NA_TOTALS + up256(sizeof(ScanTotals));
constexpr size_t NA_RUNS = NA_FLAG + 256;
constexpr size_t NA_GROUPS = NA_RUNS + up256((NOALLOC_MAX_RUNS + 1) * sizeof(uint64_t));
constexpr size_t NA_GROUP_RUNS = NA_GROUPS + up256(NOALLOC_MAX_WINDOWS * sizeof(uint32_t));
constexpr size_t NA_BYTES = NA_GROUP_RUNS + up256(NOALLOC_MAX_WINDOWS * sizeof(RunEntry));
static_assert(NA_BYTES < (8u << 20), "the static scratch array is meant to stay modest");
constexpr int NOALLOC_MAX_DEVICES = 64;
}  // namespace

static __device__ __align__(256) unsigned char scratch[NA_BYTES];

namespace {
std::mutex g_scratch_lock;
unsigned char *g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};  // per device: each GPU holds its own copy

// Address of `scratch` on the current device (which must be `device`); looked up once per device.
hipError_t noalloc_scratch(int device, unsigned char **ptr) {
    *ptr = nullptr;
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(g_scratch_lock);
    if (!g_scratch_ptr[device]) {
        unsigned char *p = nullptr;
        const hipError_t e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = p;
    }
    *ptr = g_scratch_ptr[device];
    return hipSuccess;
}
}  // namespace

struct orcd_ctx {
    int device = 0;
    unsigned char *scratch_dev = nullptr;  // this device's copy of `scratch` (see noalloc_scratch)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[N_EVENTS] = {};
    bool ev_valid[N_EVENTS] = {};
    void *slot[N_SLOTS] = {};
    size_t cap[N_SLOTS] = {};
    ScanTotals *h_totals = nullptr;  // pinned, coherent: the scan kernel writes it
    uint32_t totals_stamp = 0;
    uint64_t *h_scalars = nullptr;   // pinned, 8 words
    StageSet stage;                    // large host transfers of the calling thread, queued on `stream`
    // where the host's threads leave the chunks they inflate: pinned when that can be had (one
    // copy to the device then), kept from call to call
    void *inflate_pinned = nullptr;
    size_t inflate_pinned_cap = 0;
    std::vector<uint8_t> inflate_scratch;
    Prefetcher *prefetcher = nullptr;  // background reader of stripes to come (orcd_stripe_prefetch)
    // Result arrays and per-call temporaries of the column calls: released blocks are kept and
    // handed out again (see dev_alloc).
    BlockCache dev_cache{false};   // device memory
    BlockCache host_cache{true};   // pinned host memory: host results of the column calls
    // beyond these, released blocks go back to the runtime (orcd_ctx_set_cache_limit)
    size_t dev_cache_limit = 16ull << 30, host_cache_limit = 8ull << 30;  // (orcd_ctx_set_cache_limit)
    hipError_t last_hip = hipSuccess;
    // Speculative decode of regular columns (try_strided): which way the last RLE v2 call went
    // (ORCD_PATH_*), and how many eligible calls skip the attempt after it failed twice in a row.
    cpu_set_t numa_cpus;           // CPUs of the NUMA node the device hangs off, for the helper threads
    bool have_numa_cpus = false;
    int last_path = 0;
    const char *last_kernel = "";  // the kernel that expanded the values of the last RLE v2 call
    float stripe_feed_ms = -1.0f, stripe_decode_ms = -1.0f;  // the last orcd_decode_stripe call (host clock)
    int spec_pref = ORCD_PATH_STRIDED;  // the speculative path that worked last, tried first
    int selfplan_rest = 0;              // calls for which the plan is a pass of its own without asking (see attempt_planned)
    uint32_t spec_fail_streak = 0, spec_skip = 0;
    bool spec_flag_clean = false;  // the device word of SLOT_SPEC_FLAG is known to be zero
};

namespace {

void prefetch_shutdown(orcd_ctx *c);

#define HIP_TRY(ctx, expr)                     \
    do {                                       \
        hipError_t e__ = (expr);               \
        if (e__ != hipSuccess) {               \
            (ctx)->last_hip = e__;             \
            return ORCD_HIP_ERROR;             \
        }                                      \
    } while (0)

#define ORCD_TRY(expr)                 \
    do {                               \
        int s__ = (expr);              \
        if (s__ != ORCD_OK) return s__; \
    } while (0)

// Grows a recycled device buffer.  Earlier work on the stream may still read the old buffer, so
// the stream is drained before it is released.
int ensure(orcd_ctx *c, Slot s, size_t bytes, void **out) {
    if (bytes == 0) bytes = 16;
    if (c->cap[s] < bytes) {
        if (c->slot[s]) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, hipFree(c->slot[s]));
            c->slot[s] = nullptr;
            c->cap[s] = 0;
        }
        size_t want = bytes + bytes / 4;
        want = (want + 255) & ~(size_t)255;
        HIP_TRY(c, hipMalloc(&c->slot[s], want));
        c->cap[s] = want;
    }
    *out = c->slot[s];
    return ORCD_OK;
}

// ---- large transfers between host and device ---------------------------------------------------
// A single thread moves a few GB/s out of the page cache or between pageable buffers (and the
// runtime's own staging of a pageable hipMemcpy is one such thread), far less than the link.
// Here N_READERS threads take the chunks of a transfer in turn, each through a pinned chunk of
// its own: host-side copies (or file reads) overlap each other and the DMA.
constexpr size_t STAGE_CHUNK = 8u << 20;
constexpr size_t THREADED_COPY_MIN = 32u << 20;  // below this a plain hipMemcpyAsync is used

struct HostSource {       // where the bytes of a transfer to the device come from
    orcd_file *file;      // a file region starting at file_off ...
    uint64_t file_off;
    const uint8_t *mem;   // ... or host memory
};

// one chunk-sized step of worker t: returns an ORCD status, hip errors in *hip.  Chunks are as
// large as the pinned chunks for long transfers and down to CHUNK_MIN for short ones, so that a
// transfer of a few megabytes still spreads over the workers.
constexpr size_t CHUNK_MIN = 1u << 20;

template <typename Step>
int run_chunked(orcd_ctx *c, StageSet &ss, uint64_t len, Step step) {
    hipStream_t st = ss.stream ? ss.stream : c->stream;
    for (int i = 0; i < N_READERS; i++) {
        if (!ss.h_stage[i]) HIP_TRY(c, hipHostMalloc((void **)&ss.h_stage[i], STAGE_CHUNK, hipHostMallocDefault));
        if (!ss.ev_stage[i]) HIP_TRY(c, hipEventCreateWithFlags(&ss.ev_stage[i], hipEventDisableTiming));
    }
    uint64_t chunk = (len / N_READERS + CHUNK_MIN - 1) / CHUNK_MIN * CHUNK_MIN;
    chunk = std::min<uint64_t>(std::max<uint64_t>(chunk, CHUNK_MIN), STAGE_CHUNK);
    const uint64_t n_chunks = (len + chunk - 1) / chunk;
    const int n_threads = (int)std::min<uint64_t>(N_READERS, n_chunks);
    std::atomic<int> status{ORCD_OK};
    std::atomic<int> hip_status{(int)hipSuccess};
    std::atomic<uint64_t> next{0};
    // worker t owns pinned chunk t; the chunks of the transfer go to whoever is free
    auto worker = [&](int t) {
        if (hipSetDevice(c->device) != hipSuccess) {
            status = ORCD_HIP_ERROR;
            return;
        }
        for (uint64_t k; (k = next.fetch_add(1)) < n_chunks && status == ORCD_OK;) {
            const uint64_t pos = k * chunk, n = std::min<uint64_t>(chunk, len - pos);
            hipError_t e = hipSuccess;
            const int rc = step(t, pos, n, &e);
            if (rc != ORCD_OK) {
                if (e != hipSuccess) hip_status = (int)e;
                status = rc;
                return;
            }
        }
    };
    // (this thread and whatever helpers the process's pool can spare, orcd_helper_threads)
    if (n_chunks) orcd_parallel(n_threads - 1, worker);
    if (status != ORCD_OK) {
        if (hip_status != (int)hipSuccess) c->last_hip = (hipError_t)hip_status.load();
        (void)hipStreamSynchronize(st);  // queued copies still use the pinned chunks
        for (int i = 0; i < N_READERS; i++) ss.stage_busy[i] = false;
        return status;
    }
    return ORCD_OK;
}

// src -> d_dst, queued on the set's stream (the copies may still be running on return)
int chunked_to_device(orcd_ctx *c, StageSet &ss, const HostSource &src, uint64_t len, uint8_t *d_dst) {
    hipStream_t st = ss.stream ? ss.stream : c->stream;
    return run_chunked(c, ss, len, [&](int t, uint64_t pos, uint64_t n, hipError_t *e) -> int {
        if (ss.stage_busy[t]) {  // the chunk's last copy is done
            *e = hipEventSynchronize(ss.ev_stage[t]);
            if (*e != hipSuccess) return ORCD_HIP_ERROR;
            ss.stage_busy[t] = false;
        }
        if (src.file) {
            const int rc = orcd_file_read_at(src.file, src.file_off + pos, n, ss.h_stage[t]);
            if (rc != ORCD_OK) return rc;
        } else {
            memcpy(ss.h_stage[t], src.mem + pos, n);
        }
        *e = hipMemcpyAsync(d_dst + pos, ss.h_stage[t], n, hipMemcpyHostToDevice, st);
        if (*e == hipSuccess) *e = hipEventRecord(ss.ev_stage[t], st);
        if (*e != hipSuccess) return ORCD_HIP_ERROR;
        ss.stage_busy[t] = true;
        return ORCD_OK;
    });
}
int chunked_to_device(orcd_ctx *c, const HostSource &src, uint64_t len, uint8_t *d_dst) {
    return chunked_to_device(c, c->stage, src, len, d_dst);
}

// d_src -> h_dst; everything queued on the stream before has finished and the bytes are in
// h_dst on return
int chunked_to_host(orcd_ctx *c, const uint8_t *d_src, uint64_t len, uint8_t *h_dst) {
    StageSet &ss = c->stage;
    return run_chunked(c, ss, len, [&](int t, uint64_t pos, uint64_t n, hipError_t *e) -> int {
        if (ss.stage_busy[t]) {
            *e = hipEventSynchronize(ss.ev_stage[t]);
            if (*e != hipSuccess) return ORCD_HIP_ERROR;
            ss.stage_busy[t] = false;
        }
        *e = hipMemcpyAsync(ss.h_stage[t], d_src + pos, n, hipMemcpyDeviceToHost, c->stream);
        if (*e == hipSuccess) *e = hipEventRecord(ss.ev_stage[t], c->stream);
        if (*e == hipSuccess) *e = hipEventSynchronize(ss.ev_stage[t]);
        if (*e != hipSuccess) return ORCD_HIP_ERROR;
        memcpy(h_dst + pos, ss.h_stage[t], n);
        return ORCD_OK;
    });
}

// Host memory that the runtime does not know as pinned: its hipMemcpy would be staged by one
// thread of the runtime.
bool is_pageable(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return true;
    }
    return a.type == hipMemoryTypeUnregistered;
}

// Device view of an input: the pointer itself, or a staged copy of host memory.
int stage_in(orcd_ctx *c, Slot s, const void *p, size_t bytes, bool on_device, const void **dev) {
    if (on_device) {
        *dev = p;
        return ORCD_OK;
    }
    void *d;
    ORCD_TRY(ensure(c, s, bytes, &d));
    if (bytes >= THREADED_COPY_MIN && is_pageable(p)) {
        const HostSource src = {nullptr, 0, (const uint8_t *)p};
        ORCD_TRY(chunked_to_device(c, src, bytes, (uint8_t *)d));
    } else if (bytes) {
        HIP_TRY(c, hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, c->stream));
    }
    *dev = d;
    return ORCD_OK;
}

void mark(orcd_ctx *c, int i) {
    if (hipEventRecord(c->ev[i], c->stream) == hipSuccess) c->ev_valid[i] = true;
}

void reset_events(orcd_ctx *c) {
    for (int i = 0; i < N_EVENTS; i++) c->ev_valid[i] = false;
}

struct ScanResult {
    SegTable table;
    ScanTotals totals;
};

// Waits until the kernel that writes the pinned totals has stamped them with `stamp`: the host
// watches the stamp for a while (a queued copy and a wait for the stream cost 10-15 microseconds
// more than the kernels they wait for) before it falls back to waiting for the stream.
// d_totals (nullable): the device copy of the totals, read when the stamp never came.
int wait_totals_stamp(orcd_ctx *c, uint32_t stamp, const void *d_totals) {
    volatile ScanTotals *h = c->h_totals;
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (uint32_t spins = 0; !(seen = h->pad == stamp); spins++) {
        if ((spins & 255) == 255 &&
            std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300))
            break;
    }
    if (!seen) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (h->pad != stamp) {  // the kernel did not reach its end: read what the device holds
            if (!d_totals) return ORCD_HIP_ERROR;
            HIP_TRY(c, hipMemcpy(c->h_totals, d_totals, sizeof(ScanTotals), hipMemcpyDeviceToHost));
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return ORCD_OK;
}

uint32_t next_stamp(orcd_ctx *c) { return ++c->totals_stamp ? c->totals_stamp : ++c->totals_stamp; }  // never 0

// Header count + prefix sums + read-back of the totals (the one host sync of a decode).
template <typename CountFn>
int count_and_scan(orcd_ctx *c, uint32_t n_seg, uint64_t num_values, CountFn count, ScanResult *res) {
    const uint32_t n_blocks = (n_seg + SEG_BLOCK - 1) / SEG_BLOCK;
    void *seg_count, *seg_off, *blk, *boff, *totals;
    ORCD_TRY(ensure(c, SLOT_SEG_COUNT, (size_t)n_seg * sizeof(SegCount), &seg_count));
    ORCD_TRY(ensure(c, SLOT_SEG_OFF, (size_t)n_seg * sizeof(SegOffset), &seg_off));
    ORCD_TRY(ensure(c, SLOT_BLK_TOTALS, (size_t)n_blocks * sizeof(BlockTotals), &blk));
    ORCD_TRY(ensure(c, SLOT_BLK_OFF, (size_t)n_blocks * sizeof(BlockOffset), &boff));
    ORCD_TRY(ensure(c, SLOT_TOTALS, sizeof(ScanTotals), &totals));
    SegTable &t = res->table;
    t.seg_count = (SegCount *)seg_count;
    t.seg_local = (SegOffset *)seg_off;
    t.blk = (BlockTotals *)blk;
    t.boff = (BlockOffset *)boff;
    t.totals = (ScanTotals *)totals;
    // The scan kernel writes the totals straight into host memory and stamps them.
    t.host_totals = c->h_totals;
    t.host_stamp = next_stamp(c);
    HIP_TRY(c, count(t));
    HIP_TRY(c, launch_block_totals_scan(c->stream, t, n_blocks, num_values));
    ORCD_TRY(wait_totals_stamp(c, t.host_stamp, totals));
    res->totals = *c->h_totals;
    return ORCD_OK;
}

uint32_t segments_for(size_t n_entries) { return n_entries ? (uint32_t)n_entries : 1u; }

// Entry points for a long stream that came without any: stretches of equal-sized runs are
// confirmed in parallel (launch_rlev2_verify_stride) and cut every ENTRY_RUNS runs; whatever does
// not follow the pattern is left to the last segment's serial walk.  Only a matter of speed: the
// header scan validates every segment whatever the entries are.
constexpr uint64_t GUESS_MIN_BYTES = 1u << 18;
constexpr uint64_t ENTRY_RUNS = 32;
constexpr int GUESS_MAX_ROUNDS = 48;

int guess_entry_points(orcd_ctx *c, const uint8_t *d_stream, size_t len, int is_signed,
                       std::vector<uint64_t> &entries) {
    void *d_tmp;
    ORCD_TRY(ensure(c, SLOT_FLAGS, 64, &d_tmp));
    uint64_t *d_word = (uint64_t *)d_tmp;
    uint64_t p = 0;
    int short_rounds = 0;
    entries.clear();
    entries.push_back(0);
    for (int round = 0; round < GUESS_MAX_ROUNDS && p < len; round++) {
        HIP_TRY(c, launch_rlev2_probe(c->stream, d_stream, len, p, is_signed, d_word));
        HIP_TRY(c, hipMemcpyAsync(c->h_scalars, d_word, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const uint64_t stride = c->h_scalars[0];
        if (stride == 0) break;  // not a header: the scan will say what is wrong
        const uint64_t n = (len - p) / stride;
        c->h_scalars[1] = n;
        HIP_TRY(c, hipMemcpyAsync(d_word, c->h_scalars + 1, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, launch_rlev2_verify_stride(c->stream, d_stream, len, p, stride, n, is_signed, d_word));
        HIP_TRY(c, hipMemcpyAsync(c->h_scalars, d_word, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const uint64_t good = c->h_scalars[0];  // runs p, p + stride, ... confirmed
        if (good == 0) break;
        if (p != entries.back()) entries.push_back(p);
        for (uint64_t k = ENTRY_RUNS; k < good; k += ENTRY_RUNS) entries.push_back(p + k * stride);
        p += good * stride;
        // no long stretches here: stop paying a round per few runs (what is left is walked by
        // the speculative entries of rlev2_table, or serially)
        short_rounds = good < ENTRY_RUNS ? short_rounds + 1 : 0;
        if (short_rounds >= 2) break;
    }
    if (p < len && p != entries.back()) entries.push_back(p);  // the rest: one serial segment
    return ORCD_OK;
}

// How many runs per segment the counting pass stages (0: none, the run table is written by a second
// walk).  Many segments: by the bytes an average segment holds.  Few segments (a stream without
// entry points, a small column): every run there can be -- a run takes at least two stream bytes --
// as long as that stays a modest amount of memory; the walk of such a stream is one lane's serial
// work (about half a microsecond per run), and staging spares the second one.
uint32_t pick_staged_cap(uint64_t len, uint32_t n_seg) {
    if (n_seg >= 64 && n_seg <= COOP_MAX_SEGMENTS) {
        // A column of a file: an entry point per row group of 10 000 values, which is 20 runs of 512
        // or 1 500 of a handful (string lengths, low-cardinality keys).  Whatever is not staged is
        // emitted by a second walk, one lane per segment at half a microsecond a run -- 1 ms for
        // such a segment, with a few hundred lanes on the whole chip.  So: room for every run twice
        // the average segment's bytes can hold (a run takes two bytes at least), while that stays a
        // modest amount of memory.
        const uint64_t avg = len / n_seg;
        const uint64_t cap = (std::max<uint64_t>(avg, 64) + 63) / 64 * 64;
        if (cap * n_seg * sizeof(uint64_t) <= (256ull << 20) && cap <= 0xFFFFFF00ull)
            return (uint32_t)std::max<uint64_t>(cap, staged_runs_cap(len, n_seg));
    }
    if (n_seg >= 64) return staged_runs_cap(len, n_seg);
    const uint64_t cap = (len / 2 + 64) / 64 * 64;
    if (cap > 0xFFFFFF00ull || cap * n_seg * sizeof(uint64_t) > (1ull << 30)) return 0;
    return (uint32_t)cap;
}

// Builds the run table of one RLE v2 stream that is already on the device.  d_seg_ends (nullable):
// the end of every segment, for slices of a larger buffer decoded as one stream; seg_table
// (nullable) receives the segment prefix sums for the caller's own checks.
// Spacing of the speculative entry points of an irregular stream (rlev2_resync_kernel): as close
// as the wavefront-per-segment counting pass allows, and never so close that a walk that started
// off the chain has too little room to find it.
constexpr uint64_t RESYNC_MIN_SPACING = 4096;
constexpr uint64_t RESYNC_MIN_BYTES = 4 * RESYNC_MIN_SPACING;

int rlev2_table_with(orcd_ctx *c, const uint8_t *d_stream, size_t len, size_t num_values,
                     const uint64_t *d_entries, size_t n_entries, int is_signed, Rlev2Table *t,
                     bool full_table, const uint64_t *d_seg_ends, SegTable *seg_table);

int rlev2_table(orcd_ctx *c, const uint8_t *d_stream, size_t len, size_t num_values,
                const uint64_t *d_entries, size_t n_entries, int is_signed, Rlev2Table *t,
                bool full_table = false, const uint64_t *d_seg_ends = nullptr,
                SegTable *seg_table = nullptr) {
    if (n_entries > 0xFFFFFFFFull) return ORCD_INVALID_ARGUMENT;
    const bool no_entries = n_entries == 0;
    std::vector<uint64_t> guessed;
    if (n_entries == 0 && len >= GUESS_MIN_BYTES) {
        ORCD_TRY(guess_entry_points(c, d_stream, len, is_signed, guessed));
        if (guessed.size() > 1 && guessed.size() <= 0xFFFFFFFFull) {
            const void *d;
            ORCD_TRY(stage_in(c, SLOT_IN_ENTRIES, guessed.data(), guessed.size() * sizeof(uint64_t), false, &d));
            d_entries = (const uint64_t *)d;
            n_entries = guessed.size();
        }
    }
    // Not regular (what is left for one serial walk is long): speculative entry points, which the
    // counting pass verifies; if they do not hold, the stream is walked as one segment after all,
    // which also is what tells a stream that is out of spec from a guess that was off.
    const uint64_t serial_tail = len - (guessed.empty() ? 0 : guessed.back());
    if (no_entries && !d_seg_ends && serial_tail >= RESYNC_MIN_BYTES && serial_tail >= len / 2) {
        uint64_t spacing = (len + COOP_MAX_SEGMENTS - 1) / COOP_MAX_SEGMENTS;
        spacing = std::max<uint64_t>(RESYNC_MIN_SPACING, (spacing + 4095) / 4096 * 4096);
        const uint64_t k = (len + spacing - 1) / spacing;
        void *d;
        ORCD_TRY(ensure(c, SLOT_IN_ENTRIES, k * sizeof(uint64_t), &d));
        HIP_TRY(c, launch_rlev2_resync(c->stream, d_stream, len, is_signed, spacing, (uint32_t)k, (uint64_t *)d));
        const int rc = rlev2_table_with(c, d_stream, len, num_values, (const uint64_t *)d, k, is_signed, t,
                                        full_table, nullptr, seg_table);
        if (rc != ORCD_OUT_OF_SPEC) return rc;
        if (n_entries) {  // back to the entries of the regular part
            const void *g;
            ORCD_TRY(stage_in(c, SLOT_IN_ENTRIES, guessed.data(), guessed.size() * sizeof(uint64_t), false, &g));
            d_entries = (const uint64_t *)g;
        }
    }
    return rlev2_table_with(c, d_stream, len, num_values, d_entries, n_entries, is_signed, t, full_table,
                            d_seg_ends, seg_table);
}

int rlev2_table_with(orcd_ctx *c, const uint8_t *d_stream, size_t len, size_t num_values,
                     const uint64_t *d_entries, size_t n_entries, int is_signed, Rlev2Table *t,
                     bool full_table, const uint64_t *d_seg_ends, SegTable *seg_table) {
    const uint32_t n_seg = segments_for(n_entries);
    const uint64_t *entries = n_entries ? d_entries : nullptr;
    const uint64_t *seg_ends = n_entries ? d_seg_ends : nullptr;
    // The counting pass keeps the descriptors it parses, and the run table is then written
    // without a second walk of the header chain.
    void *staged = nullptr;
    const uint32_t staged_cap = pick_staged_cap(len, n_seg);
    if (staged_cap) ORCD_TRY(ensure(c, SLOT_STAGED_RUNS, (size_t)n_seg * staged_cap * sizeof(uint64_t), &staged));
    ScanResult sr;
    ORCD_TRY(count_and_scan(c, n_seg, num_values, [&](const SegTable &st) {
        if (n_seg <= COOP_MAX_SEGMENTS)
            return launch_rlev2_coop_count(c->stream, d_stream, len, entries, seg_ends, n_seg, is_signed, st,
                                           (uint64_t *)staged, staged_cap);
        return launch_rlev2_count(c->stream, d_stream, len, entries, seg_ends, n_seg, is_signed, st,
                                  (uint64_t *)staged, staged_cap);
    }, &sr));
    mark(c, 1);
    if (seg_table) *seg_table = sr.table;
    if (sr.totals.error) return (int)sr.totals.error;
    if (sr.totals.total_runs >= 0xFFFFFFFFull) return ORCD_UNSUPPORTED;
    const uint64_t n_slots = ((sr.totals.total_vals + WINDOW - 1) >> WINDOW_SHIFT) + 1;
    if (n_slots > 0xFFFFFFFFull) return ORCD_UNSUPPORTED;
    void *runs, *groups, *group_runs;
    ORCD_TRY(ensure(c, SLOT_RUNS, (size_t)(sr.totals.total_runs + 1) * sizeof(uint64_t), &runs));
    ORCD_TRY(ensure(c, SLOT_GROUPS, (size_t)n_slots * sizeof(uint32_t), &groups));
    ORCD_TRY(ensure(c, SLOT_GROUP_RUNS, (size_t)n_slots * sizeof(RunEntry), &group_runs));
    if (staged)
        HIP_TRY(c, launch_rlev2_emit_staged(c->stream, d_stream, len, entries, seg_ends, n_seg, is_signed,
                                            sr.table, (const uint64_t *)staged, staged_cap, (uint64_t *)runs,
                                            (uint32_t *)groups, (RunEntry *)group_runs, (uint32_t)n_slots,
                                            full_table, sr.totals.max_seg_runs));
    if (!staged || sr.totals.max_seg_runs > staged_cap)
        HIP_TRY(c, launch_rlev2_emit(c->stream, d_stream, len, entries, seg_ends, n_seg, is_signed, sr.table,
                                     (uint64_t *)runs, (uint32_t *)groups, (RunEntry *)group_runs,
                                     (uint32_t)n_slots, full_table, staged ? staged_cap : 0));
    mark(c, 2);
    const uint64_t covered = std::min<uint64_t>(num_values, sr.totals.total_vals);
    t->stream = d_stream;
    t->len = len;
    t->runs = (const uint64_t *)runs;
    t->group_first = (const uint32_t *)groups;
    t->group_run = (const RunEntry *)group_runs;
    t->n_groups = (uint32_t)((covered + WINDOW - 1) >> WINDOW_SHIFT);
    t->n_runs = (uint32_t)sr.totals.total_runs;
    t->num_values = num_values;
    t->is_signed = is_signed;
    return ORCD_OK;
}

// RLE v2 decode with device-side input and output.
// The regular column -- every 512 values one Direct run, equally many equal-sized runs between two
// entry points -- decoded by one launch with no header scan in front of it (see
// rlev2_strided_values_kernel); *done says whether that worked out.  Whether a stream is like that
// is only known afterwards: a failed attempt costs a few microseconds (its wavefronts leave as soon
// as one has raised the flag), and after two in a row the next SPEC_BACKOFF eligible calls go
// straight to the general path -- twice as many after every further failure, up to 1024, until an
// attempt succeeds.  (On a column of 10^9 values a failed attempt measured 0.16 to 0.22 ms: two
// million wavefronts that each fetch an entry point and a header before they can tell.)
constexpr size_t SPEC_MIN_VALUES = 8 * WINDOW;
constexpr size_t SELFPLAN_MAX_BYTES_PER_WINDOW = 832;
constexpr uint32_t SPEC_BACKOFF = 16;

// Whether a call is for the speculative paths at all, and the count-down after failed attempts.
bool spec_gate(orcd_ctx *c, size_t len, size_t num_values, size_t n_entries) {
    // ORCD_NO_SPECULATION=1 in the environment: always the general path (for profiles of it)
    static const bool disabled = [] {
        const char *e = getenv("ORCD_NO_SPECULATION");
        return e && *e && *e != '0';
    }();
    if (disabled) return false;
    if (num_values < SPEC_MIN_VALUES || n_entries > 0xFFFFFFFFull) return false;
    const uint64_t n_win = (num_values + WINDOW - 1) >> WINDOW_SHIFT;
    const uint64_t n_seg = n_entries ? n_entries : 1;
    if (n_win > 0xFFFFFFFFull || n_seg > n_win || len < n_win * 3) return false;
    if (c->spec_skip) {
        c->spec_skip--;
        return false;
    }
    return true;
}

void spec_account(orcd_ctx *c, bool done) {
    if (done) {
        c->spec_fail_streak = 0;
    } else if (++c->spec_fail_streak >= 2) {
        // 16 calls, then 32, 64, ... up to 1024 while the attempts keep failing
        const uint32_t doublings = std::min<uint32_t>(c->spec_fail_streak - 2, 6);
        c->spec_skip = SPEC_BACKOFF << doublings;
    }
}

// the context's flag word is zero between launches: the publishing kernel leaves it so
int spec_flag(orcd_ctx *c, uint32_t **flag) {
    void *f;
    ORCD_TRY(ensure(c, SLOT_SPEC_FLAG, 256, &f));
    if (!c->spec_flag_clean) {
        HIP_TRY(c, hipMemsetAsync(f, 0, 256, c->stream));
        c->spec_flag_clean = true;
    }
    *flag = (uint32_t *)f;
    return ORCD_OK;
}

int spec_verdict(orcd_ctx *c, uint32_t stamp, bool *done) {
    const int rc = wait_totals_stamp(c, stamp, nullptr);
    if (rc != ORCD_OK) {
        c->spec_flag_clean = false;
        return rc;
    }
    *done = c->h_totals->error == 0;
    return ORCD_OK;
}

// flag_word (nullable): the device word to use instead of the context's own, zero on entry.
int attempt_strided(orcd_ctx *c, const uint8_t *d_stream, size_t len, size_t num_values,
                    const uint64_t *d_entries, size_t n_entries, int is_signed, int64_t *d_out, bool *done,
                    uint32_t *flag_word = nullptr, uint32_t stride_vals = 0) {
    *done = false;
    if (((uintptr_t)d_out) & 15) return ORCD_OK;
    const uint64_t n_win = (num_values + WINDOW - 1) >> WINDOW_SHIFT;
    const uint64_t n_seg = n_entries ? n_entries : 1;
    uint32_t *flag = flag_word;
    if (!flag) ORCD_TRY(spec_flag(c, &flag));
    const uint32_t stamp = next_stamp(c);
    HIP_TRY(c, launch_rlev2_strided_values(c->stream, d_stream, len, n_entries ? d_entries : nullptr,
                                           (uint32_t)n_seg, (uint32_t)n_win, num_values, is_signed, d_out,
                                           flag, c->h_totals, stamp, stride_vals));
    mark(c, 3);  // (phases 1 and 2, header count and run table, did not run)
    c->last_kernel = "rlev2_strided_values_kernel";
    return spec_verdict(c, stamp, done);
}

// The column of full runs of any kind (see rlev2_regular_plan_kernel): the plan of the windows and
// the value kernel behind it in one go, no prefix sums and no wait in between.
int attempt_planned(orcd_ctx *c, const uint8_t *d_stream, size_t len, size_t num_values,
                    const uint64_t *d_entries, size_t n_entries, int is_signed, int64_t *d_out, bool *done,
                    uint32_t stride_vals = 0) {
    *done = false;
    const uint64_t n_win = (num_values + WINDOW - 1) >> WINDOW_SHIFT;
    uint32_t r_lo, r_hi;
    const bool even = rlev2_regular_plan_shape((uint32_t)n_entries, (uint32_t)n_win, &r_lo, &r_hi);
    if (n_entries < 2 || (!even && !stride_vals)) return ORCD_OK;
    uint32_t *flag;
    ORCD_TRY(spec_flag(c, &flag));
    // Plan and expansion in one launch, or the plan as a pass of its own ahead of the expansion?
    // The first spares the pass (0.12 ms per 10^9 values) and costs every window the search for
    // its header and the header's decoding (0.09 to 0.16 ms), which weighs the more the longer the
    // window's bytes take to arrive anyway: measured faster up to about 800 stream bytes per
    // window (10-bit deltas: 1.69 against 1.74 ms), slower above (16-bit: 1.83 against 1.79).
    // Segments of unequal run counts (stride_vals) are only for the first.
    // (a stream too long for the last-level cache: the one launch loads it with the nontemporal
    // hint, which the separate expansion cannot use, and then wins at every width -- 16-bit deltas
    // 1.68 against 1.79 ms, 24-bit 1.85 against 1.97, 32-bit 2.02 against 2.11 per 10^9 values)
    const bool dense = len / n_win <= SELFPLAN_MAX_BYTES_PER_WINDOW || len > SELFPLAN_STREAMING_BYTES;
    bool selfplan = (stride_vals || dense || !even) && !(((uintptr_t)d_out) & 15) &&
                    rlev2_selfplan_shape((uint32_t)n_entries, (uint32_t)n_win, stride_vals);
    // (a context whose last such column was one the single launch gave up on and the separate pass
    // decoded -- PatchedBase runs -- goes to that pass directly for a while)
    if (selfplan && even && !stride_vals && c->selfplan_rest > 0) {
        c->selfplan_rest--;
        selfplan = false;
    }
    const bool tried_selfplan = selfplan;
    if (selfplan) {
        void *seg_table;
        ORCD_TRY(ensure(c, SLOT_GROUPS, rlev2_selfplan_table_bytes((uint32_t)n_entries), &seg_table));
        HIP_TRY(c, launch_rlev2_selfplan_values(c->stream, d_stream, len, d_entries, (uint32_t)n_entries, (uint32_t)n_win,
                                                stride_vals, num_values, is_signed, d_out, flag, seg_table));
        mark(c, 3);
        c->last_kernel = "rlev2_selfplan_values_kernel";
        const uint32_t stamp = next_stamp(c);
        HIP_TRY(c, launch_spec_publish(c->stream, flag, c->h_totals, stamp));
        ORCD_TRY(spec_verdict(c, stamp, done));
        if (*done) {
            c->h_totals->error_seg = 1;  // (no word on whether the strided kernel would do: stay with this path)
            return ORCD_OK;
        }
        // not a column of like runs a regular distance apart: the plan as a pass of its own copes
        // with runs of any size (its lanes walk where their guesses miss)
    }
    if (stride_vals || !even) return ORCD_OK;  // (that pass wants equally many runs in every segment)
    void *group_runs;
    ORCD_TRY(ensure(c, SLOT_GROUP_RUNS, (size_t)(n_win + 1) * sizeof(RunEntry), &group_runs));
    HIP_TRY(c, launch_rlev2_regular_plan(c->stream, d_stream, len, d_entries, (uint32_t)n_entries, (uint32_t)n_win,
                                         num_values, is_signed, (RunEntry *)group_runs, flag));
    mark(c, 1);
    mark(c, 2);
    Rlev2Table t;
    t.stream = d_stream;
    t.len = len;
    t.runs = nullptr;  // every window is one run: the table of the shared windows is never looked at
    t.group_first = nullptr;
    t.group_run = (const RunEntry *)group_runs;
    t.n_groups = (uint32_t)n_win;
    t.n_runs = 0;
    t.num_values = num_values;
    t.is_signed = is_signed;
    HIP_TRY(c, launch_rlev2_planned_values(c->stream, t, d_out, flag));
    mark(c, 3);
    c->last_kernel = "rlev2_planned_values_kernel";
    const uint32_t stamp = next_stamp(c);
    HIP_TRY(c, launch_spec_publish(c->stream, flag, c->h_totals, stamp));
    ORCD_TRY(spec_verdict(c, stamp, done));
    if (*done && tried_selfplan) c->selfplan_rest = 15;
    return ORCD_OK;
}

int try_strided(orcd_ctx *c, const uint8_t *d_stream, size_t len, size_t num_values,
                const uint64_t *d_entries, size_t n_entries, int is_signed, int64_t *d_out, bool *done,
                uint32_t *flag_word = nullptr) {
    *done = false;
    if ((((uintptr_t)d_out) & 15) || !spec_gate(c, len, num_values, n_entries)) return ORCD_OK;
    ORCD_TRY(attempt_strided(c, d_stream, len, num_values, d_entries, n_entries, is_signed, d_out, done, flag_word));
    spec_account(c, *done);
    return ORCD_OK;
}

int rlev2_device(orcd_ctx *c, const uint8_t *d_stream, size_t len, size_t num_values,
                 const uint64_t *d_entries, size_t n_entries, int is_signed, int64_t *d_out,
                 uint64_t stride_vals = 0) {
    c->last_path = ORCD_PATH_GENERAL;
    if (stride_vals >= (1u << 23) || stride_vals < WINDOW) stride_vals = 0;  // (not a stride the paths below take)
    if (spec_gate(c, len, num_values, n_entries)) {
        // the way that worked last time first, then the other
        bool done = false;
        const int first = c->spec_pref;
        for (int attempt = 0; attempt < 2 && !done; attempt++) {
            const int way = attempt == 0 ? first
                                         : (first == ORCD_PATH_STRIDED ? ORCD_PATH_PLANNED : ORCD_PATH_STRIDED);
            if (way == ORCD_PATH_STRIDED)
                ORCD_TRY(attempt_strided(c, d_stream, len, num_values, d_entries, n_entries, is_signed, d_out, &done,
                                         nullptr, (uint32_t)stride_vals));
            else
                ORCD_TRY(attempt_planned(c, d_stream, len, num_values, d_entries, n_entries, is_signed, d_out, &done,
                                         (uint32_t)stride_vals));
            if (done) {
                c->spec_pref = c->last_path = way;
                // equal-sized Direct runs all through: the strided kernel needs no plan for such a
                // column (the plan noted it beside its verdict)
                if (way == ORCD_PATH_PLANNED && c->h_totals->error_seg == 0) c->spec_pref = ORCD_PATH_STRIDED;
            }
        }
        spec_account(c, done);
        if (done) return ORCD_OK;
    }
    Rlev2Table t;
    ORCD_TRY(rlev2_table(c, d_stream, len, num_values, d_entries, n_entries, is_signed, &t));
    HIP_TRY(c, launch_rlev2_values(c->stream, t, d_out));
    mark(c, 3);
    c->last_kernel = rlev2_values_short_variant(t) ? "rlev2_values_kernel<0, true>" : "rlev2_values_kernel<0, false>";
    return ORCD_OK;
}

// Integer RLE version 1, device-side input and output.
int rlev1_device(orcd_ctx *c, const uint8_t *d_stream, size_t len, size_t num_values,
                 const uint64_t *d_entries, size_t n_entries, int is_signed, int64_t *d_out) {
    if (n_entries > 0xFFFFFFFFull) return ORCD_INVALID_ARGUMENT;
    const uint32_t n_seg = segments_for(n_entries);
    const uint64_t *entries = n_entries ? d_entries : nullptr;
    ScanResult sr;
    ORCD_TRY(count_and_scan(c, n_seg, num_values, [&](const SegTable &st) {
        return launch_rlev1_count(c->stream, d_stream, len, entries, n_seg, is_signed, st);
    }, &sr));
    mark(c, 1);
    if (sr.totals.error) return (int)sr.totals.error;
    HIP_TRY(c, launch_rlev1_emit(c->stream, d_stream, len, entries, n_seg, is_signed, sr.table, num_values, d_out));
    mark(c, 3);
    return ORCD_OK;
}

constexpr size_t BYTE_RLE_COMPOSE_MIN = 8 * BC_SEG;

struct ByteRleTable {
    const ByteRun *runs;
    const uint32_t *win_first;
    uint32_t n_win_slots;
    uint32_t n_runs;
};

// Builds the run table of one byte-RLE stream that is already on the device.
// whole_stream: every run the stream holds is wanted, however many bytes that makes (num_bytes is
// then the largest number there is, and running out of stream before it is no error).
int byte_rle_table(orcd_ctx *c, const uint8_t *d_stream, size_t len, size_t num_bytes,
                   const uint64_t *d_entries, size_t n_entries, ByteRleTable *t,
                   const uint64_t *d_seg_ends = nullptr, SegTable *seg_table = nullptr,
                   bool whole_stream = false, uint64_t *total_bytes = nullptr) {
    if (n_entries > 0xFFFFFFFFull) return ORCD_INVALID_ARGUMENT;
    // a long stream without entry points: they are computed (exactly) in front of the scan
    if (n_entries == 0 && !d_seg_ends && len >= BYTE_RLE_COMPOSE_MIN && len / BC_SEG < 0xFFFFFFF0ull) {
        uint32_t k = 0;
        const size_t work_bytes = byte_rle_compose_bytes(len, &k);
        void *work, *ent;
        ORCD_TRY(ensure(c, SLOT_COMPOSE, work_bytes, &work));
        ORCD_TRY(ensure(c, SLOT_IN_ENTRIES, (size_t)k * sizeof(uint64_t), &ent));
        HIP_TRY(c, launch_byte_rle_compose_entries(c->stream, d_stream, len, (uint8_t *)work, (uint64_t *)ent));
        d_entries = (const uint64_t *)ent;
        n_entries = k;
    }
    const uint32_t n_seg = segments_for(n_entries);
    const uint64_t *entries = n_entries ? d_entries : nullptr;
    const uint64_t *seg_ends = n_entries ? d_seg_ends : nullptr;
    void *staged = nullptr;  // as in rlev2_table
    const uint32_t staged_cap = pick_staged_cap(len, n_seg);
    if (staged_cap) ORCD_TRY(ensure(c, SLOT_STAGED_RUNS, (size_t)n_seg * staged_cap * sizeof(uint64_t), &staged));
    ScanResult sr;
    ORCD_TRY(count_and_scan(c, n_seg, num_bytes, [&](const SegTable &st) {
        return launch_byte_rle_count(c->stream, d_stream, len, entries, seg_ends, n_seg, st, (uint64_t *)staged,
                                     staged_cap);
    }, &sr));
    mark(c, 1);
    if (seg_table) *seg_table = sr.table;
    // (no segment named: no walk stopped early, the stream just holds fewer bytes than asked for)
    const bool only_short = sr.totals.error == ORCD_OUT_OF_SPEC && sr.totals.error_seg == 0xFFFFFFFFu;
    if (sr.totals.error && !(whole_stream && only_short)) return (int)sr.totals.error;
    if (total_bytes) *total_bytes = sr.totals.total_vals;
    if (sr.totals.total_runs >= 0xFFFFFFFFull) return ORCD_UNSUPPORTED;
    const size_t n_slots = byte_rle_window_slots(sr.totals.total_vals);
    if (n_slots > 0xFFFFFFFFull) return ORCD_UNSUPPORTED;
    void *runs, *wins;
    ORCD_TRY(ensure(c, SLOT_RUNS, (size_t)(sr.totals.total_runs + 1) * sizeof(ByteRun), &runs));
    ORCD_TRY(ensure(c, SLOT_GROUPS, n_slots * sizeof(uint32_t), &wins));
    if (staged)
        HIP_TRY(c, launch_byte_rle_emit_staged(c->stream, n_seg, sr.table, (const uint64_t *)staged, staged_cap,
                                               (ByteRun *)runs, (uint32_t *)wins, (uint32_t)n_slots,
                                               sr.totals.max_seg_runs));
    if (!staged || sr.totals.max_seg_runs > staged_cap)
        HIP_TRY(c, launch_byte_rle_emit(c->stream, d_stream, len, entries, seg_ends, n_seg, sr.table,
                                        (ByteRun *)runs, (uint32_t *)wins, (uint32_t)n_slots,
                                        staged ? staged_cap : 0));
    mark(c, 2);
    t->runs = (const ByteRun *)runs;
    t->win_first = (const uint32_t *)wins;
    t->n_win_slots = (uint32_t)n_slots;
    t->n_runs = (uint32_t)sr.totals.total_runs;
    return ORCD_OK;
}

// Byte-RLE decode with device-side input and output.
int byte_rle_device(orcd_ctx *c, const uint8_t *d_stream, size_t len, size_t num_bytes,
                    const uint64_t *d_entries, size_t n_entries, uint8_t *d_out) {
    ByteRleTable t;
    ORCD_TRY(byte_rle_table(c, d_stream, len, num_bytes, d_entries, n_entries, &t));
    HIP_TRY(c, launch_byte_rle_expand(c->stream, d_stream, len, t.runs, t.win_first, t.n_win_slots, t.n_runs,
                                      num_bytes, d_out));
    mark(c, 3);
    return ORCD_OK;
}

int copy_out(orcd_ctx *c, void *host, const void *dev, size_t bytes) {
    if (bytes >= THREADED_COPY_MIN && is_pageable(host)) {
        ORCD_TRY(chunked_to_host(c, (const uint8_t *)dev, bytes, (uint8_t *)host));
    } else if (bytes) {
        HIP_TRY(c, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ORCD_OK;
}

// A result array to the caller's host buffer, without the final wait of copy_out (the large
// pageable way waits by itself).
int result_to_host(orcd_ctx *c, void *host, const void *dev, size_t bytes) {
    if (bytes >= THREADED_COPY_MIN && is_pageable(host))
        return chunked_to_host(c, (const uint8_t *)dev, bytes, (uint8_t *)host);
    if (bytes) HIP_TRY(c, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    return ORCD_OK;
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

extern "C" {

const char *orcd_strerror(int status) {
    switch (status) {
    case ORCD_OK: return "ok";
    case ORCD_OUT_OF_SPEC: return "out of spec";
    case ORCD_INVALID_UTF8: return "invalid utf-8";
    case ORCD_INVALID_COLUMN: return "invalid column";
    case ORCD_INVALID_KIND: return "invalid stream kind for column";
    case ORCD_DECODE_FLOAT: return "failed to decode float";
    case ORCD_DECOMPRESSION: return "decompression failed";
    case ORCD_INVALID_PROTO: return "invalid protobuf";
    case ORCD_UNSUPPORTED: return "unsupported feature";
    case ORCD_HIP_ERROR: return "HIP runtime error";
    case ORCD_INVALID_ARGUMENT: return "invalid argument";
    case ORCD_NO_MEMORY: return "out of host memory";
    default: return "unknown status";
    }
}

int orcd_ctx_create(int device_id, orcd_ctx **out) try {
    if (!out) return ORCD_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return ORCD_HIP_ERROR;
    DeviceGuard g(device_id);
    orcd_ctx *c = new orcd_ctx();
    c->device = device_id;
    bool ok = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < N_EVENTS; i++) ok = hipEventCreate(&c->ev[i]) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&c->h_totals, sizeof(ScanTotals), hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess;
    if (ok) memset(c->h_totals, 0, sizeof(ScanTotals));
    ok = ok && noalloc_scratch(device_id, &c->scratch_dev) == hipSuccess;  // device_id is current here
    ok = ok && hipHostMalloc((void **)&c->h_scalars, 8 * sizeof(uint64_t), hipHostMallocDefault) == hipSuccess;
    if (!ok) {
        orcd_ctx_destroy(c);
        return ORCD_HIP_ERROR;
    }
    c->stream = c->own_stream;
    {
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device_id) == hipSuccess)
            c->have_numa_cpus = orcd_internal_numa_cpus(bus, &c->numa_cpus, sizeof(c->numa_cpus)) != 0;
            if (c->have_numa_cpus) orcd_internal_pool_bind(&c->numa_cpus, sizeof(c->numa_cpus));
    }
    *out = c;
    return ORCD_OK;
} ORCD_CATCH

int orcd_ctx_destroy(orcd_ctx *c) try {
    if (!c) return ORCD_OK;
    DeviceGuard g(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < N_SLOTS; i++)
        if (c->slot[i]) (void)hipFree(c->slot[i]);
    for (int i = 0; i < N_EVENTS; i++)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->h_totals) (void)hipHostFree(c->h_totals);
    if (c->h_scalars) (void)hipHostFree(c->h_scalars);
    if (c->inflate_pinned) (void)hipHostFree(c->inflate_pinned);
    prefetch_shutdown(c);  // (joins the background reader; its device blocks go back to the cache)
    c->stage.release();
    c->dev_cache.release_all();
    c->host_cache.release_all();
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return ORCD_OK;
} ORCD_CATCH

int orcd_ctx_set_stream(orcd_ctx *c, void *hip_stream) try {
    if (!c) return ORCD_INVALID_ARGUMENT;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return ORCD_OK;
} ORCD_CATCH

int orcd_ctx_release_cached(orcd_ctx *c) try {
    if (!c) return ORCD_INVALID_ARGUMENT;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // queued work may still use a released block
    c->dev_cache.release_all();
    c->host_cache.release_all();
    std::vector<uint8_t>().swap(c->inflate_scratch);
    if (c->inflate_pinned) (void)hipHostFree(c->inflate_pinned);
    c->inflate_pinned = nullptr;
    c->inflate_pinned_cap = 0;
    return ORCD_OK;
} ORCD_CATCH

int orcd_ctx_set_cache_limit(orcd_ctx *c, size_t device_bytes, size_t host_bytes) try {
    if (!c) return ORCD_INVALID_ARGUMENT;
    c->dev_cache_limit = device_bytes;
    c->host_cache_limit = host_bytes;
    return ORCD_OK;
} ORCD_CATCH

int orcd_ctx_copy_to_host(orcd_ctx *c, void *dst, const void *src_device, size_t bytes) try {
    if (!c || (bytes && (!dst || !src_device))) return ORCD_INVALID_ARGUMENT;
    DeviceGuard g(c->device);
    return copy_out(c, dst, src_device, bytes);
} ORCD_CATCH

int orcd_ctx_synchronize(orcd_ctx *c) try {
    if (!c) return ORCD_INVALID_ARGUMENT;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ORCD_OK;
} ORCD_CATCH

float orcd_ctx_phase_ms(orcd_ctx *c, int phase) {
    if (!c || phase < 0 || phase >= N_EVENTS) return -1.0f;
    DeviceGuard g(c->device);
    int last = N_EVENTS - 1;
    while (last > 0 && !c->ev_valid[last]) last--;
    if (!c->ev_valid[0] || last == 0) return -1.0f;
    int a = 0, b = last;
    if (phase > 0) {  // from the nearest earlier event that was recorded (a phase that did not run has none)
        if (!c->ev_valid[phase]) return -1.0f;
        a = phase - 1;
        while (a > 0 && !c->ev_valid[a]) a--;
        b = phase;
    }
    if (hipEventSynchronize(c->ev[b]) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, c->ev[a], c->ev[b]) != hipSuccess) return -1.0f;
    return ms;
}

float orcd_ctx_last_elapsed_ms(orcd_ctx *c) { return orcd_ctx_phase_ms(c, 0); }

int orcd_ctx_last_path(orcd_ctx *c) { return c ? c->last_path : -1; }

const char *orcd_ctx_last_value_kernel(orcd_ctx *c) { return c ? c->last_kernel : ""; }

int orcd_ctx_last_stripe_ms(orcd_ctx *c, float *feed_ms, float *decode_ms) {
    if (!c || c->stripe_decode_ms < 0) return ORCD_INVALID_ARGUMENT;
    if (feed_ms) *feed_ms = c->stripe_feed_ms;
    if (decode_ms) *decode_ms = c->stripe_decode_ms;
    return ORCD_OK;
}

const char *orcd_ctx_last_hip_error(orcd_ctx *c) {
    return c ? hipGetErrorString(c->last_hip) : "no context";
}

// ---- RLE v2 ---------------------------------------------------------------------------------

int orcd_rlev2_decode(orcd_ctx *c, const uint8_t *stream, size_t len, size_t num_values,
                      const uint64_t *entry_points, size_t n_entries, int64_t *out,
                      unsigned flags) try {
    if (!c || (len && !stream) || (num_values && !out) || (n_entries && !entry_points))
        return ORCD_INVALID_ARGUMENT;
    if (num_values == 0) return ORCD_OK;
    DeviceGuard g(c->device);
    reset_events(c);
    const bool in_dev = flags & ORCD_IN_DEVICE, out_dev = flags & ORCD_OUT_DEVICE;
    mark(c, 0);
    const void *d_stream, *d_entries;
    ORCD_TRY(stage_in(c, SLOT_IN_A, stream, len, in_dev, &d_stream));
    ORCD_TRY(stage_in(c, SLOT_IN_ENTRIES, entry_points, n_entries * sizeof(uint64_t),
                      in_dev && !(flags & ORCD_ENTRIES_HOST), &d_entries));
    int64_t *d_out = out;
    if (!out_dev) {
        void *p;
        ORCD_TRY(ensure(c, SLOT_OUT_A, num_values * sizeof(int64_t), &p));
        d_out = (int64_t *)p;
    }
    ORCD_TRY(rlev2_device(c, (const uint8_t *)d_stream, len, num_values, (const uint64_t *)d_entries,
                          n_entries, (flags & ORCD_SIGNED) ? 1 : 0, d_out));
    if (!out_dev) ORCD_TRY(copy_out(c, out, d_out, num_values * sizeof(int64_t)));
    return ORCD_OK;
} ORCD_CATCH

int orcd_rlev2_decode_groups(orcd_ctx *c, const uint8_t *stream, size_t len, size_t num_values,
                             const uint64_t *entry_points, size_t n_entries, uint64_t values_per_group,
                             int64_t *out, unsigned flags) try {
    if (!c || (len && !stream) || (num_values && !out) || (n_entries && !entry_points))
        return ORCD_INVALID_ARGUMENT;
    if (num_values == 0) return ORCD_OK;
    DeviceGuard g(c->device);
    reset_events(c);
    const bool in_dev = flags & ORCD_IN_DEVICE, out_dev = flags & ORCD_OUT_DEVICE;
    mark(c, 0);
    const void *d_stream, *d_entries;
    ORCD_TRY(stage_in(c, SLOT_IN_A, stream, len, in_dev, &d_stream));
    ORCD_TRY(stage_in(c, SLOT_IN_ENTRIES, entry_points, n_entries * sizeof(uint64_t),
                      in_dev && !(flags & ORCD_ENTRIES_HOST), &d_entries));
    int64_t *d_out = out;
    if (!out_dev) {
        void *p;
        ORCD_TRY(ensure(c, SLOT_OUT_A, num_values * sizeof(int64_t), &p));
        d_out = (int64_t *)p;
    }
    ORCD_TRY(rlev2_device(c, (const uint8_t *)d_stream, len, num_values, (const uint64_t *)d_entries,
                          n_entries, (flags & ORCD_SIGNED) ? 1 : 0, d_out, values_per_group));
    if (!out_dev) ORCD_TRY(copy_out(c, out, d_out, num_values * sizeof(int64_t)));
    return ORCD_OK;
} ORCD_CATCH

// The chunks' table to the device, the kernel, the sizes back (waits for the stream).
int inflate_device(orcd_ctx *c, const uint8_t *d_src, size_t src_len, const orcd_chunk *chunks, size_t n_chunks,
                   uint8_t *d_dst, size_t dst_len, uint32_t *sizes) {
    static_assert(sizeof(orcd_chunk) == sizeof(InflateChunk) && offsetof(orcd_chunk, src_length) == offsetof(InflateChunk, src_len) &&
                  offsetof(orcd_chunk, dst_offset) == offsetof(InflateChunk, dst_off), "one layout");
    if (n_chunks > 0x7FFFFFFFull) return ORCD_INVALID_ARGUMENT;
    for (size_t k = 0; k < n_chunks; k++) {
        const orcd_chunk &ck = chunks[k];
        if (ck.src_offset > src_len || ck.src_length > src_len - ck.src_offset || ck.dst_offset > dst_len ||
            ck.dst_capacity > dst_len - ck.dst_offset)
            return ORCD_INVALID_ARGUMENT;
    }
    if (n_chunks == 0) return ORCD_OK;
    const void *d_chunks;
    void *d_sizes;
    ORCD_TRY(stage_in(c, SLOT_GROUPS, chunks, n_chunks * sizeof(orcd_chunk), false, &d_chunks));
    ORCD_TRY(ensure(c, SLOT_FLAGS, std::max<size_t>(64, n_chunks * sizeof(uint32_t)), &d_sizes));
    HIP_TRY(c, launch_inflate_chunks(c->stream, d_src, src_len, (const InflateChunk *)d_chunks, (uint32_t)n_chunks, d_dst,
                                     (uint32_t *)d_sizes));
    HIP_TRY(c, hipMemcpyAsync(sizes, d_sizes, n_chunks * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t k = 0; k < n_chunks; k++)
        if (sizes[k] == 0xFFFFFFFFu) return ORCD_DECOMPRESSION;
    return ORCD_OK;
}

int orcd_inflate_chunks(orcd_ctx *c, const uint8_t *src, size_t src_len, const orcd_chunk *chunks, size_t n_chunks,
                        uint8_t *dst, size_t dst_len, uint32_t *sizes, unsigned flags) try {
    if (!c || (src_len && !src) || (dst_len && !dst) || (n_chunks && (!chunks || !sizes))) return ORCD_INVALID_ARGUMENT;
    if (n_chunks == 0) return ORCD_OK;
    DeviceGuard g(c->device);
    reset_events(c);
    const bool in_dev = flags & ORCD_IN_DEVICE, out_dev = flags & ORCD_OUT_DEVICE;
    mark(c, 0);
    const void *d_src;
    ORCD_TRY(stage_in(c, SLOT_IN_A, src, src_len, in_dev, &d_src));
    uint8_t *d_dst = dst;
    if (!out_dev) {
        void *p;
        ORCD_TRY(ensure(c, SLOT_OUT_A, dst_len, &p));
        d_dst = (uint8_t *)p;
    }
    mark(c, 1);
    const int rc = inflate_device(c, (const uint8_t *)d_src, src_len, chunks, n_chunks, d_dst, dst_len, sizes);
    mark(c, 3);
    if (rc != ORCD_OK && rc != ORCD_DECOMPRESSION) return rc;
    if (!out_dev) {
        // only the bytes that were written (the rest of the staging block is not the caller's), and
        // chunks that fill their places and follow one another -- all but the last of a stream, as
        // the writers make them -- in one copy: a stripe has thousands of chunks, and a copy to
        // pageable memory is staged by the runtime one at a time
        for (size_t k = 0; k < n_chunks;) {
            if (!sizes[k] || sizes[k] == 0xFFFFFFFFu) {
                k++;
                continue;
            }
            const uint64_t from = chunks[k].dst_offset;
            uint64_t to = from + sizes[k];
            size_t j = k + 1;
            while (j < n_chunks && sizes[j] && sizes[j] != 0xFFFFFFFFu && chunks[j].dst_offset == to) to += sizes[j++];
            HIP_TRY(c, hipMemcpyAsync(dst + from, d_dst + from, to - from, hipMemcpyDeviceToHost, c->stream));
            k = j;
        }
    }
    if (!out_dev) HIP_TRY(c, hipStreamSynchronize(c->stream));
    return rc;
} ORCD_CATCH

int orcd_rlev1_decode(orcd_ctx *c, const uint8_t *stream, size_t len, size_t num_values,
                      const uint64_t *entry_points, size_t n_entries, int64_t *out,
                      unsigned flags) try {
    if (!c || (len && !stream) || (num_values && !out) || (n_entries && !entry_points))
        return ORCD_INVALID_ARGUMENT;
    if (num_values == 0) return ORCD_OK;
    DeviceGuard g(c->device);
    reset_events(c);
    const bool in_dev = flags & ORCD_IN_DEVICE, out_dev = flags & ORCD_OUT_DEVICE;
    mark(c, 0);
    const void *d_stream, *d_entries;
    ORCD_TRY(stage_in(c, SLOT_IN_A, stream, len, in_dev, &d_stream));
    ORCD_TRY(stage_in(c, SLOT_IN_ENTRIES, entry_points, n_entries * sizeof(uint64_t),
                      in_dev && !(flags & ORCD_ENTRIES_HOST), &d_entries));
    int64_t *d_out = out;
    if (!out_dev) {
        void *p;
        ORCD_TRY(ensure(c, SLOT_OUT_A, num_values * sizeof(int64_t), &p));
        d_out = (int64_t *)p;
    }
    ORCD_TRY(rlev1_device(c, (const uint8_t *)d_stream, len, num_values, (const uint64_t *)d_entries,
                          n_entries, (flags & ORCD_SIGNED) ? 1 : 0, d_out));
    if (!out_dev) ORCD_TRY(copy_out(c, out, d_out, num_values * sizeof(int64_t)));
    return ORCD_OK;
} ORCD_CATCH

// orcd_rlev2_decode for callers that cannot allocate: device pointers only, every table in the
// library's static device array, limits instead of growth.
int orcd_rlev2_decode_noalloc(orcd_ctx *c, const uint8_t *stream, size_t len, size_t num_values,
                              const uint64_t *entry_points, size_t n_entries, int64_t *out,
                              unsigned flags) try {
    if (!c || (len && !stream) || (num_values && !out) || (n_entries && !entry_points))
        return ORCD_INVALID_ARGUMENT;
    if (num_values == 0) return ORCD_OK;
    if (num_values > NOALLOC_MAX_VALUES || n_entries > NOALLOC_MAX_SEGMENTS || len > RUN_PAYLOAD_MASK)
        return ORCD_INVALID_ARGUMENT;  // beyond what the static tables hold
    DeviceGuard g(c->device);
    unsigned char *base = c->scratch_dev;
    if (!base) {  // the context was created before the symbol could be resolved: once more, this device current
        HIP_TRY(c, noalloc_scratch(c->device, &base));
        c->scratch_dev = base;
    }
    const int is_signed = (flags & ORCD_SIGNED) ? 1 : 0;
    reset_events(c);
    mark(c, 0);
    c->last_path = ORCD_PATH_GENERAL;
    // the array keeps what the last call left: the one word that must start at zero is cleared
    uint32_t *flag = (uint32_t *)(base + NA_FLAG);
    HIP_TRY(c, hipMemsetAsync(flag, 0, 256, c->stream));
    bool done = false;
    ORCD_TRY(try_strided(c, stream, len, num_values, entry_points, n_entries, is_signed, out, &done, flag));
    if (done) {
        c->last_path = ORCD_PATH_STRIDED;
        return ORCD_OK;
    }
    const uint32_t n_seg = segments_for(n_entries);
    const uint64_t *entries = n_entries ? entry_points : nullptr;
    const uint32_t n_blocks = (n_seg + SEG_BLOCK - 1) / SEG_BLOCK;
    SegTable t;
    t.seg_count = (SegCount *)(base + NA_SEG_COUNT);
    t.seg_local = (SegOffset *)(base + NA_SEG_OFF);
    t.blk = (BlockTotals *)(base + NA_BLK);
    t.boff = (BlockOffset *)(base + NA_BOFF);
    t.totals = (ScanTotals *)(base + NA_TOTALS);
    t.host_totals = c->h_totals;
    t.host_stamp = next_stamp(c);
    if (n_seg <= COOP_MAX_SEGMENTS)
        HIP_TRY(c, launch_rlev2_coop_count(c->stream, stream, len, entries, nullptr, n_seg, is_signed, t, nullptr, 0));
    else
        HIP_TRY(c, launch_rlev2_count(c->stream, stream, len, entries, nullptr, n_seg, is_signed, t, nullptr, 0));
    HIP_TRY(c, launch_block_totals_scan(c->stream, t, n_blocks, num_values));
    ORCD_TRY(wait_totals_stamp(c, t.host_stamp, t.totals));
    mark(c, 1);
    const ScanTotals totals = *c->h_totals;
    if (totals.error) return (int)totals.error;
    const uint64_t n_slots = ((totals.total_vals + WINDOW - 1) >> WINDOW_SHIFT) + 1;
    if (totals.total_runs > NOALLOC_MAX_RUNS || n_slots > NOALLOC_MAX_WINDOWS)
        return ORCD_INVALID_ARGUMENT;  // the stream holds more than the static tables do
    uint64_t *runs = (uint64_t *)(base + NA_RUNS);
    uint32_t *groups = (uint32_t *)(base + NA_GROUPS);
    RunEntry *group_runs = (RunEntry *)(base + NA_GROUP_RUNS);
    HIP_TRY(c, launch_rlev2_emit(c->stream, stream, len, entries, nullptr, n_seg, is_signed, t, runs, groups,
                                 group_runs, (uint32_t)n_slots, false, 0));
    mark(c, 2);
    Rlev2Table tab;
    tab.stream = stream;
    tab.len = len;
    tab.runs = runs;
    tab.group_first = groups;
    tab.group_run = group_runs;
    tab.n_groups = (uint32_t)((std::min<uint64_t>(num_values, totals.total_vals) + WINDOW - 1) >> WINDOW_SHIFT);
    tab.n_runs = (uint32_t)totals.total_runs;
    tab.num_values = num_values;
    tab.is_signed = is_signed;
    HIP_TRY(c, launch_rlev2_values(c->stream, tab, out));
    mark(c, 3);
    return ORCD_OK;
} ORCD_CATCH

}  // extern "C"

namespace {

// The slices of a batch as one segment list: a slice starts a segment, and so does each of its
// entry points; every segment ends where the next one of its slice starts, or with the slice.
// `units` of a slice = decoded items it must hold (values; for boolean streams bytes of 8 values).
struct SegmentPlan {
    std::vector<uint64_t> starts, ends;
    std::vector<uint32_t> first_seg;   // per live slice: its first segment
    std::vector<uint64_t> want_off;    // ... and the units before it
    std::vector<size_t> slice_of;      // ... and its index in the caller's array
    uint64_t total_units = 0;
};

int plan_segments(const orcd_rlev2_slice *slices, size_t n_slices, size_t buffer_len, bool bits,
                  SegmentPlan &plan, size_t *bad_slice) {
    uint64_t prev_end = 0;
    for (size_t i = 0; i < n_slices; i++) {
        const orcd_rlev2_slice &sl = slices[i];
        if (sl.offset > buffer_len || sl.length > buffer_len - sl.offset || (sl.n_entries && !sl.entry_points))
            return ORCD_INVALID_ARGUMENT;
        if (sl.num_values == 0) continue;
        // the value kernels address a window's runs relative to the first of them
        if (sl.offset < prev_end) return ORCD_INVALID_ARGUMENT;
        prev_end = sl.offset + sl.length;
        plan.first_seg.push_back((uint32_t)plan.starts.size());
        plan.want_off.push_back(plan.total_units);
        plan.slice_of.push_back(i);
        plan.starts.push_back(sl.offset);
        uint64_t prev = 0;
        for (size_t j = 1; j < sl.n_entries; j++) {  // the first entry is the start of the slice
            const uint64_t e = sl.entry_points[j];
            if (e <= prev || e >= sl.length) {  // not ascending inside the slice
                if (bad_slice) *bad_slice = i;
                return ORCD_OUT_OF_SPEC;
            }
            plan.ends.push_back(sl.offset + e);
            plan.starts.push_back(sl.offset + e);
            prev = e;
        }
        plan.ends.push_back(sl.offset + sl.length);
        plan.total_units += bits ? (sl.num_values + 7) / 8 : sl.num_values;
    }
    return plan.starts.size() > 0xFFFFFFFFull ? ORCD_INVALID_ARGUMENT : ORCD_OK;
}

// Decoded as one stream, the items of a slice land where they belong only if every slice before
// it holds exactly what it was said to hold: the scan's offsets at the slice starts are compared
// with the prefix sums of the counts -- on the device, the verdict left in pinned host memory
// behind a stamp, so that the expansion can be queued behind the check and the host learns the
// verdict while it runs (a copy back and a wait for the stream left the device idle for 40
// microseconds per batch).
int verify_slices_launch(orcd_ctx *c, const SegTable &seg, const uint32_t *d_first, const uint64_t *d_want,
                         size_t n, uint32_t *stamp) {
    *stamp = next_stamp(c);
    HIP_TRY(c, launch_seg_offsets_verify(c->stream, seg, d_first, d_want, (uint32_t)n, c->h_scalars + 6, *stamp));
    return ORCD_OK;
}

int verify_slices_wait(orcd_ctx *c, uint32_t stamp, const SegmentPlan &plan, size_t *bad_slice) {
    volatile uint64_t *h = c->h_scalars + 6;
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (uint32_t spins = 0; !(seen = h[1] == stamp); spins++) {
        if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
    }
    if (!seen) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (h[1] != stamp) return ORCD_HIP_ERROR;  // the kernel did not reach its end
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    const uint64_t k = h[0];
    if (k != 0xFFFFFFFFull) {  // slice k - 1 is the first that is off
        if (bad_slice) *bad_slice = plan.slice_of[k && k <= plan.slice_of.size() ? k - 1 : 0];
        return ORCD_OUT_OF_SPEC;
    }
    return ORCD_OK;
}

// The plan's arrays in one piece, so that they reach the device in one copy: starts | ends |
// want_off | first_seg (32-bit).  Pointers into the device copy come back.
struct PlanOnDevice {
    const uint64_t *starts, *ends, *want_off;
    const uint32_t *first_seg;
    const uint64_t *extra;  // the caller's own words, behind the plan's
};

int upload_plan(orcd_ctx *c, const SegmentPlan &plan, std::vector<uint64_t> &pack, PlanOnDevice *d,
                const std::vector<uint64_t> &extra = {}) {
    const size_t n_seg = plan.starts.size(), n_sl = plan.first_seg.size();
    const size_t at_extra = 2 * n_seg + n_sl + (n_sl + 1) / 2;
    pack.resize(at_extra + extra.size());
    memcpy(pack.data(), plan.starts.data(), n_seg * sizeof(uint64_t));
    memcpy(pack.data() + n_seg, plan.ends.data(), n_seg * sizeof(uint64_t));
    memcpy(pack.data() + 2 * n_seg, plan.want_off.data(), n_sl * sizeof(uint64_t));
    memcpy(pack.data() + 2 * n_seg + n_sl, plan.first_seg.data(), n_sl * sizeof(uint32_t));
    if (!extra.empty()) memcpy(pack.data() + at_extra, extra.data(), extra.size() * sizeof(uint64_t));
    const void *p;
    ORCD_TRY(stage_in(c, SLOT_IN_ENTRIES, pack.data(), pack.size() * sizeof(uint64_t), false, &p));
    const uint64_t *base = (const uint64_t *)p;
    d->starts = base;
    d->ends = base + n_seg;
    d->want_off = base + 2 * n_seg;
    d->first_seg = (const uint32_t *)(base + 2 * n_seg + n_sl);
    d->extra = base + at_extra;
    return ORCD_OK;
}

}  // namespace

extern "C" {

int orcd_rlev2_decode_batch(orcd_ctx *c, const uint8_t *buffer, size_t buffer_len,
                            const orcd_rlev2_slice *slices, size_t n_slices, int64_t *out,
                            size_t *bad_slice, unsigned flags) try {
    if (bad_slice) *bad_slice = (size_t)-1;
    if (!c || (buffer_len && !buffer) || (n_slices && !slices)) return ORCD_INVALID_ARGUMENT;
    SegmentPlan plan;
    ORCD_TRY(plan_segments(slices, n_slices, buffer_len, false, plan, bad_slice));
    const uint64_t total_values = plan.total_units;
    if (total_values == 0) return ORCD_OK;
    if (!out) return ORCD_INVALID_ARGUMENT;
    DeviceGuard g(c->device);
    reset_events(c);
    const bool in_dev = flags & ORCD_IN_DEVICE, out_dev = flags & ORCD_OUT_DEVICE;
    mark(c, 0);
    const void *d_buffer;
    ORCD_TRY(stage_in(c, SLOT_IN_A, buffer, buffer_len, in_dev, &d_buffer));
    std::vector<uint64_t> pack;
    PlanOnDevice dp;
    ORCD_TRY(upload_plan(c, plan, pack, &dp));
    int64_t *d_out = out;
    if (!out_dev) {
        void *p;
        ORCD_TRY(ensure(c, SLOT_OUT_A, total_values * sizeof(int64_t), &p));
        d_out = (int64_t *)p;
    }
    Rlev2Table t;
    SegTable seg = {};
    const int rc = rlev2_table(c, (const uint8_t *)d_buffer, buffer_len, total_values, dp.starts,
                               plan.starts.size(), (flags & ORCD_SIGNED) ? 1 : 0, &t, false, dp.ends, &seg);
    if (!seg.seg_local) return rc;  // failed before the scan: there is nothing to look at
    uint32_t verdict_stamp;
    ORCD_TRY(verify_slices_launch(c, seg, dp.first_seg, dp.want_off, plan.first_seg.size(), &verdict_stamp));
    if (rc != ORCD_OK) {  // a run that does not parse, or too few values: in or after the last good start
        ORCD_TRY(verify_slices_wait(c, verdict_stamp, plan, bad_slice));
        if (bad_slice) *bad_slice = plan.slice_of.back();
        return rc;
    }
    // (the tables are sound whatever the verdict: the expansion writes total_values values at most)
    HIP_TRY(c, launch_rlev2_values(c->stream, t, d_out));
    mark(c, 3);
    ORCD_TRY(verify_slices_wait(c, verdict_stamp, plan, bad_slice));
    if (!out_dev) ORCD_TRY(copy_out(c, out, d_out, total_values * sizeof(int64_t)));
    return ORCD_OK;
} ORCD_CATCH

int orcd_boolean_decode_batch(orcd_ctx *c, const uint8_t *buffer, size_t buffer_len,
                              const orcd_rlev2_slice *slices, size_t n_slices, uint8_t *out_bytes,
                              uint64_t *set_bits, size_t *bad_slice, unsigned flags) try {
    if (bad_slice) *bad_slice = (size_t)-1;
    if (!c || (buffer_len && !buffer) || (n_slices && !slices)) return ORCD_INVALID_ARGUMENT;
    if (set_bits)
        for (size_t i = 0; i < n_slices; i++) set_bits[i] = 0;
    SegmentPlan plan;
    ORCD_TRY(plan_segments(slices, n_slices, buffer_len, true, plan, bad_slice));
    const uint64_t total_bytes = plan.total_units;  // decoded bytes, 8 values each
    if (total_bytes == 0) return ORCD_OK;
    if (!out_bytes) return ORCD_INVALID_ARGUMENT;
    DeviceGuard g(c->device);
    reset_events(c);
    const bool in_dev = flags & ORCD_IN_DEVICE, out_dev = flags & ORCD_OUT_DEVICE;
    mark(c, 0);
    const void *d_buffer;
    ORCD_TRY(stage_in(c, SLOT_IN_A, buffer, buffer_len, in_dev, &d_buffer));
    // (for the counts of set bits: per slice, where its values start in the output and how many
    // there are -- sent along with the plan)
    std::vector<uint64_t> where;
    if (set_bits) {
        where.resize(2 * plan.slice_of.size());
        for (size_t k = 0; k < plan.slice_of.size(); k++) {
            where[k] = plan.want_off[k] * 8;
            where[plan.slice_of.size() + k] = slices[plan.slice_of[k]].num_values;
        }
    }
    std::vector<uint64_t> pack;
    PlanOnDevice dp;
    ORCD_TRY(upload_plan(c, plan, pack, &dp, where));
    uint8_t *d_out = out_bytes;
    if (!out_dev) {
        void *p;
        ORCD_TRY(ensure(c, SLOT_OUT_A, total_bytes * 8, &p));
        d_out = (uint8_t *)p;
    }
    ByteRleTable t;
    SegTable seg = {};
    const int rc = byte_rle_table(c, (const uint8_t *)d_buffer, buffer_len, total_bytes, dp.starts,
                                  plan.starts.size(), &t, dp.ends, &seg);
    if (!seg.seg_local) return rc;
    uint32_t verdict_stamp;
    ORCD_TRY(verify_slices_launch(c, seg, dp.first_seg, dp.want_off, plan.first_seg.size(), &verdict_stamp));
    if (rc != ORCD_OK) {
        ORCD_TRY(verify_slices_wait(c, verdict_stamp, plan, bad_slice));
        if (bad_slice) *bad_slice = plan.slice_of.back();
        return rc;
    }
    // every slice fills whole bytes, so the values of slice i start at 8 * (bytes before it); the
    // up to 7 values between two slices are padding
    HIP_TRY(c, launch_bool_expand(c->stream, (const uint8_t *)d_buffer, buffer_len, t.runs, t.win_first,
                                  t.n_win_slots, t.n_runs, total_bytes * 8, d_out, nullptr, nullptr));
    mark(c, 3);
    ORCD_TRY(verify_slices_wait(c, verdict_stamp, plan, bad_slice));
    if (set_bits) {
        const uint64_t *d_where = dp.extra;
        void *d_counts;
        ORCD_TRY(ensure(c, SLOT_TMP, plan.slice_of.size() * sizeof(uint64_t), &d_counts));
        uint64_t largest = 0;
        for (size_t k = 0; k < plan.slice_of.size(); k++) largest = std::max<uint64_t>(largest, slices[plan.slice_of[k]].num_values);
        HIP_TRY(c, launch_count_nonzero_slices(c->stream, d_out, d_where, d_where + plan.slice_of.size(),
                                               (uint32_t)plan.slice_of.size(), (uint64_t *)d_counts, largest));
        std::vector<uint64_t> counts(plan.slice_of.size());
        HIP_TRY(c, hipMemcpyAsync(counts.data(), d_counts, counts.size() * sizeof(uint64_t),
                                  hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (size_t k = 0; k < counts.size(); k++) set_bits[plan.slice_of[k]] = counts[k];
    }
    if (!out_dev) ORCD_TRY(copy_out(c, out_bytes, d_out, total_bytes * 8));
    return ORCD_OK;
} ORCD_CATCH

int orcd_rlev2_runs(orcd_ctx *c, const uint8_t *stream, size_t len, size_t num_values,
                    const uint64_t *entry_points, size_t n_entries, orcd_run *runs,
                    size_t max_runs, size_t *n_runs, unsigned flags) try {
    if (!c || (len && !stream) || !n_runs || (max_runs && !runs) || (n_entries && !entry_points))
        return ORCD_INVALID_ARGUMENT;
    *n_runs = 0;
    if (num_values == 0) return ORCD_OK;
    DeviceGuard g(c->device);
    reset_events(c);
    const bool in_dev = flags & ORCD_IN_DEVICE;
    mark(c, 0);
    const void *d_stream, *d_entries;
    ORCD_TRY(stage_in(c, SLOT_IN_A, stream, len, in_dev, &d_stream));
    ORCD_TRY(stage_in(c, SLOT_IN_ENTRIES, entry_points, n_entries * sizeof(uint64_t),
                      in_dev && !(flags & ORCD_ENTRIES_HOST), &d_entries));
    Rlev2Table t;
    ORCD_TRY(rlev2_table(c, (const uint8_t *)d_stream, len, num_values, (const uint64_t *)d_entries,
                         n_entries, (flags & ORCD_SIGNED) ? 1 : 0, &t, true));
    std::vector<uint64_t> h((size_t)t.n_runs + 1);
    hipError_t e = hipMemcpyAsync(h.data(), t.runs, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost,
                                  c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        c->last_hip = e;
        return ORCD_HIP_ERROR;
    }
    size_t count = 0;
    uint64_t out_off = 0;  // the table holds descriptors: a run starts where the lengths before it end
    for (uint64_t i = 0; i < t.n_runs && out_off < num_values; i++, count++) {
        if (count < max_runs) {
            runs[count].payload_offset = desc_payload(h[i]);
            runs[count].out_offset = out_off;
            runs[count].kind = desc_kind(h[i]);
            runs[count].bit_width = desc_width(h[i]);
            runs[count].length = desc_len(h[i]);
            runs[count].reserved = 0;
        }
        out_off += desc_len(h[i]);
    }
    *n_runs = count;
    return ORCD_OK;
} ORCD_CATCH

// ---- byte-RLE and booleans ------------------------------------------------------------------

int orcd_byte_rle_runs(orcd_ctx *c, const uint8_t *stream, size_t len, size_t num_bytes,
                       const uint64_t *entry_points, size_t n_entries, orcd_byte_run *runs,
                       size_t max_runs, size_t *n_runs, uint64_t *total_bytes, unsigned flags) try {
    if (!c || (len && !stream) || !n_runs || (max_runs && !runs) || (n_entries && !entry_points))
        return ORCD_INVALID_ARGUMENT;
    *n_runs = 0;
    if (total_bytes) *total_bytes = 0;
    if (num_bytes == 0 || len == 0) return num_bytes == 0 || num_bytes == ORCD_WHOLE_STREAM ? ORCD_OK : ORCD_OUT_OF_SPEC;
    DeviceGuard g(c->device);
    reset_events(c);
    const bool in_dev = flags & ORCD_IN_DEVICE;
    mark(c, 0);
    const void *d_stream, *d_entries;
    ORCD_TRY(stage_in(c, SLOT_IN_A, stream, len, in_dev, &d_stream));
    ORCD_TRY(stage_in(c, SLOT_IN_ENTRIES, entry_points, n_entries * sizeof(uint64_t),
                      in_dev && !(flags & ORCD_ENTRIES_HOST), &d_entries));
    ByteRleTable t;
    uint64_t held = 0;
    ORCD_TRY(byte_rle_table(c, (const uint8_t *)d_stream, len, num_bytes, (const uint64_t *)d_entries,
                            n_entries, &t, nullptr, nullptr, num_bytes == ORCD_WHOLE_STREAM, &held));
    std::vector<ByteRun> h((size_t)t.n_runs);
    if (!h.empty()) {
        hipError_t e = hipMemcpyAsync(h.data(), t.runs, h.size() * sizeof(ByteRun), hipMemcpyDeviceToHost,
                                      c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            c->last_hip = e;
            return ORCD_HIP_ERROR;
        }
    }
    size_t count = 0;
    uint64_t covered = 0;
    for (size_t i = 0; i < h.size() && h[i].out_off < num_bytes; i++, count++) {
        const uint32_t n = (uint32_t)((h[i].src >> 40) & 0xFF) + 1;
        if (count < max_runs) {
            runs[count].payload_offset = h[i].src & ((1ull << 40) - 1);
            runs[count].out_offset = h[i].out_off;
            runs[count].length = n;
            runs[count].is_run = (h[i].src >> 48) & 1;
        }
        covered = h[i].out_off + n;
    }
    *n_runs = count;
    if (total_bytes) *total_bytes = covered;
    return ORCD_OK;
} ORCD_CATCH

int orcd_byte_rle_decode(orcd_ctx *c, const uint8_t *stream, size_t len, size_t num_bytes,
                         const uint64_t *entry_points, size_t n_entries, uint8_t *out,
                         unsigned flags) try {
    if (!c || (len && !stream) || (num_bytes && !out) || (n_entries && !entry_points))
        return ORCD_INVALID_ARGUMENT;
    if (num_bytes == 0) return ORCD_OK;
    DeviceGuard g(c->device);
    reset_events(c);
    const bool in_dev = flags & ORCD_IN_DEVICE, out_dev = flags & ORCD_OUT_DEVICE;
    mark(c, 0);
    const void *d_stream, *d_entries;
    ORCD_TRY(stage_in(c, SLOT_IN_A, stream, len, in_dev, &d_stream));
    ORCD_TRY(stage_in(c, SLOT_IN_ENTRIES, entry_points, n_entries * sizeof(uint64_t),
                      in_dev && !(flags & ORCD_ENTRIES_HOST), &d_entries));
    uint8_t *d_out = out;
    if (!out_dev) {
        void *p;
        ORCD_TRY(ensure(c, SLOT_OUT_A, num_bytes, &p));
        d_out = (uint8_t *)p;
    }
    ORCD_TRY(byte_rle_device(c, (const uint8_t *)d_stream, len, num_bytes, (const uint64_t *)d_entries,
                             n_entries, d_out));
    if (!out_dev) ORCD_TRY(copy_out(c, out, d_out, num_bytes));
    return ORCD_OK;
} ORCD_CATCH

int orcd_boolean_decode(orcd_ctx *c, const uint8_t *stream, size_t len, size_t num_bits,
                        const uint64_t *entry_points, size_t n_entries, uint8_t *out_bytes,
                        uint8_t *out_bitmap, uint64_t *set_bits, unsigned flags) try {
    if (!c || (len && !stream) || (n_entries && !entry_points)) return ORCD_INVALID_ARGUMENT;
    if (set_bits) *set_bits = 0;
    if (num_bits == 0) return ORCD_OK;
    DeviceGuard g(c->device);
    reset_events(c);
    const bool in_dev = flags & ORCD_IN_DEVICE, out_dev = flags & ORCD_OUT_DEVICE;
    const size_t n_bytes = (num_bits + 7) / 8;
    mark(c, 0);
    const void *d_stream, *d_entries;
    ORCD_TRY(stage_in(c, SLOT_IN_A, stream, len, in_dev, &d_stream));
    ORCD_TRY(stage_in(c, SLOT_IN_ENTRIES, entry_points, n_entries * sizeof(uint64_t),
                      in_dev && !(flags & ORCD_ENTRIES_HOST), &d_entries));
    void *flagbuf;
    ORCD_TRY(ensure(c, SLOT_FLAGS, BOOL_COUNT_BYTES, &flagbuf));
    ByteRleTable t;
    ORCD_TRY(byte_rle_table(c, (const uint8_t *)d_stream, len, n_bytes, (const uint64_t *)d_entries,
                            n_entries, &t));
    uint8_t *d_bytes = out_bytes, *d_bitmap = out_bitmap;
    if (!out_dev) {
        void *p;
        if (out_bytes) {
            ORCD_TRY(ensure(c, SLOT_OUT_A, num_bits, &p));
            d_bytes = (uint8_t *)p;
        }
        if (out_bitmap) {
            ORCD_TRY(ensure(c, SLOT_OUT_B, n_bytes, &p));
            d_bitmap = (uint8_t *)p;
        }
    }
    uint64_t *d_count = set_bits ? (uint64_t *)flagbuf : nullptr;
    // the run table expands straight into values: no packed intermediate
    if (d_bytes || d_bitmap || d_count)
        HIP_TRY(c, launch_bool_expand(c->stream, (const uint8_t *)d_stream, len, t.runs, t.win_first,
                                      t.n_win_slots, t.n_runs, num_bits, d_bytes, d_bitmap, d_count));
    mark(c, 3);
    if (d_count)
        HIP_TRY(c, hipMemcpyAsync(c->h_scalars, d_count, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    if (!out_dev) {
        if (out_bytes) ORCD_TRY(result_to_host(c, out_bytes, d_bytes, num_bits));
        if (out_bitmap) ORCD_TRY(result_to_host(c, out_bitmap, d_bitmap, n_bytes));
    }
    if (d_count || !out_dev) HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (set_bits) *set_bits = c->h_scalars[0];
    return ORCD_OK;
} ORCD_CATCH

// ---- floats ---------------------------------------------------------------------------------

int orcd_float_decode(orcd_ctx *c, const uint8_t *stream, size_t len, size_t elem_size,
                      size_t num_values, void *out, unsigned flags) try {
    if (!c || (len && !stream) || (num_values && !out)) return ORCD_INVALID_ARGUMENT;
    if (elem_size != 4 && elem_size != 8) return ORCD_INVALID_ARGUMENT;
    if (num_values > len / elem_size) return ORCD_DECODE_FLOAT;  // float.rs:78-81
    if (num_values == 0) return ORCD_OK;
    DeviceGuard g(c->device);
    reset_events(c);
    const bool in_dev = flags & ORCD_IN_DEVICE, out_dev = flags & ORCD_OUT_DEVICE;
    const size_t bytes = num_values * elem_size;
    mark(c, 0);
    // little-endian IEEE in the file, little-endian IEEE on the GPU: a bit-for-bit copy
    hipMemcpyKind kind = in_dev ? (out_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost)
                                : (out_dev ? hipMemcpyHostToDevice : hipMemcpyHostToHost);
    HIP_TRY(c, hipMemcpyAsync(out, stream, bytes, kind, c->stream));
    mark(c, 3);
    if (!out_dev) HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ORCD_OK;
} ORCD_CATCH

// ---- strings --------------------------------------------------------------------------------

int orcd_string_direct_decode(orcd_ctx *c, const uint8_t *length_stream, size_t length_len,
                              const uint64_t *length_entry_points, size_t n_length_entries,
                              const uint8_t *data, size_t data_len, size_t num_values,
                              int64_t *offsets, uint8_t *bytes_out, unsigned flags) try {
    if (!c || !offsets || (length_len && !length_stream) || (data_len && !data) ||
        (n_length_entries && !length_entry_points))
        return ORCD_INVALID_ARGUMENT;
    if (data_len >= (1ull << 52)) return ORCD_INVALID_ARGUMENT;
    DeviceGuard g(c->device);
    reset_events(c);
    const bool in_dev = flags & ORCD_IN_DEVICE, out_dev = flags & ORCD_OUT_DEVICE;
    mark(c, 0);
    int64_t *d_offsets = offsets;
    if (!out_dev) {
        void *p;
        ORCD_TRY(ensure(c, SLOT_OUT_A, (num_values + 1) * sizeof(int64_t), &p));
        d_offsets = (int64_t *)p;
    }
    uint64_t total = 0;
    if (num_values == 0) {
        HIP_TRY(c, hipMemsetAsync(d_offsets, 0, sizeof(int64_t), c->stream));
    } else {
        const void *d_len_stream, *d_entries;
        ORCD_TRY(stage_in(c, SLOT_IN_A, length_stream, length_len, in_dev, &d_len_stream));
        ORCD_TRY(stage_in(c, SLOT_IN_ENTRIES, length_entry_points, n_length_entries * sizeof(uint64_t),
                          in_dev && !(flags & ORCD_ENTRIES_HOST), &d_entries));
        const uint8_t *ls = (const uint8_t *)d_len_stream;
        const bool v1 = (flags & ORCD_LENGTH_RLE_V1) != 0;
        Rlev2Table t;
        void *gsums = nullptr, *sums, *flag;
        ORCD_TRY(ensure(c, SLOT_FLAGS, 64, &flag));
        uint32_t bad = 0;
        if (v1) {
            // a file written as ORC 0.11: the lengths are decoded as they are and scanned in place
            ORCD_TRY(rlev1_device(c, ls, length_len, num_values, (const uint64_t *)d_entries, n_length_entries, 0,
                                  d_offsets));
            ORCD_TRY(ensure(c, SLOT_SCAN_SUMS, scan_block_sums_len(num_values) * sizeof(uint64_t), &sums));
        } else {
        ORCD_TRY(rlev2_table(c, ls, length_len, num_values, (const uint64_t *)d_entries,
                             n_length_entries, 0, &t));
        // Offsets are the exclusive prefix sums of the lengths (variable_length.rs:15-23: each
        // next(len) continues where the last one ended).  The lengths themselves are never
        // stored: one pass over the LENGTH stream reduces every group of runs to its sum, the
        // few group sums are scanned, and a second pass expands the runs straight into offsets.
        ORCD_TRY(ensure(c, SLOT_TMP, ((size_t)t.n_groups + 1) * sizeof(uint64_t), &gsums));
        ORCD_TRY(ensure(c, SLOT_SCAN_SUMS, scan_block_sums_len(std::max<uint64_t>(num_values, t.n_groups)) * sizeof(uint64_t), &sums));
        HIP_TRY(c, hipMemsetAsync(flag, 0, sizeof(uint32_t), c->stream));
        HIP_TRY(c, launch_rlev2_group_sums(c->stream, t, (uint64_t *)gsums, (uint32_t *)flag));
        HIP_TRY(c, launch_exclusive_scan_u64(c->stream, (uint64_t *)gsums, t.n_groups, (uint64_t *)sums,
                                             data_len, (uint32_t *)flag));
        HIP_TRY(c, hipMemcpyAsync(c->h_scalars, (uint64_t *)gsums + t.n_groups, sizeof(uint64_t),
                                  hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->h_scalars + 1, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        bad = *(uint32_t *)(c->h_scalars + 1);
        total = c->h_scalars[0];
        }
        if (v1 || (bad & 4u)) {
            // PatchedBase runs in the LENGTH stream: their patches are applied after the fact,
            // so the lengths are materialised and scanned in place
            if (!v1) HIP_TRY(c, launch_rlev2_values(c->stream, t, d_offsets));
            HIP_TRY(c, hipMemsetAsync(flag, 0, sizeof(uint32_t), c->stream));
            HIP_TRY(c, launch_exclusive_scan_u64(c->stream, (uint64_t *)d_offsets, num_values,
                                                 (uint64_t *)sums, data_len, (uint32_t *)flag));
            HIP_TRY(c, hipMemcpyAsync(c->h_scalars, d_offsets + num_values, sizeof(uint64_t),
                                      hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(c->h_scalars + 1, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            bad = *(uint32_t *)(c->h_scalars + 1) & 1u;
            total = c->h_scalars[0];
            // Values::next would run out of bytes: one length alone, or their sum
            if (bad || total > data_len) return ORCD_OUT_OF_SPEC;
        } else {
            if ((bad & 1u) || total > data_len) return ORCD_OUT_OF_SPEC;
            HIP_TRY(c, launch_rlev2_offsets(c->stream, t, (const uint64_t *)gsums, d_offsets));
            HIP_TRY(c, hipMemcpyAsync(d_offsets + num_values, (uint64_t *)gsums + t.n_groups,
                                      sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
        }
    }
    mark(c, 3);
    const uint8_t *d_data = data;
    uint8_t *d_bytes = bytes_out;
    const bool want_utf8 = (flags & ORCD_CHECK_UTF8) && total;
    if (total && (bytes_out || want_utf8)) {
        if (!in_dev) {
            const void *p;
            ORCD_TRY(stage_in(c, SLOT_IN_B, data, total, false, &p));
            d_data = (const uint8_t *)p;
        }
        if (bytes_out && !out_dev) {
            d_bytes = const_cast<uint8_t *>(d_data);  // the staged copy is the result
            if (in_dev) {
                void *p;
                ORCD_TRY(ensure(c, SLOT_OUT_B, total, &p));
                d_bytes = (uint8_t *)p;
            }
        }
        void *flag = nullptr;
        if (want_utf8) {
            ORCD_TRY(ensure(c, SLOT_FLAGS, 64, &flag));
            HIP_TRY(c, hipMemsetAsync(flag, 0, sizeof(uint32_t), c->stream));
        }
        HIP_TRY(c, launch_string_bytes(c->stream, d_data, bytes_out ? d_bytes : nullptr, d_offsets,
                                       num_values, total, (uint32_t *)flag));
        mark(c, 4);
        if (want_utf8) {
            HIP_TRY(c, hipMemcpyAsync(c->h_scalars, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (*(uint32_t *)c->h_scalars) return ORCD_INVALID_UTF8;
        }
    }
    if (!out_dev) {
        ORCD_TRY(result_to_host(c, offsets, d_offsets, (num_values + 1) * sizeof(int64_t)));
        if (bytes_out && total)
            ORCD_TRY(result_to_host(c, bytes_out, d_bytes, total));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return ORCD_OK;
} ORCD_CATCH

// First phase of a dictionary decode, all on the device: the bytes every chunk of 64 values
// holds, scanned.  *chunk_prefix (in a buffer of the context, valid until its next call) gets
// n_chunks + 1 entries, the last one the total.  num_values > 0.  Only launches: what the phase
// found is for dictionary_sums_verdict to fetch.
// max_len: null, or receives the longest entry (which picks the gather's kernel) with the verdict.
static int dictionary_sums_launch(orcd_ctx *c, const int64_t *d_dict_off, size_t dict_size, size_t dict_total,
                                  const int64_t *d_idx, size_t num_values, const uint64_t **chunk_prefix,
                                  bool want_max_len) {
    const uint64_t n_chunks = ((uint64_t)num_values + 63) / 64;
    void *flag, *sums, *chunk_sums;
    ORCD_TRY(ensure(c, SLOT_FLAGS, 64, &flag));
    ORCD_TRY(ensure(c, SLOT_TMP, (n_chunks + 1) * sizeof(uint64_t), &chunk_sums));
    ORCD_TRY(ensure(c, SLOT_SCAN_SUMS, scan_block_sums_len(n_chunks) * sizeof(uint64_t), &sums));
    HIP_TRY(c, hipMemsetAsync(flag, 0, 2 * sizeof(uint32_t), c->stream));
    if (want_max_len)  // (the second word)
        HIP_TRY(c, launch_dictionary_max_len(c->stream, d_dict_off, dict_size, (uint32_t *)flag + 1));
    HIP_TRY(c, launch_dictionary_chunk_sums(c->stream, d_dict_off, dict_size, dict_total, d_idx, num_values,
                                            (uint64_t *)chunk_sums, (uint32_t *)flag));
    // (a chunk holds at most 64 * dict_total bytes; the limit only has to stay below 2^52)
    const uint64_t chunk_limit = std::min<uint64_t>((uint64_t)dict_total * 64, (1ull << 52) - 1);
    HIP_TRY(c, launch_exclusive_scan_u64(c->stream, (uint64_t *)chunk_sums, n_chunks, (uint64_t *)sums,
                                         chunk_limit, (uint32_t *)flag));
    *chunk_prefix = (const uint64_t *)chunk_sums;
    return ORCD_OK;
}

// What dictionary_sums_launch found (waits for the stream): the column's bytes in *total, or the error.
static int dictionary_sums_verdict(orcd_ctx *c, size_t dict_total, size_t num_values, const uint64_t *chunk_prefix,
                                   uint64_t *total, uint32_t *max_len) {
    const uint64_t n_chunks = ((uint64_t)num_values + 63) / 64;
    void *flag;
    ORCD_TRY(ensure(c, SLOT_FLAGS, 64, &flag));
    HIP_TRY(c, hipMemcpyAsync(c->h_scalars, chunk_prefix + n_chunks, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->h_scalars + 1, flag, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const uint32_t bad = *(uint32_t *)(c->h_scalars + 1);
    if (max_len) *max_len = ((uint32_t *)(c->h_scalars + 1))[1];
    if (bad & 2u) return ORCD_INVALID_ARGUMENT;  // dict_offsets do not describe dict_bytes
    if (bad & 1u) return dict_total >= (1ull << 46) ? ORCD_INVALID_ARGUMENT : ORCD_OUT_OF_SPEC;
    *total = c->h_scalars[0];
    if (*total == ~0ull) return ORCD_INVALID_ARGUMENT;  // the column would exceed 2^64 bytes
    return ORCD_OK;
}

static int dictionary_sums(orcd_ctx *c, const int64_t *d_dict_off, size_t dict_size, size_t dict_total,
                           const int64_t *d_idx, size_t num_values, const uint64_t **chunk_prefix,
                           uint64_t *total, uint32_t *max_len) {
    ORCD_TRY(dictionary_sums_launch(c, d_dict_off, dict_size, dict_total, d_idx, num_values, chunk_prefix, true));
    return dictionary_sums_verdict(c, dict_total, num_values, *chunk_prefix, total, max_len);
}

int orcd_string_dictionary_decode(orcd_ctx *c, const int64_t *dict_offsets,
                                  const uint8_t *dict_bytes, size_t dict_size,
                                  const int64_t *indices, size_t num_values, int64_t *offsets_out,
                                  uint8_t *bytes_out, size_t bytes_capacity, unsigned flags) try {
    if (!c || !offsets_out || !dict_offsets || (num_values && !indices)) return ORCD_INVALID_ARGUMENT;
    DeviceGuard g(c->device);
    reset_events(c);
    const bool in_dev = flags & ORCD_IN_DEVICE, out_dev = flags & ORCD_OUT_DEVICE;
    mark(c, 0);
    size_t dict_total = 0;
    const void *d_dict_off, *d_dict_bytes, *d_idx;
    ORCD_TRY(stage_in(c, SLOT_IN_C, dict_offsets, (dict_size + 1) * sizeof(int64_t), in_dev, &d_dict_off));
    // the dictionary's size, and its longest entry (which picks the gather's kernel)
    uint32_t max_len = 0;
    if (in_dev) {
        void *flag;
        ORCD_TRY(ensure(c, SLOT_FLAGS, 64, &flag));
        HIP_TRY(c, hipMemsetAsync(flag, 0, 2 * sizeof(uint32_t), c->stream));
        HIP_TRY(c, launch_dictionary_max_len(c->stream, (const int64_t *)d_dict_off, dict_size, (uint32_t *)flag + 1));
        HIP_TRY(c, hipMemcpyAsync(c->h_scalars, dict_offsets + dict_size, sizeof(int64_t),
                                  hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->h_scalars + 1, flag, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        dict_total = (size_t)c->h_scalars[0];
        max_len = ((uint32_t *)(c->h_scalars + 1))[1];
    } else {
        dict_total = (size_t)dict_offsets[dict_size];
        for (size_t k = 0; k < dict_size; k++) {
            const uint64_t b = (uint64_t)dict_offsets[k], e = (uint64_t)dict_offsets[k + 1];
            if (e > b) max_len = (uint32_t)std::max<uint64_t>(max_len, std::min<uint64_t>(e - b, 0xFFFFFFFFull));
        }
    }
    if ((dict_total && !dict_bytes) || dict_total >= (1ull << 52)) return ORCD_INVALID_ARGUMENT;
    ORCD_TRY(stage_in(c, SLOT_IN_B, dict_bytes, dict_total, in_dev, &d_dict_bytes));
    ORCD_TRY(stage_in(c, SLOT_IN_D, indices, num_values * sizeof(int64_t), in_dev, &d_idx));
    int64_t *d_offsets = offsets_out;
    if (!out_dev) {
        void *p;
        ORCD_TRY(ensure(c, SLOT_OUT_A, (num_values + 1) * sizeof(int64_t), &p));
        d_offsets = (int64_t *)p;
    }
    if (num_values == 0) {
        HIP_TRY(c, hipMemsetAsync(d_offsets, 0, sizeof(int64_t), c->stream));
        if (!out_dev) ORCD_TRY(copy_out(c, offsets_out, d_offsets, sizeof(int64_t)));
        return ORCD_OK;
    }
    uint64_t total = 0;
    const uint64_t *chunk_prefix = nullptr;
    ORCD_TRY(dictionary_sums_launch(c, (const int64_t *)d_dict_off, dict_size, dict_total, (const int64_t *)d_idx,
                                    num_values, &chunk_prefix, false));
    uint8_t *d_bytes = bytes_out;
    if (out_dev) {
        // The gather follows at once -- the device does not sit idle while the host looks at the
        // first phase's verdict -- and writes no byte past the capacity, whatever the total is.
        HIP_TRY(c, launch_dictionary_gather(c->stream, (const int64_t *)d_dict_off, (const uint8_t *)d_dict_bytes,
                                            dict_size, dict_total, (const int64_t *)d_idx, num_values,
                                            chunk_prefix, d_offsets, d_bytes, max_len, bytes_capacity));
        ORCD_TRY(dictionary_sums_verdict(c, dict_total, num_values, chunk_prefix, &total, nullptr));
        if (bytes_out && total > bytes_capacity) return ORCD_INVALID_ARGUMENT;
        mark(c, 3);
        return ORCD_OK;
    }
    ORCD_TRY(dictionary_sums_verdict(c, dict_total, num_values, chunk_prefix, &total, nullptr));
    if (bytes_out) {
        if (total > bytes_capacity) return ORCD_INVALID_ARGUMENT;
        void *p;
        ORCD_TRY(ensure(c, SLOT_OUT_B, total, &p));
        d_bytes = (uint8_t *)p;
    }
    HIP_TRY(c, launch_dictionary_gather(c->stream, (const int64_t *)d_dict_off, (const uint8_t *)d_dict_bytes,
                                        dict_size, dict_total, (const int64_t *)d_idx, num_values,
                                        chunk_prefix, d_offsets, d_bytes, max_len));
    mark(c, 3);
    if (!out_dev) {
        ORCD_TRY(result_to_host(c, offsets_out, d_offsets, (num_values + 1) * sizeof(int64_t)));
        if (bytes_out && total)
            ORCD_TRY(result_to_host(c, bytes_out, d_bytes, total));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return ORCD_OK;
} ORCD_CATCH

// ---- null scatter ---------------------------------------------------------------------------

namespace {

// Device-side scatter; validity, dense and out are device pointers.
int scatter_device(orcd_ctx *c, const uint8_t *d_validity, size_t num_rows, const void *d_dense,
                   size_t num_values, size_t elem_size, void *d_out, bool carry) {
    const size_t tiles = scatter_tiles(num_rows);
    void *counts, *sums;
    ORCD_TRY(ensure(c, SLOT_TMP, (tiles + 1) * sizeof(uint64_t), &counts));
    ORCD_TRY(ensure(c, SLOT_SCAN_SUMS, scan_block_sums_len(tiles) * sizeof(uint64_t), &sums));
    void *flag;
    ORCD_TRY(ensure(c, SLOT_FLAGS, 64, &flag));
    HIP_TRY(c, hipMemsetAsync(flag, 0, sizeof(uint32_t), c->stream));
    HIP_TRY(c, launch_valid_count(c->stream, d_validity, num_rows, (uint64_t *)counts));
    HIP_TRY(c, launch_exclusive_scan_u64(c->stream, (uint64_t *)counts, tiles, (uint64_t *)sums,
                                         SC_MAX_TILE_COUNT, (uint32_t *)flag));
    HIP_TRY(c, hipMemcpyAsync(c->h_scalars, (uint64_t *)counts + tiles, sizeof(uint64_t),
                              hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->h_scalars[0] != num_values) return ORCD_INVALID_ARGUMENT;
    HIP_TRY(c, launch_scatter_valid(c->stream, d_validity, num_rows, (const uint64_t *)counts, d_dense,
                                    num_values + (carry ? 1 : 0), elem_size, d_out, carry));
    return ORCD_OK;
}

}  // namespace

int orcd_scatter_valid(orcd_ctx *c, const uint8_t *validity, size_t num_rows, const void *dense,
                       size_t num_values, size_t elem_size, void *out, unsigned flags) try {
    if (!c || (num_rows && (!validity || !out)) || (num_values && !dense)) return ORCD_INVALID_ARGUMENT;
    if (elem_size != 1 && elem_size != 4 && elem_size != 8) return ORCD_INVALID_ARGUMENT;
    const bool carry = flags & ORCD_SCATTER_CARRY;
    if (carry && !dense) return ORCD_INVALID_ARGUMENT;
    if (num_rows == 0) return num_values == 0 ? ORCD_OK : ORCD_INVALID_ARGUMENT;
    DeviceGuard g(c->device);
    reset_events(c);
    const bool in_dev = flags & ORCD_IN_DEVICE, out_dev = flags & ORCD_OUT_DEVICE;
    mark(c, 0);
    const void *d_validity, *d_dense;
    ORCD_TRY(stage_in(c, SLOT_IN_A, validity, num_rows, in_dev, &d_validity));
    ORCD_TRY(stage_in(c, SLOT_IN_B, dense, (num_values + (carry ? 1 : 0)) * elem_size, in_dev, &d_dense));
    void *d_out = out;
    if (!out_dev) ORCD_TRY(ensure(c, SLOT_OUT_A, num_rows * elem_size, &d_out));
    ORCD_TRY(scatter_device(c, (const uint8_t *)d_validity, num_rows, d_dense, num_values, elem_size,
                            d_out, carry));
    mark(c, 3);
    if (!out_dev) ORCD_TRY(copy_out(c, out, d_out, num_rows * elem_size));
    return ORCD_OK;
} ORCD_CATCH

// ---- whole columns --------------------------------------------------------------------------

namespace {

enum TypeKind {  // proto Type.Kind (src/proto.rs:181-203)
    T_BOOLEAN = 0, T_BYTE = 1, T_SHORT = 2, T_INT = 3, T_LONG = 4, T_FLOAT = 5, T_DOUBLE = 6,
    T_STRING = 7, T_BINARY = 8, T_TIMESTAMP = 9, T_LIST = 10, T_MAP = 11, T_STRUCT = 12,
    T_UNION = 13, T_DECIMAL = 14, T_DATE = 15, T_VARCHAR = 16, T_CHAR = 17
};

// The bytes [lo, hi) of a stripe (offsets from the stripe's start) held on the device.
struct StripeBytes {
    const uint8_t *d_base = nullptr;
    uint64_t lo = 0, hi = 0;
};

struct ColumnStreams {
    orcd_file *file;
    orcd_column *col;
    const StripeBytes *dev;  // the device copy the column's streams lie in
    // a compressed file: the column's streams lie decompressed on the device, stream k at
    // plain_dev + the offset the column handle keeps for it
    const uint8_t *plain_dev = nullptr;
    // flags for the stream decoders: streams on the device, ROW_INDEX positions on the host,
    // results stay on the device
    unsigned io() const { return ORCD_OUT_DEVICE | ORCD_IN_DEVICE | ORCD_ENTRIES_HOST; }
    // a stream's bytes (empty when absent) and its ROW_INDEX entry points
    int get(int kind, const uint8_t **bytes, size_t *len, const uint64_t **entries, size_t *n_entries,
            bool required) const {
        *bytes = nullptr;
        *len = 0;
        *entries = nullptr;
        *n_entries = 0;
        if (plain_dev) {
            uint64_t off = 0, n = 0;
            const int rc = orcd_internal_column_plain_range(col, kind, &off, &n);
            if (rc == ORCD_INVALID_KIND && !required) return ORCD_OK;
            if (rc != ORCD_OK) return rc;
            *bytes = plain_dev + off;
            *len = (size_t)n;
            return orcd_column_entry_points(file, col, kind, entries, n_entries);
        }
        uint64_t off = 0, n = 0;
        const int rc = orcd_column_stream_range(col, kind, &off, &n);
        if (rc == ORCD_INVALID_KIND && !required) return ORCD_OK;
        if (rc != ORCD_OK) return rc;
        if (off < dev->lo || off > dev->hi || n > dev->hi - off) return ORCD_OUT_OF_SPEC;
        *bytes = dev->d_base + (off - dev->lo);
        *len = (size_t)n;
        return orcd_column_entry_points(file, col, kind, entries, n_entries);
    }
};

// Device memory for a result array or a temporary of the column calls.  hipMalloc and hipFree of
// a block of 10 MB and more cost 20-100 and 150-190 microseconds (hipFree waits for the device,
// too; the runtime's own memory pools give large blocks back just the same), and a column takes
// a handful of each: released blocks are kept by the context and handed out again -- to work that
// is queued on the context's stream after whatever used them before, so no wait is needed.

int dev_alloc(orcd_ctx *c, size_t bytes, void **p) {
    HIP_TRY(c, c->dev_cache.take(bytes, p));
    return ORCD_OK;
}

void dev_free(orcd_ctx *c, void *p) {
    if (!p) return;
    if (!c->dev_cache.give_back(p, c->dev_cache_limit)) (void)hipFree(p);  // not one of ours
}

// ---- stripes read ahead ------------------------------------------------------------------------
// Reading a stripe's bytes and copying them to the device takes longer than decoding them, and a
// context does one thing at a time.  orcd_stripe_prefetch hands the read of a stripe to come to a
// background thread of the context, with pinned chunks, reader threads and a stream of its own;
// orcd_decode_stripe, when it finds the region it needs among the jobs, waits for that job instead
// of reading.  Jobs are served oldest first, at most Prefetcher::MAX_JOBS are kept.

// A pinned block of at least `len` bytes from the prefetcher's pool (a new one when none fits).
void *pinned_take(Prefetcher *pf, size_t len, size_t *cap) {
    {
        std::lock_guard<std::mutex> lock(pf->m);
        for (size_t i = 0; i < pf->pinned.size(); i++)
            if (pf->pinned[i].second >= len) {
                void *p = pf->pinned[i].first;
                *cap = pf->pinned[i].second;
                pf->pinned.erase(pf->pinned.begin() + i);
                return p;
            }
    }
    void *p = nullptr;
    const size_t want = len + len / 4 + 4096;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return nullptr;
    *cap = want;
    return p;
}

void pinned_give_back(Prefetcher *pf, void *p, size_t cap) {
    if (!p) return;
    void *drop = nullptr;
    {
        std::lock_guard<std::mutex> lock(pf->m);
        pf->pinned.push_back({p, cap});
        if (pf->pinned.size() > Prefetcher::MAX_PINNED) {  // the smallest goes
            size_t least = 0;
            for (size_t i = 1; i < pf->pinned.size(); i++)
                if (pf->pinned[i].second < pf->pinned[least].second) least = i;
            drop = pf->pinned[least].first;
            pf->pinned.erase(pf->pinned.begin() + least);
        }
    }
    if (drop) (void)hipHostFree(drop);
}

// file[off, off + len) into dst, by PINNED_READERS threads a slice each (positioned reads: the page
// cache gives one thread a few GB/s).
constexpr int PINNED_READERS = 4;
constexpr uint64_t PINNED_READ_MIN = 4u << 20;  // smaller regions are read the plain way
int read_in_slices(orcd_file *file, uint64_t off, uint64_t len, uint8_t *dst) {
    const uint64_t slice = ((len + PINNED_READERS - 1) / PINNED_READERS + 4095) & ~(uint64_t)4095;
    int status[PINNED_READERS];
    for (int i = 0; i < PINNED_READERS; i++) status[i] = ORCD_OK;
    std::atomic<int> next{0};
    // (this thread and whatever helpers the process's pool can spare take the slices in turn)
    orcd_parallel(PINNED_READERS - 1, [&](int) {
        for (int i; (i = next.fetch_add(1)) < PINNED_READERS;) {
            const uint64_t lo = std::min<uint64_t>(len, (uint64_t)i * slice), hi = std::min<uint64_t>(len, lo + slice);
            if (hi > lo) status[i] = orcd_file_read_at(file, off + lo, hi - lo, dst + lo);
        }
    });
    for (int i = 0; i < PINNED_READERS; i++)
        if (status[i] != ORCD_OK) return status[i];
    return ORCD_OK;
}

void prefetch_serve(orcd_ctx *c) {
    Prefetcher *pf = c->prefetcher;
    (void)hipSetDevice(c->device);
    for (;;) {
        PrefetchJob *job = nullptr;
        {
            std::unique_lock<std::mutex> lock(pf->m);
            pf->wake.wait(lock, [&] {
                if (pf->stop) return true;
                for (PrefetchJob *j : pf->jobs)
                    if (!j->done) return true;
                return false;
            });
            if (pf->stop) return;
            for (PrefetchJob *j : pf->jobs)
                if (!j->done) {
                    job = j;
                    break;
                }
        }
        int rc = ORCD_OK;
        if (job->keep_host) {
            job->host = pinned_take(pf, job->len, &job->host_cap);
            if (!job->host) rc = ORCD_NO_MEMORY;
            if (rc == ORCD_OK) rc = read_in_slices(job->file, job->file_off, job->len, (uint8_t *)job->host);
            if (rc == ORCD_OK && hipMemcpyAsync(job->d_buf, job->host, job->len, hipMemcpyHostToDevice,
                                                pf->stage.stream) != hipSuccess)
                rc = ORCD_HIP_ERROR;
        } else {
            const HostSource region = {job->file, job->file_off, nullptr};
            rc = chunked_to_device(c, pf->stage, region, job->len, (uint8_t *)job->d_buf);
        }
        // the bytes are on the device when the job is reported done: whoever claims it needs no
        // ordering against this thread's stream
        if (hipStreamSynchronize(pf->stage.stream) != hipSuccess && rc == ORCD_OK) rc = ORCD_HIP_ERROR;
        for (int i = 0; i < N_READERS; i++) pf->stage.stage_busy[i] = false;
        {
            std::lock_guard<std::mutex> lock(pf->m);
            job->status = rc;
            job->done = true;
        }
        pf->finished.notify_all();
    }
}

int prefetch_start(orcd_ctx *c) {
    if (c->prefetcher) return ORCD_OK;
    Prefetcher *pf = new Prefetcher();
    if (hipStreamCreateWithFlags(&pf->stage.stream, hipStreamNonBlocking) != hipSuccess) {
        delete pf;
        return ORCD_HIP_ERROR;
    }
    c->prefetcher = pf;
    pf->worker = std::thread([c] { prefetch_serve(c); });
    return ORCD_OK;
}

void prefetch_shutdown(orcd_ctx *c) {
    Prefetcher *pf = c->prefetcher;
    if (!pf) return;
    {
        std::lock_guard<std::mutex> lock(pf->m);
        pf->stop = true;
    }
    pf->wake.notify_all();
    if (pf->worker.joinable()) pf->worker.join();  // (a read that is under way finishes first)
    for (PrefetchJob *j : pf->jobs) {
        dev_free(c, j->d_buf);
        if (j->host) (void)hipHostFree(j->host);
        delete j;
    }
    pf->jobs.clear();
    for (auto &b : pf->pinned) (void)hipHostFree(b.first);
    pf->pinned.clear();
    if (pf->stage.stream) {
        (void)hipStreamSynchronize(pf->stage.stream);
        pf->stage.release();
        (void)hipStreamDestroy(pf->stage.stream);
    }
    delete pf;
    c->prefetcher = nullptr;
}

// The device copy of file[file_off, file_off + len) of `stripe`, if it was asked for ahead of
// time: waits for the read, hands the block over (the caller owns it then) or returns null (never
// asked for, or the read failed: the caller reads for itself and reports what it finds).
// (host, host_cap: where the job's pinned host copy goes -- the caller gives it back with
// pinned_give_back; a caller that does not take it leaves it to the pool)
void *prefetch_claim(orcd_ctx *c, orcd_file *file, size_t stripe, uint64_t file_off, uint64_t len,
                     void **host = nullptr, size_t *host_cap = nullptr) {
    Prefetcher *pf = c->prefetcher;
    if (!pf) return nullptr;
    std::unique_lock<std::mutex> lock(pf->m);
    for (auto it = pf->jobs.begin(); it != pf->jobs.end(); ++it) {
        PrefetchJob *j = *it;
        if (j->file != file || j->stripe != stripe || j->file_off != file_off || j->len != len) continue;
        pf->finished.wait(lock, [&] { return j->done; });
        pf->jobs.erase(it);
        void *buf = j->d_buf;
        const int rc = j->status;
        void *hp = j->host;
        const size_t hc = j->host_cap;
        delete j;
        lock.unlock();
        if (rc != ORCD_OK || !host) {
            pinned_give_back(pf, hp, hc);
        } else {
            *host = hp;
            *host_cap = hc;
        }
        if (rc != ORCD_OK) {
            dev_free(c, buf);
            return nullptr;
        }
        return buf;
    }
    return nullptr;
}

// The byte range of the stripe's data area that the columns' streams span (file offsets), the way
// decode_stripe_impl reads it.  *len = 0: nothing to read.
int stripe_columns_region(orcd_file *file, size_t stripe, const orcd_stripe_footer *footer,
                          const uint32_t *columns, size_t n_columns, uint64_t *file_off, uint64_t *len) {
    orcd_stripe_info si;
    ORCD_TRY(orcd_file_stripe(file, stripe, &si));
    uint64_t lo = ~0ull, hi = 0;
    for (size_t i = 0; i < n_columns; i++) {
        if (columns[i] >= orcd_file_num_types(file)) return ORCD_INVALID_COLUMN;
        orcd_column *col = nullptr;
        ORCD_TRY(orcd_stripe_column_open(file, stripe, footer, columns[i], &col));
        uint64_t off = 0, n = 0;
        const int rc = orcd_column_region(col, &off, &n);
        orcd_column_free(col);
        ORCD_TRY(rc);
        if (off + n < off) return ORCD_OUT_OF_SPEC;
        lo = std::min(lo, off);
        hi = std::max(hi, off + n);
    }
    if (lo > hi) lo = hi = 0;
    if (si.offset + lo < si.offset) return ORCD_OUT_OF_SPEC;
    *file_off = si.offset + lo;
    *len = hi - lo;
    return ORCD_OK;
}

// Host results of the column calls live in pinned memory from a cache of the same kind: a copy
// from the device into it runs at the speed of the link (into freshly malloc'd pageable memory it
// ran at 6-14 GB/s, page faults and the runtime's staging included), and pinning is paid once.
void host_free(orcd_ctx *c, void *p) {
    if (!p) return;
    if (!c->host_cache.give_back(p, c->host_cache_limit)) (void)hipHostFree(p);
}

// Moves a device result array to where the caller wants it: stays (device) or a host copy.
int deliver(orcd_ctx *c, void *d_ptr, size_t bytes, bool out_dev, void **result) {
    if (out_dev) {
        *result = d_ptr;
        return ORCD_OK;
    }
    void *h = nullptr;
    hipError_t e = c->host_cache.take(bytes, &h);
    if (e == hipSuccess && bytes) e = hipMemcpyAsync(h, d_ptr, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(c, d_ptr);
    if (e != hipSuccess) {
        host_free(c, h);
        c->last_hip = e;
        return ORCD_HIP_ERROR;
    }
    *result = h;
    return ORCD_OK;
}

// Device arrays of a column under construction; whatever is still set at scope exit is released.
struct Pending {
    orcd_ctx *c = nullptr;
    void *validity = nullptr, *values = nullptr, *offsets = nullptr, *bytes = nullptr, *tmp = nullptr,
         *tmp2 = nullptr;
    explicit Pending(orcd_ctx *ctx) : c(ctx) {}
    Pending(const Pending &) = delete;
    Pending &operator=(const Pending &) = delete;
    ~Pending() {
        for (void *p : {validity, values, offsets, bytes, tmp, tmp2}) dev_free(c, p);
    }
};

// validity (deserialize_validity, deserialize.rs:13-25): p.validity[rows] on the device and the
// number of non-null rows
int column_validity(orcd_ctx *c, const ColumnStreams &cs, Pending &p, uint64_t *n_values) {
    orcd_column *col = cs.col;
    const uint64_t rows = orcd_column_number_of_rows(col);
    const uint8_t *sb;
    size_t sl;
    const uint64_t *ent;
    size_t n_ent;
    *n_values = rows;
    ORCD_TRY(dev_alloc(c, rows, &p.validity));
    ORCD_TRY(cs.get(ORCD_STREAM_PRESENT, &sb, &sl, &ent, &n_ent, false));
    if (sb) {
        ORCD_TRY(orcd_boolean_decode(c, sb, sl, rows, ent, n_ent, (uint8_t *)p.validity, nullptr, n_values,
                                     cs.io()));
    } else {
        HIP_TRY(c, hipMemsetAsync(p.validity, 1, rows, c->stream));
    }
    return ORCD_OK;
}

bool is_int_kind(int k) { return k == T_SHORT || k == T_INT || k == T_LONG || k == T_DATE; }

// The values of a column whose validity is in p.  `ints_done`: p.values already holds the dense
// values of an integer column (decoded in a batch with other columns).
int column_values(orcd_ctx *c, const ColumnStreams &cs, int type_kind, unsigned flags,
                  Pending &p, uint64_t n_values, bool ints_done, orcd_column_data *out) {
    const bool out_dev = flags & ORCD_OUT_DEVICE, row_aligned = flags & ORCD_ROW_ALIGNED;
    const unsigned io = cs.io();
    orcd_column *col = cs.col;
    const uint64_t rows = orcd_column_number_of_rows(col);
    const uint8_t *sb;
    size_t sl;
    const uint64_t *ent;
    size_t n_ent;
    const uint64_t slots = row_aligned ? rows : n_values;
    size_t value_size = 0;
    uint64_t num_bytes = 0;

    switch (type_kind) {
    case T_SHORT: case T_INT: case T_LONG: case T_DATE: {  // deserialize_int_array, deserialize.rs:48-82
        value_size = 8;
        if (ints_done) break;
        // ColumnEncoding DIRECT is RLE v1 (Hive 0.11 files): its bytes must not be read as RLE v2
        int enc = 0;
        ORCD_TRY(orcd_column_encoding(col, &enc, nullptr));
        if (enc != ORCD_ENC_DIRECT_V2 && enc != ORCD_ENC_DIRECT) return ORCD_UNSUPPORTED;
        ORCD_TRY(dev_alloc(c, n_values * 8, &p.values));
        ORCD_TRY(cs.get(ORCD_STREAM_DATA, &sb, &sl, &ent, &n_ent, n_values != 0));
        if (enc == ORCD_ENC_DIRECT)  // a file written as ORC 0.11: integer RLE version 1
            ORCD_TRY(orcd_rlev1_decode(c, sb, sl, n_values, ent, n_ent, (int64_t *)p.values, ORCD_SIGNED | io));
        else  // (the index's values-to-skip, when they describe runs of 512 values: no header scan then)
            ORCD_TRY(orcd_rlev2_decode_groups(c, sb, sl, n_values, ent, n_ent,
                                              orcd_column_values_per_group(cs.file, col, ORCD_STREAM_DATA),
                                              (int64_t *)p.values, ORCD_SIGNED | io));
        break;
    }
    case T_BOOLEAN: {  // deserialize_bool_array, deserialize.rs:84-104
        value_size = 1;
        ORCD_TRY(dev_alloc(c, n_values, &p.values));
        ORCD_TRY(cs.get(ORCD_STREAM_DATA, &sb, &sl, &ent, &n_ent, n_values != 0));
        ORCD_TRY(orcd_boolean_decode(c, sb, sl, n_values, ent, n_ent, (uint8_t *)p.values, nullptr, nullptr, io));
        break;
    }
    case T_BYTE: {
        value_size = 1;
        ORCD_TRY(dev_alloc(c, n_values, &p.values));
        ORCD_TRY(cs.get(ORCD_STREAM_DATA, &sb, &sl, &ent, &n_ent, n_values != 0));
        ORCD_TRY(orcd_byte_rle_decode(c, sb, sl, n_values, ent, n_ent, (uint8_t *)p.values, io));
        break;
    }
    case T_FLOAT: case T_DOUBLE: {  // deserialize_f32_array, deserialize.rs:27-46
        value_size = type_kind == T_FLOAT ? 4 : 8;
        ORCD_TRY(dev_alloc(c, n_values * value_size, &p.values));
        ORCD_TRY(cs.get(ORCD_STREAM_DATA, &sb, &sl, &ent, &n_ent, n_values != 0));
        ORCD_TRY(orcd_float_decode(c, sb, sl, value_size, n_values, p.values, io));
        break;
    }
    case T_STRING: case T_VARCHAR: case T_CHAR: case T_BINARY: {  // deserialize_str_array, deserialize.rs:203-222
        const unsigned utf8 = (type_kind != T_BINARY) ? (flags & ORCD_CHECK_UTF8) : 0u;
        int enc = 0;
        uint32_t dict_size = 0;
        ORCD_TRY(orcd_column_encoding(col, &enc, &dict_size));
        const uint8_t *lb;
        size_t ll;
        const uint64_t *lent;
        size_t n_lent;
        ORCD_TRY(dev_alloc(c, (n_values + 1) * 8, &p.offsets));
        // (ColumnEncoding DIRECT / DICTIONARY: a file written as ORC 0.11, whose integer streams --
        // LENGTH, the dictionary indices -- are RLE version 1)
        const bool v1 = enc == ORCD_ENC_DIRECT || enc == ORCD_ENC_DICTIONARY;
        const unsigned len_v1 = v1 ? (unsigned)ORCD_LENGTH_RLE_V1 : 0u;
        if (enc == ORCD_ENC_DIRECT_V2 || enc == ORCD_ENC_DIRECT) {  // deserialize_str_array_direct, deserialize.rs:187-201
            ORCD_TRY(cs.get(ORCD_STREAM_LENGTH, &lb, &ll, &lent, &n_lent, n_values != 0));
            ORCD_TRY(cs.get(ORCD_STREAM_DATA, &sb, &sl, &ent, &n_ent, false));
            ORCD_TRY(dev_alloc(c, sl, &p.bytes));
            ORCD_TRY(orcd_string_direct_decode(c, lb, ll, lent, n_lent, sb, sl, n_values,
                                               (int64_t *)p.offsets, (uint8_t *)p.bytes, io | utf8 | len_v1));
        } else if (enc == ORCD_ENC_DICTIONARY_V2 || enc == ORCD_ENC_DICTIONARY) {  // deserialize_str_dict_array, deserialize.rs:142-185
            ORCD_TRY(cs.get(ORCD_STREAM_LENGTH, &lb, &ll, &lent, &n_lent, dict_size != 0));
            ORCD_TRY(cs.get(ORCD_STREAM_DICTIONARY_DATA, &sb, &sl, &ent, &n_ent, false));
            // the dictionary: offsets[dict_size + 1] and its bytes, on the device
            ORCD_TRY(dev_alloc(c, ((size_t)dict_size + 1) * 8, &p.tmp));
            ORCD_TRY(dev_alloc(c, sl, &p.tmp2));
            // (the LENGTH stream of a dictionary has no ROW_INDEX positions of its own)
            ORCD_TRY(orcd_string_direct_decode(c, lb, ll, nullptr, 0, sb, sl, dict_size, (int64_t *)p.tmp,
                                               (uint8_t *)p.tmp2, io | utf8 | len_v1));
            // the indices: one per non-null row
            void *d_idx;
            ORCD_TRY(ensure(c, SLOT_OUT_B, n_values * 8, &d_idx));
            const uint8_t *ib;
            size_t il;
            ORCD_TRY(cs.get(ORCD_STREAM_DATA, &ib, &il, &ent, &n_ent, n_values != 0));
            if (v1) ORCD_TRY(orcd_rlev1_decode(c, ib, il, n_values, ent, n_ent, (int64_t *)d_idx, io));
            else ORCD_TRY(orcd_rlev2_decode(c, ib, il, n_values, ent, n_ent, (int64_t *)d_idx, io));
            // how many bytes the column holds is known after the first phase: the bytes array is
            // allocated at exactly that size (a bound from the longest entry could be absurd)
            if (n_values == 0) {
                HIP_TRY(c, hipMemsetAsync(p.offsets, 0, 8, c->stream));
                ORCD_TRY(dev_alloc(c, 0, &p.bytes));
            } else {
                HIP_TRY(c, hipMemcpyAsync(c->h_scalars, (int64_t *)p.tmp + dict_size, 8, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                const uint64_t dict_total = c->h_scalars[0];
                uint64_t total = 0;
                const uint64_t *chunk_prefix = nullptr;
                uint32_t max_len = 0;
                ORCD_TRY(dictionary_sums(c, (const int64_t *)p.tmp, dict_size, dict_total, (const int64_t *)d_idx,
                                         n_values, &chunk_prefix, &total, &max_len));
                ORCD_TRY(dev_alloc(c, total, &p.bytes));
                HIP_TRY(c, launch_dictionary_gather(c->stream, (const int64_t *)p.tmp, (const uint8_t *)p.tmp2,
                                                    dict_size, dict_total, (const int64_t *)d_idx, n_values,
                                                    chunk_prefix, (int64_t *)p.offsets, (uint8_t *)p.bytes, max_len));
            }
        } else {
            return ORCD_UNSUPPORTED;  // an encoding the format does not define
        }
        HIP_TRY(c, hipMemcpyAsync(c->h_scalars, (int64_t *)p.offsets + n_values, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        num_bytes = c->h_scalars[0];
        break;
    }
    default:
        return ORCD_UNSUPPORTED;
    }

    if (row_aligned && n_values != rows) {
        // dense -> one slot per row
        void *aligned = nullptr;
        if (p.values) {
            ORCD_TRY(dev_alloc(c, rows * value_size, &aligned));
            std::swap(aligned, p.values);  // p.values: the row-aligned array; aligned: the dense one
            dev_free(c, p.tmp);
            p.tmp = aligned;                // released with the other temporaries
            ORCD_TRY(scatter_device(c, (const uint8_t *)p.validity, rows, aligned, n_values, value_size,
                                    p.values, false));
        } else {
            ORCD_TRY(dev_alloc(c, (rows + 1) * 8, &aligned));
            std::swap(aligned, p.offsets);
            dev_free(c, p.tmp);
            p.tmp = aligned;
            ORCD_TRY(scatter_device(c, (const uint8_t *)p.validity, rows, aligned, n_values, 8, p.offsets, true));
            HIP_TRY(c, hipMemcpyAsync((int64_t *)p.offsets + rows, (int64_t *)aligned + n_values, 8,
                                      hipMemcpyDeviceToDevice, c->stream));
        }
    }

    memset(out, 0, sizeof(*out));
    out->type_kind = type_kind;
    out->on_device = out_dev;
    out->row_aligned = row_aligned;
    out->num_rows = rows;
    out->num_values = n_values;
    out->num_slots = slots;
    out->value_size = value_size;
    out->num_bytes = num_bytes;
    out->owner = c;
    // (no wait here: arrays that stay on the device are complete when the stripe call returns --
    // it waits for the stream once, at its end -- and a copy to the host waits for itself; blocks
    // released in between go to work that is queued behind their last use)
    void *r;
    int rc = deliver(c, p.validity, rows, out_dev, &r);
    p.validity = nullptr;
    out->validity = (uint8_t *)r;
    if (rc == ORCD_OK && p.values) {
        rc = deliver(c, p.values, slots * value_size, out_dev, &r);
        p.values = nullptr;
        out->values = r;
    }
    if (rc == ORCD_OK && p.offsets) {
        rc = deliver(c, p.offsets, (slots + 1) * 8, out_dev, &r);
        p.offsets = nullptr;
        out->offsets = (int64_t *)r;
    }
    if (rc == ORCD_OK && p.bytes) {
        rc = deliver(c, p.bytes, num_bytes, out_dev, &r);
        p.bytes = nullptr;
        out->bytes = (uint8_t *)r;
    }
    if (rc != ORCD_OK) orcd_column_data_free(out);
    return rc;
}

// Columns of a stripe under construction.
struct StripeJob {
    orcd_column *col = nullptr;
    int type_kind = 0;
    uint64_t n_values = 0;
    bool have_validity = false, batched = false;
    const uint8_t *plain_dev = nullptr;  // (see ColumnStreams)
    Pending p;
    explicit StripeJob(orcd_ctx *c) : p(c) {}
};

// Streams up to this size share one set of launches -- and one wait for the header totals -- with
// their peers (measured on a 26-column file: the larger the limit, the faster, up to here).
constexpr uint64_t BATCH_STREAM_MAX = 64u << 20;
constexpr uint64_t BATCH_REGULAR_MIN = 1u << 22;  // values from which a column of full runs goes alone

// One batch of streams that lie in the stripe's device copy: slice offsets are relative to the
// first byte any of them uses.
struct StreamBatch {
    std::vector<orcd_rlev2_slice> slices;
    std::vector<size_t> owner;
    uint64_t lo = ~0ull, hi = 0;
    void add(size_t job, const uint8_t *d_bytes, size_t len, uint64_t units, const uint64_t *ent, size_t n_ent) {
        orcd_rlev2_slice s;
        s.offset = (uint64_t)(uintptr_t)d_bytes;  // absolute for now, rebased by finish()
        s.length = len;
        s.num_values = units;
        s.entry_points = ent;
        s.n_entries = n_ent;
        lo = std::min<uint64_t>(lo, s.offset);
        hi = std::max<uint64_t>(hi, s.offset + len);
        slices.push_back(s);
        owner.push_back(job);
    }
    // slices in the order they lie in memory (the batch calls want that), offsets from the first
    const uint8_t *finish() {
        std::vector<size_t> order(slices.size());
        for (size_t k = 0; k < order.size(); k++) order[k] = k;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return slices[a].offset < slices[b].offset; });
        std::vector<orcd_rlev2_slice> s2;
        std::vector<size_t> o2;
        for (size_t k : order) {
            s2.push_back(slices[k]);
            s2.back().offset -= lo;
            o2.push_back(owner[k]);
        }
        slices.swap(s2);
        owner.swap(o2);
        return (const uint8_t *)(uintptr_t)lo;
    }
};

// ---- the columns of a compressed stripe, decompressed, on the device ----------------------------
// Zlib chunks are inflated by the GPU (a workgroup inflates a chunk at a few megabytes a second;
// what pays is hundreds at once) and, the largest of them, by the host's threads at the same time
// (split_for_host); a stripe of only a few chunks, and the other codecs, are decompressed by the
// host's threads alone and copied over.  ORCD_GPU_INFLATE=0 keeps everything on the host,
// ORCD_GPU_INFLATE_MIN_CHUNKS moves the limit.
constexpr size_t GPU_INFLATE_MIN_CHUNKS = 64;
constexpr uint64_t MOVE_PIECE = 16384;  // bytes one workgroup of the move kernel copies

size_t gpu_inflate_min_chunks() {
    if (const char *e = getenv("ORCD_GPU_INFLATE"))
        if (e[0] == '0') return ~(size_t)0;
    if (const char *e = getenv("ORCD_GPU_INFLATE_MIN_CHUNKS"))
        if (*e) return (size_t)strtoull(e, nullptr, 10);
    return GPU_INFLATE_MIN_CHUNKS;
}

// The GPU inflates a chunk in a time that goes with its compressed size, however few chunks there
// are, while the host's threads share theirs out: when the chunks do not fill the chip, the largest
// are better off on the host, at the same time.  to_host[k]: chunk k of `work` is the host's.
// Rates as measured on an MI355X and its host: 0.165 microseconds per compressed byte for a lone
// chunk on the GPU, INFLATE_PLACES chunks at a time; 0.4 GB/s decompressed per host thread.
// ORCD_GPU_INFLATE_SPLIT=0: everything to the GPU; =2: every other chunk to the host (tests).
constexpr double GPU_US_PER_BYTE = 0.165, HOST_US_PER_BYTE = 1.0 / 400.0;
constexpr size_t INFLATE_PLACES = 768;
constexpr uint64_t HOST_SHARE_MAX = 1ull << 30;

std::vector<char> split_for_host(const std::vector<orcd_chunk> &work) {
    std::vector<char> to_host(work.size(), 0);
    int mode = 1;
    if (const char *e = getenv("ORCD_GPU_INFLATE_SPLIT"))
        if (*e) mode = atoi(e);
    if (mode == 0) return to_host;
    if (mode == 2) {
        for (size_t k = 1; k < work.size(); k += 2) to_host[k] = 1;
        return to_host;
    }
    double gpu_rate = GPU_US_PER_BYTE;
    if (const char *e = getenv("ORCD_GPU_INFLATE_US_PER_BYTE"))  // (for measuring where the balance lies)
        if (*e) gpu_rate = atof(e);
    std::vector<size_t> order(work.size());
    for (size_t k = 0; k < order.size(); k++) order[k] = k;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return work[a].src_length > work[b].src_length; });
    int budget = 1;
    (void)orcd_helper_threads(&budget, nullptr, nullptr);
    const unsigned threads = (unsigned)std::min(16, std::max(1, budget));
    double gpu_bytes = 0, host_us = 0;
    for (const orcd_chunk &w : work) gpu_bytes += w.src_length;
    // largest first: the host takes a chunk as long as it still ends before the GPU would
    for (size_t i = 0; i < order.size(); i++) {
        const orcd_chunk &w = work[order[i]];
        const double gpu_us = std::max<double>(w.src_length, gpu_bytes / INFLATE_PLACES) * gpu_rate;
        const double host_next = host_us + (double)w.dst_capacity * HOST_US_PER_BYTE / threads;
        if (host_next >= gpu_us) break;
        to_host[order[i]] = 1;
        host_us = host_next;
        gpu_bytes -= w.src_length;
    }
    return to_host;
}

void add_move(std::vector<InflateMove> &moves, uint64_t from, uint64_t to, uint64_t len) {
    for (uint64_t done = 0; done < len; done += MOVE_PIECE)
        moves.push_back({from + done, to + done, std::min<uint64_t>(MOVE_PIECE, len - done)});
}

int run_moves(orcd_ctx *c, const std::vector<InflateMove> &moves, const uint8_t *from, uint8_t *to) {
    if (moves.empty()) return ORCD_OK;
    if (moves.size() > 0x7FFFFFFFull) return ORCD_INVALID_ARGUMENT;
    const void *d_moves;
    ORCD_TRY(stage_in(c, SLOT_TMP, moves.data(), moves.size() * sizeof(InflateMove), false, &d_moves));
    HIP_TRY(c, launch_inflate_compact(c->stream, from, to, (const InflateMove *)d_moves, (uint32_t)moves.size()));
    return ORCD_OK;
}

// (ahead_dev / ahead_host: the stripe's bytes as orcd_stripe_prefetch left them, on the device and in
// host memory -- the columns' regions point into the latter; null: nothing was read ahead)
int compressed_columns_to_device(orcd_ctx *c, orcd_file *file, std::vector<std::unique_ptr<StripeJob>> &jobs,
                                 size_t n_columns, Pending &stripe_mem, const uint8_t *ahead_dev = nullptr,
                                 const uint8_t *ahead_host = nullptr) {
    struct Raw {
        const uint8_t *bytes;
        size_t len, n_chunks, n_streams;
        const orcd_internal_chunk *chunks;
        uint64_t at;  // where the region lies in the device copy
    };
    std::vector<Raw> raw(n_columns);
    size_t deflated = 0;
    uint64_t raw_total = 0;
    for (size_t i = 0; i < n_columns; i++) {
        Raw &r = raw[i];
        ORCD_TRY(orcd_internal_column_raw((*jobs[i]).col, &r.bytes, &r.len, &r.chunks, &r.n_chunks, &r.n_streams));
        for (size_t j = 0; j < r.n_chunks; j++) deflated += (!r.chunks[j].original && r.chunks[j].len);
        if (ahead_dev) {  // where the region lies in the copy made ahead of time
            r.at = (uint64_t)(r.bytes - ahead_host);
            raw_total = std::max<uint64_t>(raw_total, r.at + r.len);
            continue;
        }
        r.at = raw_total;
        raw_total += (r.len + 255) & ~(uint64_t)255;
    }
    // (a chunk's header has 23 bits for its length, and no writer makes chunks larger than the
    // compression block: a PostScript that claims more than that is left to the host's inflate,
    // whose memory goes with what the chunks really hold)
    const bool sane_block = orcd_internal_file_compression_block(file) >= 1 && orcd_internal_file_compression_block(file) <= (8u << 20);
    if (orcd_file_compression(file) != 1 || deflated < gpu_inflate_min_chunks() || !sane_block) {
        // the host's threads decompress; the plain streams go to the device as they are
        uint64_t total = 0;
        std::vector<uint64_t> at(n_columns);
        for (size_t i = 0; i < n_columns; i++) {
            ORCD_TRY(orcd_internal_column_inflate_host(file, (*jobs[i]).col));
            const uint8_t *hb;
            size_t hl;
            ORCD_TRY(orcd_internal_column_bytes((*jobs[i]).col, &hb, &hl));
            at[i] = total;
            total += (hl + 255) & ~(uint64_t)255;
        }
        ORCD_TRY(dev_alloc(c, std::max<uint64_t>(total, 256), &stripe_mem.tmp));
        for (size_t i = 0; i < n_columns; i++) {
            const uint8_t *hb;
            size_t hl;
            ORCD_TRY(orcd_internal_column_bytes((*jobs[i]).col, &hb, &hl));
            (*jobs[i]).plain_dev = (const uint8_t *)stripe_mem.tmp + at[i];
            if (hl) {
                const HostSource src = {nullptr, 0, hb};
                ORCD_TRY(chunked_to_device(c, src, hl, (uint8_t *)stripe_mem.tmp + at[i]));
            }
        }
        return ORCD_OK;
    }

    // every chunk gets a place of its own: a deflated one the file's compression block size (the
    // most it may decompress to) or what its bytes can grow to at most, a stored one its length; a stream's chunks lie back to back
    uint64_t block = orcd_internal_file_compression_block(file);
    if (block > (1ull << 30)) block = 1ull << 30;  // (whatever the PostScript claims)
    std::vector<orcd_chunk> work;
    std::vector<const uint8_t *> work_host;  // the same chunks in host memory
    std::vector<InflateMove> stored;
    std::vector<std::vector<uint64_t>> place(n_columns);
    uint64_t plain_total = 0;
    for (size_t i = 0; i < n_columns; i++) {
        const Raw &r = raw[i];
        place[i].resize(r.n_chunks);
        uint32_t stream = ~0u;
        for (size_t j = 0; j < r.n_chunks; j++) {
            const orcd_internal_chunk &ch = r.chunks[j];
            if (ch.stream != stream) {
                plain_total = (plain_total + 15) & ~(uint64_t)15;  // streams start 16-byte aligned
                stream = ch.stream;
            }
            place[i][j] = plain_total;
            if (ch.original) {
                add_move(stored, r.at + ch.at, plain_total, ch.len);
                plain_total += ch.len;
            } else if (ch.len) {
                orcd_chunk w;
                w.src_offset = r.at + ch.at;
                w.dst_offset = plain_total;
                w.src_length = ch.len;
                // (deflate grows a stream at most 1032-fold: a few bytes cannot need a whole block,
                // whatever block size the PostScript claims)
                w.dst_capacity = (uint32_t)std::min<uint64_t>(block, (uint64_t)ch.len * 1032 + 1024);
                work.push_back(w);
                work_host.push_back(r.bytes + ch.at);
                plain_total += w.dst_capacity;
            }
        }
    }
    Pending hold(c);  // the compressed bytes: done with when the sizes are back
    const uint8_t *d_raw = ahead_dev;
    if (!d_raw) {
        ORCD_TRY(dev_alloc(c, std::max<uint64_t>(raw_total, 256), &hold.tmp));
        d_raw = (const uint8_t *)hold.tmp;
        for (size_t i = 0; i < n_columns; i++) {
            if (!raw[i].len) continue;
            const HostSource src = {nullptr, 0, raw[i].bytes};
            ORCD_TRY(chunked_to_device(c, src, raw[i].len, (uint8_t *)hold.tmp + raw[i].at));
        }
    }
    ORCD_TRY(dev_alloc(c, std::max<uint64_t>(plain_total, 256), &stripe_mem.tmp));
    ORCD_TRY(run_moves(c, stored, d_raw, (uint8_t *)stripe_mem.tmp));
    std::vector<uint32_t> sizes(work.size());
    {
        // the host's share (see split_for_host) is inflated by its threads while the GPU does its own
        std::vector<char> to_host = split_for_host(work);
        {   // (the host's chunks wait a block apart in host memory: no more than HOST_SHARE_MAX of it)
            size_t allowed = (size_t)(HOST_SHARE_MAX / block);
            for (size_t k = 0; k < to_host.size(); k++)
                if (to_host[k]) {
                    if (allowed) allowed--;
                    else to_host[k] = 0;
                }
        }
        std::vector<orcd_chunk> gpu_work;
        std::vector<size_t> gpu_at, host_at;
        std::vector<orcd_internal_host_chunk> host_work;
        for (size_t k = 0; k < work.size(); k++) {
            if (to_host[k]) {
                host_at.push_back(k);
                host_work.push_back({work_host[k], work[k].src_length, work[k].dst_capacity});
            } else {
                gpu_at.push_back(k);
                gpu_work.push_back(work[k]);
            }
        }
        {   // the GPU starts workgroups in the order of the table: the longest chunks first, so that
            // the last ones to start are the shortest
            std::vector<size_t> order(gpu_work.size());
            for (size_t k = 0; k < order.size(); k++) order[k] = k;
            std::sort(order.begin(), order.end(),
                      [&](size_t a, size_t b) { return gpu_work[a].src_length > gpu_work[b].src_length; });
            std::vector<orcd_chunk> sorted_work(gpu_work.size());
            std::vector<size_t> sorted_at(gpu_at.size());
            for (size_t k = 0; k < order.size(); k++) {
                sorted_work[k] = gpu_work[order[k]];
                sorted_at[k] = gpu_at[order[k]];
            }
            gpu_work.swap(sorted_work);
            gpu_at.swap(sorted_at);
        }
        // (the host's chunks land a block apart in one piece of host memory the context keeps, go to
        // the device in one copy, and are moved to their places there)
        const size_t pitch = (size_t)block;
        const size_t host_bytes = host_work.size() * pitch;
        if (host_bytes > c->inflate_pinned_cap) {
            if (c->inflate_pinned) (void)hipHostFree(c->inflate_pinned);
            c->inflate_pinned = nullptr;
            c->inflate_pinned_cap = 0;
            const size_t want = host_bytes + host_bytes / 4;
            if (hipHostMalloc(&c->inflate_pinned, want, hipHostMallocDefault) == hipSuccess) c->inflate_pinned_cap = want;
            else c->inflate_pinned = nullptr;
        }
        const bool pinned = host_bytes && c->inflate_pinned;
        if (!pinned && c->inflate_scratch.size() < host_bytes) c->inflate_scratch.resize(host_bytes);
        uint8_t *host_out = pinned ? (uint8_t *)c->inflate_pinned : c->inflate_scratch.data();
        std::vector<uint32_t> host_sizes(host_work.size()), gpu_sizes(gpu_work.size());
        int host_rc = ORCD_OK;
        std::thread helper;
        if (!host_work.empty())
            helper = std::thread([&] {
                try {
                    host_rc = orcd_internal_inflate_list(orcd_file_compression(file), host_work.data(), host_work.size(),
                                                         host_out, pitch, host_sizes.data());
                } catch (...) {
                    host_rc = ORCD_NO_MEMORY;
                }
            });
        int gpu_rc;
        try {
            gpu_rc = inflate_device(c, d_raw, raw_total, gpu_work.data(), gpu_work.size(), (uint8_t *)stripe_mem.tmp,
                                    plain_total, gpu_sizes.data());  // (waits for the stream)
        } catch (...) {  // (the helper must be joined whatever happens here)
            gpu_rc = ORCD_NO_MEMORY;
        }
        if (helper.joinable()) helper.join();
        if (gpu_rc != ORCD_OK && gpu_rc != ORCD_DECOMPRESSION) return gpu_rc;
        if (host_rc != ORCD_OK) return host_rc;
        for (size_t j = 0; j < gpu_at.size(); j++) sizes[gpu_at[j]] = gpu_sizes[j];
        for (size_t j = 0; j < host_at.size(); j++) sizes[host_at[j]] = host_sizes[j];
        for (uint32_t s : sizes)
            if (s == 0xFFFFFFFFu) return ORCD_DECOMPRESSION;
        if (!host_at.empty()) {
            Pending staged(c);
            ORCD_TRY(dev_alloc(c, host_bytes, &staged.tmp));
            if (pinned) {
                HIP_TRY(c, hipMemcpyAsync(staged.tmp, host_out, host_bytes, hipMemcpyHostToDevice, c->stream));
            } else {
                const HostSource from = {nullptr, 0, host_out};
                ORCD_TRY(chunked_to_device(c, from, host_bytes, (uint8_t *)staged.tmp));
            }
            std::vector<InflateMove> places;
            for (size_t j = 0; j < host_at.size(); j++)
                add_move(places, j * pitch, work[host_at[j]].dst_offset, host_sizes[j]);
            ORCD_TRY(run_moves(c, places, (const uint8_t *)staged.tmp, (uint8_t *)stripe_mem.tmp));
            HIP_TRY(c, hipStreamSynchronize(c->stream));  // (the staging block and the table of moves go away)
        }
    }

    // where everything came to lie; a chunk that did not fill its place -- other than a stream's
    // last -- leaves a gap, and then the chunks are moved together
    bool gaps = false;
    {
        size_t w = 0;
        for (size_t i = 0; i < n_columns && !gaps; i++) {
            const Raw &r = raw[i];
            for (size_t j = 0; j < r.n_chunks; j++) {
                const orcd_internal_chunk &ch = r.chunks[j];
                if (ch.original || !ch.len) continue;
                const bool last = j + 1 == r.n_chunks || r.chunks[j + 1].stream != ch.stream;
                if (sizes[w] != work[w].dst_capacity && !last) gaps = true;
                w++;
            }
        }
    }
    std::vector<InflateMove> moves;
    uint64_t tight_total = 0;
    size_t w = 0;
    for (size_t i = 0; i < n_columns; i++) {
        const Raw &r = raw[i];
        std::vector<uint64_t> plain_off(r.n_streams, 0), plain_len(r.n_streams, 0), chunk_off(r.n_chunks, 0);
        uint32_t stream = ~0u;
        for (size_t j = 0; j < r.n_chunks; j++) {
            const orcd_internal_chunk &ch = r.chunks[j];
            const uint64_t size = ch.original ? ch.len : (ch.len ? sizes[w++] : 0);
            if (ch.stream != stream) {
                stream = ch.stream;
                tight_total = (tight_total + 15) & ~(uint64_t)15;
                plain_off[stream] = gaps ? tight_total : place[i][j];
            }
            chunk_off[j] = plain_len[stream];
            if (gaps) add_move(moves, place[i][j], tight_total, size);
            plain_len[stream] += size;
            tight_total += size;
        }
        // (a stream without chunks is empty, wherever it lies)
        ORCD_TRY(orcd_internal_column_set_plain((*jobs[i]).col, plain_off.data(), plain_len.data(), chunk_off.data()));
    }
    if (gaps) {
        void *tight = nullptr;
        ORCD_TRY(dev_alloc(c, std::max<uint64_t>(tight_total, 256), &tight));
        hold.tmp2 = stripe_mem.tmp;  // the places: released when the moves are done
        stripe_mem.tmp = tight;
        ORCD_TRY(run_moves(c, moves, (const uint8_t *)hold.tmp2, (uint8_t *)tight));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    for (size_t i = 0; i < n_columns; i++) (*jobs[i]).plain_dev = (const uint8_t *)stripe_mem.tmp;
    return ORCD_OK;
}

int decode_stripe_impl(orcd_ctx *c, orcd_file *file, size_t stripe, const orcd_stripe_footer *footer,
                       const uint32_t *columns, size_t n_columns, unsigned flags, orcd_column_data *out,
                       std::vector<std::unique_ptr<StripeJob>> &jobs, Pending &stripe_mem) {
    // ORCD_TRACE_STRIPE=1: where the call's time went, a line per stripe on stderr
    static const bool trace = [] {
        const char *e = getenv("ORCD_TRACE_STRIPE");
        return e && *e && *e != '0';
    }();
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
    };
    double ms_read = 0, ms_validity = 0, ms_ints = 0;
    // 1. the bytes the columns' streams span, read once and copied once to the device
    orcd_stripe_info si;
    ORCD_TRY(orcd_file_stripe(file, stripe, &si));
    const bool compressed = orcd_file_compression(file) != 0;
    // a compressed stripe that was read ahead of time (orcd_stripe_prefetch): its bytes are in host
    // memory, for the walk over the chunk headers, and on the device, for the inflate
    struct HostAhead {  // (back to the prefetcher's pool when the call is over)
        orcd_ctx *c;
        void *p = nullptr;
        size_t cap = 0;
        ~HostAhead() {
            if (p && c->prefetcher) pinned_give_back(c->prefetcher, p, cap);
            else if (p) (void)hipHostFree(p);
        }
    } ahead_host{c};
    Pending ahead(c);
    uint64_t ahead_at = 0, ahead_len = 0;
    if (compressed && orcd_file_compression(file) == 1 && gpu_inflate_min_chunks() != ~(size_t)0) {
        for (size_t i = 0; i < n_columns; i++)
            if (columns[i] >= orcd_file_num_types(file)) return ORCD_INVALID_COLUMN;
        const bool known = stripe_columns_region(file, stripe, footer, columns, n_columns, &ahead_at, &ahead_len) == ORCD_OK &&
                           ahead_len;
        if (known && c->prefetcher)
            ahead.tmp = prefetch_claim(c, file, stripe, ahead_at, ahead_len, &ahead_host.p, &ahead_host.cap);
        if (ahead.tmp && !ahead_host.p) {  // (not a job of the compressed kind)
            dev_free(c, ahead.tmp);
            ahead.tmp = nullptr;
        }
        // nothing was read ahead: the same now -- the region into a pinned block by several
        // threads, to the device in one piece (when anything of this fails, the columns read
        // for themselves below and say what is wrong)
        if (known && !ahead.tmp && ahead_len >= PINNED_READ_MIN && prefetch_start(c) == ORCD_OK) {
            ahead_host.p = pinned_take(c->prefetcher, ahead_len, &ahead_host.cap);
            bool fine = ahead_host.p && read_in_slices(file, ahead_at, ahead_len, (uint8_t *)ahead_host.p) == ORCD_OK &&
                        dev_alloc(c, ahead_len, &ahead.tmp) == ORCD_OK;
            if (fine && hipMemcpyAsync(ahead.tmp, ahead_host.p, ahead_len, hipMemcpyHostToDevice, c->stream) != hipSuccess)
                fine = false;
            if (!fine) {
                dev_free(c, ahead.tmp);
                ahead.tmp = nullptr;
            }
        }
    }
    for (size_t i = 0; i < n_columns; i++) {
        if (columns[i] >= orcd_file_num_types(file)) return ORCD_INVALID_COLUMN;
        (*jobs[i]).type_kind = orcd_file_type_kind(file, columns[i]);
        // (a compressed file: the host reads each column's region and finds its compression chunks;
        // they are decompressed below, src/read/decompress/mod.rs:60-80, on the device or the host)
        if (compressed)
            ORCD_TRY(orcd_internal_column_raw_from(file, stripe, footer, columns[i], ahead.tmp ? (const uint8_t *)ahead_host.p : nullptr,
                                                   ahead_at, ahead_len, &(*jobs[i]).col));
        else ORCD_TRY(orcd_stripe_column_open(file, stripe, footer, columns[i], &(*jobs[i]).col));
    }
    StripeBytes sb;
    if (compressed)
        ORCD_TRY(compressed_columns_to_device(c, file, jobs, n_columns, stripe_mem, (const uint8_t *)ahead.tmp,
                                              ahead.tmp ? (const uint8_t *)ahead_host.p : nullptr));
    sb.lo = ~0ull;
    for (size_t i = 0; i < n_columns && !compressed; i++) {
        uint64_t off = 0, n = 0;
        ORCD_TRY(orcd_column_region((*jobs[i]).col, &off, &n));
        if (off + n < off) return ORCD_OUT_OF_SPEC;
        sb.lo = std::min(sb.lo, off);
        sb.hi = std::max(sb.hi, off + n);
    }
    if (sb.lo > sb.hi) sb.lo = sb.hi = 0;
    if (!compressed) {
        if (si.offset + sb.lo < si.offset) return ORCD_OUT_OF_SPEC;  // offsets a file cannot have
        {   // the region must lie inside the file before memory of its size is asked for
            uint8_t probe;
            if (sb.hi > sb.lo) ORCD_TRY(orcd_file_read_at(file, si.offset + sb.lo + (sb.hi - sb.lo - 1), 1, &probe));
        }
        // read ahead of time (orcd_stripe_prefetch)?  Then the bytes are on the device already.
        stripe_mem.tmp = prefetch_claim(c, file, stripe, si.offset + sb.lo, sb.hi - sb.lo);
        if (!stripe_mem.tmp) {
            ORCD_TRY(dev_alloc(c, sb.hi - sb.lo, &stripe_mem.tmp));
            const HostSource region = {file, si.offset + sb.lo, nullptr};
            ORCD_TRY(chunked_to_device(c, region, sb.hi - sb.lo, (uint8_t *)stripe_mem.tmp));
        }
        sb.d_base = (const uint8_t *)stripe_mem.tmp;
    }

    c->stripe_feed_ms = (float)since(t_start);
    if (trace) {
        (void)hipStreamSynchronize(c->stream);
        ms_read = since(t_start);
    }
    // 2. validity: the small PRESENT streams in one set of launches
    {
        StreamBatch batch;
        uint64_t arena_bytes = 0;
        for (size_t i = 0; i < n_columns; i++) {
            const ColumnStreams cs{file, (*jobs[i]).col, &sb, (*jobs[i]).plain_dev};
            const uint64_t rows = orcd_column_number_of_rows((*jobs[i]).col);
            const uint8_t *b;
            size_t l, n_ent;
            const uint64_t *ent;
            ORCD_TRY(cs.get(ORCD_STREAM_PRESENT, &b, &l, &ent, &n_ent, false));
            if (!b || rows == 0 || l > BATCH_STREAM_MAX) continue;
            batch.add(i, b, l, rows, ent, n_ent);
            arena_bytes += 8 * ((rows + 7) / 8);
        }
        if (trace) fprintf(stderr, "  validity: streams and entry points gathered at %.3f ms\n", since(t_start) - ms_read);
        if (batch.slices.size() > 1) {
            const uint8_t *d_first = batch.finish();
            ORCD_TRY(dev_alloc(c, arena_bytes, &stripe_mem.tmp2));
            std::vector<uint64_t> counts(batch.slices.size());
            const int rc = orcd_boolean_decode_batch(c, d_first, batch.hi - batch.lo, batch.slices.data(),
                                                     batch.slices.size(), (uint8_t *)stripe_mem.tmp2,
                                                     counts.data(), nullptr, ORCD_IN_DEVICE | ORCD_OUT_DEVICE);
            if (trace) fprintf(stderr, "  validity: batch decoded at %.3f ms\n", since(t_start) - ms_read);
            if (rc == ORCD_OK) {
                uint64_t pos = 0;
                for (size_t k = 0; k < batch.owner.size(); k++) {
                    StripeJob &j = (*jobs[batch.owner[k]]);
                    const uint64_t rows = batch.slices[k].num_values;
                    ORCD_TRY(dev_alloc(c, rows, &j.p.validity));
                    HIP_TRY(c, hipMemcpyAsync(j.p.validity, (uint8_t *)stripe_mem.tmp2 + pos, rows,
                                              hipMemcpyDeviceToDevice, c->stream));
                    j.n_values = counts[k];
                    j.have_validity = true;
                    pos += 8 * ((rows + 7) / 8);
                }
            } else if (rc != ORCD_OUT_OF_SPEC) {
                return rc;
            }  // a stream that is out of spec: the per-column decode below says which
        }
    }
    for (size_t i = 0; i < n_columns; i++) {
        if ((*jobs[i]).have_validity) continue;
        const ColumnStreams cs{file, (*jobs[i]).col, &sb, (*jobs[i]).plain_dev};
        ORCD_TRY(column_validity(c, cs, (*jobs[i]).p, &(*jobs[i]).n_values));
    }
    if (trace) {
        (void)hipStreamSynchronize(c->stream);
        ms_validity = since(t_start) - ms_read;
    }
    // 3. the small DATA streams of the integer columns in one set of launches
    {
        StreamBatch batch;
        uint64_t total = 0;
        for (size_t i = 0; i < n_columns; i++) {
            if (!is_int_kind((*jobs[i]).type_kind) || (*jobs[i]).n_values == 0) continue;
            int enc = 0;
            if (orcd_column_encoding((*jobs[i]).col, &enc, nullptr) != ORCD_OK || enc != ORCD_ENC_DIRECT_V2)
                continue;  // RLE v1 or unknown: reported per column below
            const ColumnStreams cs{file, (*jobs[i]).col, &sb, (*jobs[i]).plain_dev};
            const uint8_t *b;
            size_t l, n_ent;
            const uint64_t *ent;
            if (cs.get(ORCD_STREAM_DATA, &b, &l, &ent, &n_ent, true) != ORCD_OK) continue;  // reported per column below
            if (l > BATCH_STREAM_MAX) continue;
            // a long column of full runs (the index says so) is better off by itself: no header scan
            if ((*jobs[i]).n_values >= BATCH_REGULAR_MIN && orcd_column_values_per_group(file, (*jobs[i]).col, ORCD_STREAM_DATA))
                continue;
            batch.add(i, b, l, (*jobs[i]).n_values, ent, n_ent);
            total += (*jobs[i]).n_values;
        }
        if (batch.slices.size() > 1) {
            const uint8_t *d_first = batch.finish();
            void *arena = nullptr;  // (the validity arena may still feed copies on the stream)
            ORCD_TRY(dev_alloc(c, total * 8, &arena));
            Pending hold(c);
            hold.tmp = arena;
            const int rc = orcd_rlev2_decode_batch(c, d_first, batch.hi - batch.lo, batch.slices.data(),
                                                   batch.slices.size(), (int64_t *)arena, nullptr,
                                                   ORCD_SIGNED | ORCD_IN_DEVICE | ORCD_OUT_DEVICE);
            if (rc == ORCD_OK) {
                uint64_t pos = 0;
                for (size_t k = 0; k < batch.owner.size(); k++) {
                    StripeJob &j = (*jobs[batch.owner[k]]);
                    ORCD_TRY(dev_alloc(c, j.n_values * 8, &j.p.values));
                    HIP_TRY(c, hipMemcpyAsync(j.p.values, (int64_t *)arena + pos, j.n_values * 8,
                                              hipMemcpyDeviceToDevice, c->stream));
                    j.batched = true;
                    pos += j.n_values;
                }
                HIP_TRY(c, hipStreamSynchronize(c->stream));  // `arena` is released at scope exit
            } else if (rc != ORCD_OUT_OF_SPEC) {
                return rc;
            }
        }
    }
    if (trace) {
        (void)hipStreamSynchronize(c->stream);
        ms_ints = since(t_start) - ms_read - ms_validity;
    }
    // 4. everything else, column by column
    for (size_t i = 0; i < n_columns; i++) {
        const ColumnStreams cs{file, (*jobs[i]).col, &sb, (*jobs[i]).plain_dev};
        const double t_col = trace ? since(t_start) : 0;
        ORCD_TRY(column_values(c, cs, (*jobs[i]).type_kind, flags, (*jobs[i]).p, (*jobs[i]).n_values, (*jobs[i]).batched,
                               &out[i]));
        if (trace && n_columns > 1) {
            (void)hipStreamSynchronize(c->stream);
            fprintf(stderr, "  column %u (type %d): %.3f ms\n", columns[i], (*jobs[i]).type_kind, since(t_start) - t_col);
        }
    }
    if (trace) {
        (void)hipStreamSynchronize(c->stream);
        const double total = since(t_start);
        fprintf(stderr, "orcd stripe %zu: %zu columns, read+copy %.3f ms, validity %.3f, integer batch %.3f, columns %.3f, total %.3f\n",
                stripe, n_columns, ms_read, ms_validity, ms_ints, total - ms_read - ms_validity - ms_ints, total);
    }
    return ORCD_OK;
}

}  // namespace

int orcd_decode_stripe(orcd_ctx *c, orcd_file *file, size_t stripe, const orcd_stripe_footer *footer,
                       const uint32_t *columns, size_t n_columns, unsigned flags, orcd_column_data *out) try {
    if (!c || !file || !footer || (n_columns && (!columns || !out))) return ORCD_INVALID_ARGUMENT;
    for (size_t i = 0; i < n_columns; i++) memset(&out[i], 0, sizeof(out[i]));
    DeviceGuard g(c->device);
    std::vector<std::unique_ptr<StripeJob>> jobs;
    for (size_t i = 0; i < n_columns; i++) jobs.emplace_back(new StripeJob(c));
    Pending stripe_mem(c);  // the stripe's bytes on the device, and the validity arena
    const auto t_call = std::chrono::steady_clock::now();
    c->stripe_feed_ms = c->stripe_decode_ms = -1.0f;
    const int rc = decode_stripe_impl(c, file, stripe, footer, columns, n_columns, flags, out, jobs, stripe_mem);
    (void)hipStreamSynchronize(c->stream);  // before the device buffers of this call are released
    if (rc == ORCD_OK && c->stripe_feed_ms >= 0)
        c->stripe_decode_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_call).count() -
                              c->stripe_feed_ms;
    for (auto &j : jobs)
        if (j->col) orcd_column_free(j->col);
    if (rc != ORCD_OK)
        for (size_t i = 0; i < n_columns; i++) orcd_column_data_free(&out[i]);
    return rc;
} ORCD_CATCH

int orcd_decode_column(orcd_ctx *c, orcd_file *file, size_t stripe, const orcd_stripe_footer *footer,
                       uint32_t column, unsigned flags, orcd_column_data *out) {
    if (!out) return ORCD_INVALID_ARGUMENT;
    // one column of a stripe is a stripe of one column: the same read (threaded, through pinned
    // chunks) of the column's region and the same decode from its device copy
    return orcd_decode_stripe(c, file, stripe, footer, &column, 1, flags, out);
}

int orcd_stripe_prefetch(orcd_ctx *c, orcd_file *file, size_t stripe, const orcd_stripe_footer *footer,
                         const uint32_t *columns, size_t n_columns) try {
    if (!c || !file || !footer || (n_columns && !columns)) return ORCD_INVALID_ARGUMENT;
    // (a compressed stripe is read ahead when its chunks may be inflated on the device; the host's
    // inflate reads for itself)
    const bool compressed = orcd_file_compression(file) != 0;
    if (compressed && (orcd_file_compression(file) != 1 || gpu_inflate_min_chunks() == ~(size_t)0)) return ORCD_OK;
    uint64_t file_off = 0, len = 0;
    ORCD_TRY(stripe_columns_region(file, stripe, footer, columns, n_columns, &file_off, &len));
    if (len == 0) return ORCD_OK;
    {   // the region must lie inside the file before memory of its size is asked for
        uint8_t probe;
        ORCD_TRY(orcd_file_read_at(file, file_off + (len - 1), 1, &probe));
    }
    DeviceGuard g(c->device);
    ORCD_TRY(prefetch_start(c));
    Prefetcher *pf = c->prefetcher;
    void *stale = nullptr, *stale_host = nullptr;
    size_t stale_cap = 0;
    {
        std::lock_guard<std::mutex> lock(pf->m);
        for (PrefetchJob *j : pf->jobs)
            if (j->file == file && j->stripe == stripe && j->file_off == file_off && j->len == len) return ORCD_OK;
        if (pf->jobs.size() >= Prefetcher::MAX_JOBS) {
            // full: the oldest stripe, if it was read and never asked for, makes room; otherwise
            // this hint is not taken
            PrefetchJob *oldest = pf->jobs.front();
            if (!oldest->done) return ORCD_OK;
            stale = oldest->d_buf;
            stale_host = oldest->host;
            stale_cap = oldest->host_cap;
            pf->jobs.pop_front();
            delete oldest;
        }
    }
    dev_free(c, stale);
    pinned_give_back(pf, stale_host, stale_cap);
    std::unique_ptr<PrefetchJob> job(new PrefetchJob());
    job->file = file;
    job->stripe = stripe;
    job->file_off = file_off;
    job->len = len;
    job->keep_host = compressed;
    ORCD_TRY(dev_alloc(c, len, &job->d_buf));
    {
        std::lock_guard<std::mutex> lock(pf->m);
        pf->jobs.push_back(job.release());
    }
    pf->wake.notify_all();
    return ORCD_OK;
} ORCD_CATCH

void orcd_column_data_free(orcd_column_data *d) {
    if (!d) return;
    for (void *p : {(void *)d->validity, d->values, (void *)d->offsets, (void *)d->bytes}) {
        if (!p) continue;
        if (!d->owner) continue;  // (every array a column call returns has one)
        if (d->on_device) dev_free(d->owner, p);  // back to the context that made it
        else host_free(d->owner, p);
    }
    memset(d, 0, sizeof(*d));
}

}  // extern "C"

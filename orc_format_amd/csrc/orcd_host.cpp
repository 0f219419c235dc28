// Host side of the decoder: ORC file tail, stripe footers and column regions.
//
// Mirrors read_metadata / read_stripe_footer / read_stripe_column / Column of the orc-format crate
// (src/read/mod.rs:46-159, src/read/column.rs:11-79) with a hand-written reader for the handful of
// protobuf messages involved (src/proto.rs:89-99, 113-164, 166-217, 236-278).  No GPU code here.
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <dlfcn.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <vector>
#include <utility>
#include <zlib.h>

#include "../../include/orcd.h"
#include "orcd_column_internal.h"

// No exception leaves the C ABI: host allocations (std::vector, new) are what throws here.
#define ORCD_CATCH                                     \
    catch (...) {                                      \
        return ORCD_NO_MEMORY;                         \
    }

namespace {

constexpr uint64_t DEFAULT_FOOTER_SIZE = 16 * 1024;  // src/read/mod.rs:22

// Every length and offset in a file is a varint of up to 64 bits that a crafted file may set to
// anything: sums of them are formed with these, and an overflow is a malformed file.
inline bool add_ok(uint64_t a, uint64_t b, uint64_t &out) { return !__builtin_add_overflow(a, b, &out); }
inline bool add3_ok(uint64_t a, uint64_t b, uint64_t c, uint64_t &out) {
    uint64_t t;
    return add_ok(a, b, t) && add_ok(t, c, out);
}

// ---- protobuf wire format --------------------------------------------------------------------

struct Cursor {
    const uint8_t *p;
    const uint8_t *end;
    bool ok = true;

    bool done() const { return !ok || p >= end; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 70; shift += 7) {
            if (p >= end) break;
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7F) << (shift & 63);
            if (b < 0x80) return v;
        }
        ok = false;
        return 0;
    }
    // next field: number, wire type; for length-delimited fields `sub` spans the payload and for
    // varints `value` holds it.  Fixed-width fields are skipped.
    bool next(uint32_t &field, uint32_t &wire, uint64_t &value, Cursor &sub) {
        if (done()) return false;
        const uint64_t key = varint();
        if (!ok) return false;
        field = (uint32_t)(key >> 3);
        wire = (uint32_t)(key & 7);
        value = 0;
        switch (wire) {
        case 0: value = varint(); break;
        case 1: if (end - p < 8) ok = false; else p += 8; break;
        case 5: if (end - p < 4) ok = false; else p += 4; break;
        case 2: {
            const uint64_t n = varint();
            if (!ok || (uint64_t)(end - p) < n) { ok = false; break; }
            sub.p = p;
            sub.end = p + n;
            sub.ok = true;
            p += n;
            break;
        }
        default: ok = false;
        }
        return ok;
    }
};

// repeated uint32/uint64, packed or not
void read_repeated(uint32_t wire, uint64_t value, Cursor sub, std::vector<uint64_t> &out, bool &ok) {
    if (wire == 0) {
        out.push_back(value);
    } else if (wire == 2) {
        while (!sub.done()) out.push_back(sub.varint());
        ok = ok && sub.ok;
    } else {
        ok = false;
    }
}

struct TypeInfo {
    int kind = 0;
    std::vector<uint64_t> subtypes;
};

struct Encoding {
    int kind = 0;
    bool has_dict = false;
    uint32_t dictionary_size = 0;
};

// ---- byte source: a file or a borrowed memory image --------------------------------------------

struct Source {
    FILE *f = nullptr;
    const uint8_t *mem = nullptr;
    uint64_t len = 0;

    bool read_at(uint64_t off, uint64_t n, std::vector<uint8_t> &out) const {
        if (off > len || n > len - off) return false;
        out.resize(n);
        return read_into(off, n, out.data());
    }
    bool read_into(uint64_t off, uint64_t n, uint8_t *dst) const {
        if (off > len || n > len - off) return false;
        if (n == 0) return true;
        if (mem) {
            memcpy(dst, mem + off, n);
            return true;
        }
        // positioned reads on the descriptor: safe from several threads at once
        const int fd = fileno(f);
        while (n) {
            const ssize_t got = pread(fd, dst, n, (off_t)off);
            if (got <= 0) return false;
            dst += got;
            off += (uint64_t)got;
            n -= (uint64_t)got;
        }
        return true;
    }
};

}  // namespace

struct orcd_file {
    Source src;
    int compression = 0;
    uint64_t compression_block = 256 * 1024;  // PostScript.compressionBlockSize: most a chunk decompresses to
    uint64_t number_of_rows = 0;
    uint32_t row_index_stride = 0;
    std::vector<orcd_stripe_info> stripes;
    std::vector<TypeInfo> types;
};

struct orcd_stripe_footer {
    std::vector<orcd_stream_info> streams;
    std::vector<Encoding> columns;
    size_t stripe = 0;
};

struct orcd_column {
    bool has_data = true;       // false: a handle without the column's bytes (orcd_stripe_column_open)
    std::vector<uint8_t> data;  // all non-ROW_INDEX streams of the column, in file order
    // A compressed file (CompressionKind ZLIB): `data` holds the streams decompressed, back to
    // back, stream k at plain_off[k] with plain_len[k] bytes (k counts the column's non-ROW_INDEX
    // streams in file order), and per stream the chunk table of the compressed form -- offset of
    // each chunk header in the compressed stream and of its first byte in the decompressed one --
    // which is what ROW_INDEX positions of such a file refer to.
    bool compressed = false;
    std::vector<uint64_t> plain_off, plain_len;
    std::vector<std::vector<std::pair<uint64_t, uint64_t>>> chunks;
    // In between (the column calls, which may inflate on the device): `data` holds the column's
    // region as it is in the file, `raw_chunks` every compression chunk of it in file order.
    bool raw = false;
    std::vector<orcd_internal_chunk> raw_chunks;
    uint32_t raw_streams = 0;
    // ... or the region lies in memory of the caller's (read ahead of time), not in `data`
    const uint8_t *ext = nullptr;
    size_t ext_len = 0;
    const uint8_t *region() const { return ext ? ext : data.data(); }
    size_t region_len() const { return ext ? ext_len : data.size(); }
    uint32_t column = 0;
    uint64_t number_of_rows = 0;
    size_t stripe = 0;
    const orcd_stripe_footer *footer = nullptr;
    // entry points per stream kind, filled on demand from the ROW_INDEX stream
    bool index_loaded = false;
    int index_status = ORCD_OK;  // what loading the ROW_INDEX entry points came to (kept: asked once)
    std::vector<uint64_t> entries[8];
    // For the integer streams: how many values lie between two positions when entry k is the header
    // of the run that holds the stream's value k * that many -- i.e. when the index holds one
    // position per row group for the stream, none was dropped as a duplicate, and every position's
    // values-to-skip is what runs of 512 values throughout would make it -- else 0.
    uint64_t group_values[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace {

// ORC compression chunks (src/read/decompress/mod.rs:9-17, 60-80): a 3-byte little-endian header,
// bit 0 "the bytes that follow are not compressed", the rest their number; a compressed chunk is
// a raw deflate stream.  Appends the decompressed bytes of src[0, n) to `out`; `chunks` (nullable)
// receives (offset of the chunk header in src, offset of its first byte in out - out's size on
// entry).  A header or chunk that reaches past the end, or bytes that do not inflate: the
// reference panics or fails with Error::Decompression -- ORCD_DECOMPRESSION here.
// ---- codecs beyond zlib ---------------------------------------------------------------------
// The reference stops at zlib (decompress/mod.rs:81 is todo!() for the others); files in the
// field are as often Snappy, LZ4 or Zstandard.  Their chunk framing is the same; the codecs are:
// Snappy, decoded here (the format is a length and a series of literals and back-references);
// LZ4 (raw blocks) and Zstandard (frames) through the system's liblz4 / libzstd when the
// process can load them -- no headers are needed for the two calls used -- else ORCD_UNSUPPORTED.
constexpr int COMP_NONE = 0, COMP_ZLIB = 1, COMP_SNAPPY = 2, COMP_LZO = 3, COMP_LZ4 = 4, COMP_ZSTD = 5;

struct CodecLibs {
    int (*lz4_decompress_safe)(const char *, char *, int, int) = nullptr;
    size_t (*zstd_decompress)(void *, size_t, const void *, size_t) = nullptr;
    unsigned (*zstd_is_error)(size_t) = nullptr;
    CodecLibs() {
        if (void *h = dlopen("liblz4.so.1", RTLD_NOW | RTLD_LOCAL))
            lz4_decompress_safe = (int (*)(const char *, char *, int, int))dlsym(h, "LZ4_decompress_safe");
        if (void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL)) {
            zstd_decompress = (size_t (*)(void *, size_t, const void *, size_t))dlsym(h, "ZSTD_decompress");
            zstd_is_error = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
        }
    }
};
const CodecLibs &codec_libs() {
    static const CodecLibs libs;  // loaded once, on first use
    return libs;
}

bool codec_available(int kind) {
    if (kind == COMP_NONE || kind == COMP_ZLIB || kind == COMP_SNAPPY) return true;
    if (kind == COMP_LZ4) return codec_libs().lz4_decompress_safe != nullptr;
    if (kind == COMP_ZSTD) return codec_libs().zstd_decompress != nullptr && codec_libs().zstd_is_error != nullptr;
    return false;
}

// Snappy, raw format: a varint with the decompressed length, then elements whose tag byte's low
// two bits say literal (00: length in the tag or in 1..4 bytes after it) or copy (01: 11-bit
// offset, length 4..11; 10: 16-bit offset; 11: 32-bit offset; lengths 1..64).  Copies may overlap
// their own output.  Appends to `out`, at most `block` bytes.
int snappy_one(const uint8_t *src, size_t len, uint64_t block, std::vector<uint8_t> &out) {
    size_t p = 0;
    uint64_t want = 0;
    for (int shift = 0;; shift += 7) {
        if (p >= len || shift > 28) return ORCD_DECOMPRESSION;
        const uint8_t b = src[p++];
        want |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) break;
    }
    if (want > block) return ORCD_DECOMPRESSION;
    const size_t at = out.size();
    out.resize(at + want);
    uint8_t *dst = out.data() + at;
    size_t o = 0;
    while (p < len) {
        const uint8_t tag = src[p++];
        if ((tag & 3) == 0) {
            size_t n = (size_t)(tag >> 2) + 1;
            if (n > 60) {
                const size_t extra = n - 60;  // 1..4 bytes of length
                if (len - p < extra) return ORCD_DECOMPRESSION;
                n = 0;
                for (size_t i = 0; i < extra; i++) n |= (size_t)src[p + i] << (8 * i);
                n += 1;
                p += extra;
            }
            if (n > len - p || n > want - o) return ORCD_DECOMPRESSION;
            memcpy(dst + o, src + p, n);
            p += n;
            o += n;
        } else {
            size_t n, off;
            if ((tag & 3) == 1) {
                if (len - p < 1) return ORCD_DECOMPRESSION;
                n = (size_t)((tag >> 2) & 7) + 4;
                off = ((size_t)(tag >> 5) << 8) | src[p];
                p += 1;
            } else if ((tag & 3) == 2) {
                if (len - p < 2) return ORCD_DECOMPRESSION;
                n = (size_t)(tag >> 2) + 1;
                off = (size_t)src[p] | ((size_t)src[p + 1] << 8);
                p += 2;
            } else {
                if (len - p < 4) return ORCD_DECOMPRESSION;
                n = (size_t)(tag >> 2) + 1;
                off = (size_t)src[p] | ((size_t)src[p + 1] << 8) | ((size_t)src[p + 2] << 16) | ((size_t)src[p + 3] << 24);
                p += 4;
            }
            if (off == 0 || off > o || n > want - o) return ORCD_DECOMPRESSION;
            for (size_t i = 0; i < n; i++) dst[o + i] = dst[o + i - off];  // (may overlap itself)
            o += n;
        }
    }
    return o == want ? ORCD_OK : ORCD_DECOMPRESSION;
}

// One LZ4 block / one Zstandard frame: neither says in ORC how large it gets, but none may exceed
// the file's compression block size.  The block size is the file's word (up to a gigabyte) and a
// chunk may be a few bytes, so the room is not zero-filled at that size for every chunk: a first
// try gets sixteen times the chunk's size (at least 64 KiB), and only a chunk that does not fit
// that -- or does not decode at all, which the libraries do not tell apart -- gets the whole block.
template <typename Try>
int grow_and_decode(size_t len, uint64_t block, std::vector<uint8_t> &out, Try attempt) {
    const size_t at = out.size();
    uint64_t room = std::min<uint64_t>(block, std::max<uint64_t>(64u << 10, (uint64_t)len * 16));
    for (;;) {
        out.resize(at + room);
        size_t got = 0;
        if (attempt(out.data() + at, (size_t)room, &got)) {
            out.resize(at + got);
            return ORCD_OK;
        }
        if (room >= block) break;
        room = block;
    }
    out.resize(at);
    return ORCD_DECOMPRESSION;
}

int lz4_one(const uint8_t *src, size_t len, uint64_t block, std::vector<uint8_t> &out) {
    if (len > 0x7FFFFFFFull || block > 0x7FFFFFFFull) return ORCD_DECOMPRESSION;
    return grow_and_decode(len, block, out, [&](uint8_t *dst, size_t room, size_t *got) {
        const int n = codec_libs().lz4_decompress_safe((const char *)src, (char *)dst, (int)len, (int)room);
        if (n < 0) return false;
        *got = (size_t)n;
        return true;
    });
}

int zstd_one(const uint8_t *src, size_t len, uint64_t block, std::vector<uint8_t> &out) {
    return grow_and_decode(len, block, out, [&](uint8_t *dst, size_t room, size_t *got) {
        const size_t n = codec_libs().zstd_decompress(dst, room, src, len);
        if (codec_libs().zstd_is_error(n)) return false;
        *got = n;
        return true;
    });
}

// One compressed chunk (raw deflate, src[0, len)) appended to `out`; it may not grow past `block`.
int inflate_one(const uint8_t *src, size_t len, uint64_t block, std::vector<uint8_t> &out) {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return ORCD_NO_MEMORY;
    zs.next_in = const_cast<Bytef *>(src);
    zs.avail_in = (uInt)len;
    int zrc = Z_OK;
    const size_t chunk_at = out.size();
    while (zrc != Z_STREAM_END) {
        const size_t at = out.size();
        // (no chunk decompresses to more than the file's compression block size: damaged
        // bytes must not make this grow without bound)
        if (at - chunk_at > block) {
            zrc = Z_DATA_ERROR;
            break;
        }
        const size_t room = std::min<size_t>(std::max<size_t>(4 * len, 1 << 16), (size_t)(block - (at - chunk_at)) + 1);
        out.resize(at + room);
        zs.next_out = out.data() + at;
        zs.avail_out = (uInt)room;
        zrc = inflate(&zs, Z_NO_FLUSH);
        out.resize(at + (room - zs.avail_out));
        if (zrc != Z_OK && zrc != Z_STREAM_END) break;        // data error, or input exhausted
        if (zrc == Z_OK && zs.avail_in == 0 && zs.avail_out != 0) break;  // ends before the stream does
    }
    inflateEnd(&zs);
    if (out.size() - chunk_at > block) return ORCD_DECOMPRESSION;  // (the last call may have filled the one byte too many)
    return zrc == Z_STREAM_END ? ORCD_OK : ORCD_DECOMPRESSION;
}

// Chunks are independent deflate streams, and inflating is by far the slowest thing the host does
// for a compressed file: streams of INFLATE_PARALLEL_MIN compressed bytes and more are inflated
// by several threads, each chunk into a buffer of its own, and put together afterwards.
constexpr size_t INFLATE_PARALLEL_MIN = 1u << 20;
constexpr unsigned INFLATE_MAX_THREADS = 16;

// ---- the process's budget of helper threads ------------------------------------------------------
// A context starts up to 8 copy threads, a read-ahead thread with readers of its own, 4 readers per
// compressed stripe and up to 16 decompression threads; a rank may hold several contexts and a node
// runs a rank per GPU.  Left alone that is hundreds of threads on a host of a few dozen cores, all
// of them wanted at the same moments.  The budget: ORCD_HELPER_THREADS when set, else the CPUs this
// process may use (OMP_NUM_THREADS where a launcher set it, else the hardware's) shared out over
// the ranks of the node (LOCAL_WORLD_SIZE, else WORLD_SIZE), at least 2 and at most 64.
struct HelperBudget {
    int budget;
    std::atomic<int> in_use{0}, peak{0};
    HelperBudget() {
        auto env_int = [](const char *name) {
            const char *e = getenv(name);
            return e && *e ? atoi(e) : 0;
        };
        int b = env_int("ORCD_HELPER_THREADS");
        if (b <= 0) {
            int cpus = env_int("OMP_NUM_THREADS");
            if (cpus <= 0) cpus = (int)std::max(1u, std::thread::hardware_concurrency());
            int ranks = env_int("LOCAL_WORLD_SIZE");
            if (ranks <= 0) ranks = env_int("WORLD_SIZE");
            if (ranks <= 0) ranks = 1;
            b = std::max(2, cpus / ranks);
        }
        budget = std::min(std::max(b, 1), 64);
    }
};
HelperBudget &helper_budget() {
    static HelperBudget b;
    return b;
}

// The helper threads themselves: one pool for the process, as many threads as the budget, started
// when first needed and never ended (they sleep between jobs; the process's exit ends them).  A
// caller posts a job -- a function that takes pieces of work off a counter of its own until none
// is left -- runs it itself, and the idle threads join in, each going to the open job that has
// the fewest helpers: contexts that feed stripes side by side share the threads evenly, and none
// waits for another to finish before it gets any.  (Leases handed out first come first served
// gave the first of four contexts fifteen threads and the others none: 2.1 instead of 6 GB/s on
// a zlib file.)
struct PoolJob {
    void (*fn)(void *, int);
    void *arg;
    int want;        // helpers that may join
    int joined = 0;  // ... and have (each gets slot 1 + its number; the caller is slot 0)
    int running = 0;
    bool closed = false;
};
struct HelperPool {
    std::mutex m;
    std::condition_variable work, done;
    std::vector<PoolJob *> jobs;
    int n_threads = 0;
    cpu_set_t bind;
    bool have_bind = false;

    void serve() {
        bool bound = false;
        std::unique_lock<std::mutex> lock(m);
        for (;;) {
            PoolJob *pick = nullptr;
            for (PoolJob *j : jobs)
                if (!j->closed && j->joined < j->want && (!pick || j->running < pick->running)) pick = j;
            if (!pick) {
                work.wait(lock);
                continue;
            }
            const int slot = ++pick->joined;
            pick->running++;
            HelperBudget &b = helper_budget();
            const int now = b.in_use.fetch_add(1) + 1;
            int pk = b.peak.load();
            while (pk < now && !b.peak.compare_exchange_weak(pk, now)) {
            }
            const bool bind_now = have_bind && !bound;
            lock.unlock();
            if (bind_now) {
                (void)pthread_setaffinity_np(pthread_self(), sizeof(bind), &bind);  // (best effort)
                bound = true;
            }
            pick->fn(pick->arg, slot);
            lock.lock();
            b.in_use.fetch_sub(1);
            if (--pick->running == 0) done.notify_all();
        }
    }
    // (with the lock held)
    void ensure_threads() {
        const int budget = helper_budget().budget;
        while (n_threads < budget) {
            try {
                std::thread([this] { serve(); }).detach();
            } catch (...) {
                break;  // as many as can be had
            }
            n_threads++;
        }
    }
};
HelperPool &helper_pool() {
    static HelperPool *p = new HelperPool();  // (never destroyed: its threads outlive every static)
    return *p;
}

int decompress_one(int codec, const uint8_t *src, size_t len, uint64_t block, std::vector<uint8_t> &out) {
    switch (codec) {
    case COMP_ZLIB: return inflate_one(src, len, block, out);
    case COMP_SNAPPY: return snappy_one(src, len, block, out);
    case COMP_LZ4: return lz4_one(src, len, block, out);
    case COMP_ZSTD: return zstd_one(src, len, block, out);
    default: return ORCD_UNSUPPORTED;
    }
}

int inflate_chunks(int codec, const uint8_t *src, size_t n, uint64_t block, std::vector<uint8_t> &out,
                   std::vector<std::pair<uint64_t, uint64_t>> *chunks) {
    const size_t out0 = out.size();
    if (block > (1ull << 30)) block = 1ull << 30;  // (whatever the PostScript claims)
    // the chunk headers first: where each chunk lies and whether it is stored as it is
    struct Chunk {
        size_t header, at, len;
        bool original;
    };
    std::vector<Chunk> list;
    size_t compressed_bytes = 0;
    for (size_t pos = 0; pos < n;) {
        if (n - pos < 3) return ORCD_DECOMPRESSION;
        const uint32_t h = (uint32_t)src[pos] | ((uint32_t)src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16);
        const size_t len = h >> 1;
        if (len > n - pos - 3) return ORCD_DECOMPRESSION;
        list.push_back({pos, pos + 3, len, (h & 1) != 0});
        if (!(h & 1)) compressed_bytes += len;
        pos += 3 + len;
    }
    unsigned n_threads = 1;
    size_t parallel_min = INFLATE_PARALLEL_MIN;
    if (const char *e = getenv("ORCD_INFLATE_PARALLEL_MIN"))  // (tests: the threaded way on small files)
        if (*e) parallel_min = (size_t)strtoull(e, nullptr, 10);
    if (compressed_bytes >= parallel_min && list.size() > 1) {
        n_threads = (unsigned)std::min<size_t>(INFLATE_MAX_THREADS, list.size());
    }
    if (n_threads <= 1) {
        for (const Chunk &c : list) {
            if (chunks) chunks->push_back({(uint64_t)c.header, (uint64_t)(out.size() - out0)});
            if (c.original) {
                out.insert(out.end(), src + c.at, src + c.at + c.len);
            } else if (c.len) {
                const int rc = decompress_one(codec, src + c.at, c.len, block, out);
                if (rc != ORCD_OK) return rc;
            }
        }
        return ORCD_OK;
    }
    // the threads take the chunks in turn
    std::vector<std::vector<uint8_t>> plain(list.size());
    std::vector<int> status(list.size(), ORCD_OK);
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    auto work = [&]() {
        for (size_t k; (k = next.fetch_add(1)) < list.size() && !failed;) {
            const Chunk &c = list[k];
            if (c.original || c.len == 0) continue;  // copied below, straight from the source
            try {
                status[k] = decompress_one(codec, src + c.at, c.len, block, plain[k]);
            } catch (const std::bad_alloc &) {
                status[k] = ORCD_NO_MEMORY;
            }
            if (status[k] != ORCD_OK) failed = true;
        }
    };
    // (this thread and whatever helpers the process's pool can spare)
    orcd_parallel((int)n_threads - 1, [&](int) { work(); });
    for (size_t k = 0; k < list.size(); k++)  // the first failure in stream order, as a serial walk reports it
        if (status[k] != ORCD_OK) return status[k];
    size_t total = 0;
    for (size_t k = 0; k < list.size(); k++) total += list[k].original ? list[k].len : plain[k].size();
    out.reserve(out.size() + total);
    for (size_t k = 0; k < list.size(); k++) {
        if (chunks) chunks->push_back({(uint64_t)list[k].header, (uint64_t)(out.size() - out0)});
        if (list[k].original) out.insert(out.end(), src + list[k].at, src + list[k].at + list[k].len);
        else out.insert(out.end(), plain[k].begin(), plain[k].end());
    }
    return ORCD_OK;
}

}  // namespace

extern "C" int orcd_internal_inflate_list(int codec, const orcd_internal_host_chunk *list, size_t n,
                               uint8_t *out, size_t pitch, uint32_t *sizes) {
    std::atomic<size_t> next{0};
    std::atomic<int> stopped{ORCD_OK};
    auto work = [&]() {
        for (size_t k; (k = next.fetch_add(1)) < n && stopped == ORCD_OK;) {
            int rc;
            std::vector<uint8_t> plain;
            try {
                rc = decompress_one(codec, list[k].src, list[k].len, std::min<uint64_t>(list[k].cap, pitch), plain);
            } catch (const std::bad_alloc &) {
                rc = ORCD_NO_MEMORY;
            }
            if (rc == ORCD_OK && plain.size() <= pitch) {
                sizes[k] = (uint32_t)plain.size();
                if (!plain.empty()) memcpy(out + k * pitch, plain.data(), plain.size());
            } else if (rc == ORCD_OK || rc == ORCD_DECOMPRESSION) {
                sizes[k] = 0xFFFFFFFFu;
            } else {
                stopped = rc;
            }
        }
    };
    const unsigned n_threads = (unsigned)std::min<size_t>(INFLATE_MAX_THREADS, std::max<size_t>(n, 1));
    orcd_parallel((int)n_threads - 1, [&](int) { work(); });
    return stopped;
}

namespace {

// A footer-like section as the protobuf reader wants it: the bytes themselves, or decompressed
// (src/read/mod.rs:161-183).
int plain_section(const orcd_file *f, const uint8_t *p, size_t n, std::vector<uint8_t> &out);

int parse_tail(orcd_file *f) {
    const Source &s = f->src;
    if (s.len < 4) return ORCD_OUT_OF_SPEC;
    // initial read of the tail (src/read/mod.rs:53-61)
    uint64_t tail_len = std::min<uint64_t>(s.len, DEFAULT_FOOTER_SIZE);
    std::vector<uint8_t> tail;
    if (!s.read_at(s.len - tail_len, tail_len, tail)) return ORCD_OUT_OF_SPEC;
    const uint64_t ps_len = tail.back();  // src/read/mod.rs:65
    if (ps_len + 1 > tail_len) return ORCD_OUT_OF_SPEC;
    uint64_t footer_length = 0, metadata_length = 0;
    bool has_footer_length = false;
    {
        Cursor c{tail.data() + tail_len - 1 - ps_len, tail.data() + tail_len - 1};
        uint32_t field, wire;
        uint64_t v;
        Cursor sub{nullptr, nullptr};
        while (c.next(field, wire, v, sub)) {
            if (wire != 0) continue;
            if (field == 1) { footer_length = v; has_footer_length = true; }
            else if (field == 2) f->compression = (int)v;
            else if (field == 3) f->compression_block = v;
            else if (field == 5) metadata_length = v;
        }
        if (!c.ok) return ORCD_INVALID_PROTO;
    }
    if (!has_footer_length) return ORCD_OUT_OF_SPEC;  // src/read/mod.rs:73
    // CompressionKind None and Zlib, as the reference (decompress/mod.rs:60-82: the rest is todo!())
    if (!codec_available(f->compression)) return ORCD_UNSUPPORTED;  // LZO, an unknown kind, or a codec library that is not there
    uint64_t need;
    if (!add3_ok(1 + ps_len, footer_length, metadata_length, need) || need > s.len) return ORCD_OUT_OF_SPEC;
    if (need > tail_len) {  // the crate panics here; re-read the tail at its exact size
        tail_len = need;
        if (!s.read_at(s.len - tail_len, tail_len, tail)) return ORCD_OUT_OF_SPEC;
    }
    // (need <= tail_len now, so the footer lies inside the buffer)
    if (footer_length > tail_len - 1 - ps_len) return ORCD_OUT_OF_SPEC;
    const uint8_t *footer_end = tail.data() + tail_len - 1 - ps_len;
    std::vector<uint8_t> footer;
    const int frc = plain_section(f, footer_end - footer_length, footer_length, footer);
    if (frc != ORCD_OK) return frc;
    Cursor c{footer.data(), footer.data() + footer.size()};
    uint32_t field, wire;
    uint64_t v;
    Cursor sub{nullptr, nullptr};
    while (c.next(field, wire, v, sub)) {
        if (field == 3 && wire == 2) {  // StripeInformation
            orcd_stripe_info si = {0, 0, 0, 0, 0};
            uint32_t f2, w2;
            uint64_t v2;
            Cursor s2{nullptr, nullptr};
            while (sub.next(f2, w2, v2, s2)) {
                if (w2 != 0) continue;
                if (f2 == 1) si.offset = v2;
                else if (f2 == 2) si.index_length = v2;
                else if (f2 == 3) si.data_length = v2;
                else if (f2 == 4) si.footer_length = v2;
                else if (f2 == 5) si.number_of_rows = v2;
            }
            if (!sub.ok) return ORCD_INVALID_PROTO;
            f->stripes.push_back(si);
        } else if (field == 4 && wire == 2) {  // Type
            TypeInfo t;
            uint32_t f2, w2;
            uint64_t v2;
            Cursor s2{nullptr, nullptr};
            bool ok = true;
            while (sub.next(f2, w2, v2, s2)) {
                if (f2 == 1 && w2 == 0) t.kind = (int)v2;
                else if (f2 == 2) read_repeated(w2, v2, s2, t.subtypes, ok);
            }
            if (!sub.ok || !ok) return ORCD_INVALID_PROTO;
            f->types.push_back(t);
        } else if (field == 6 && wire == 0) {
            f->number_of_rows = v;
        } else if (field == 8 && wire == 0) {
            f->row_index_stride = (uint32_t)v;
        }
    }
    if (!c.ok) return ORCD_INVALID_PROTO;
    return ORCD_OK;
}

int plain_section(const orcd_file *f, const uint8_t *p, size_t n, std::vector<uint8_t> &out) {
    out.clear();
    if (f->compression == 0) {
        out.assign(p, p + n);
        return ORCD_OK;
    }
    return inflate_chunks(f->compression, p, n, f->compression_block, out, nullptr);
}

int parse_stripe_footer(const std::vector<uint8_t> &buf, orcd_stripe_footer *out) {
    Cursor c{buf.data(), buf.data() + buf.size()};
    uint32_t field, wire;
    uint64_t v;
    Cursor sub{nullptr, nullptr};
    uint64_t offset = 0;
    while (c.next(field, wire, v, sub)) {
        if (field == 1 && wire == 2) {  // Stream
            orcd_stream_info si = {0, 0, 0, 0};
            uint32_t f2, w2;
            uint64_t v2;
            Cursor s2{nullptr, nullptr};
            while (sub.next(f2, w2, v2, s2)) {
                if (w2 != 0) continue;
                if (f2 == 1) si.kind = (uint32_t)v2;
                else if (f2 == 2) si.column = (uint32_t)v2;
                else if (f2 == 3) si.length = v2;
            }
            if (!sub.ok) return ORCD_INVALID_PROTO;
            si.offset = offset;  // streams are laid out back to back (src/read/mod.rs:127-138)
            if (!add_ok(offset, si.length, offset)) return ORCD_OUT_OF_SPEC;
            out->streams.push_back(si);
        } else if (field == 2 && wire == 2) {  // ColumnEncoding
            Encoding e;
            uint32_t f2, w2;
            uint64_t v2;
            Cursor s2{nullptr, nullptr};
            while (sub.next(f2, w2, v2, s2)) {
                if (w2 != 0) continue;
                if (f2 == 1) e.kind = (int)v2;
                else if (f2 == 2) { e.has_dict = true; e.dictionary_size = (uint32_t)v2; }
            }
            if (!sub.ok) return ORCD_INVALID_PROTO;
            out->columns.push_back(e);
        }
    }
    return c.ok ? ORCD_OK : ORCD_INVALID_PROTO;
}

// How many ROW_INDEX positions each stream of a column consumes when the file is uncompressed:
// RLE v2 -> (byte offset, values to skip); byte-RLE -> (byte offset, bytes to skip);
// boolean -> byte-RLE + (bits to skip); raw bytes -> (byte offset).
enum PosShape { POS_RLEV2 = 2, POS_BYTE_RLE = 2, POS_BOOLEAN = 3, POS_RAW = 1 };

int positions_of(int type_kind, int encoding, uint32_t stream_kind) {
    if (stream_kind == ORCD_STREAM_PRESENT) return POS_BOOLEAN;
    switch (type_kind) {
    case 0: return stream_kind == ORCD_STREAM_DATA ? POS_BOOLEAN : -1;                 // boolean
    case 1: return stream_kind == ORCD_STREAM_DATA ? POS_BYTE_RLE : -1;                // tinyint
    case 2: case 3: case 4: case 15:                                                   // ints, date
        return stream_kind == ORCD_STREAM_DATA ? POS_RLEV2 : -1;
    case 5: case 6: return stream_kind == ORCD_STREAM_DATA ? POS_RAW : -1;             // float, double
    case 7: case 16: case 17: case 8:                                                  // string-like, binary
        if (encoding == ORCD_ENC_DICTIONARY_V2 || encoding == ORCD_ENC_DICTIONARY) {
            // DATA (indices) is positioned; the dictionary streams are not
            return stream_kind == ORCD_STREAM_DATA ? POS_RLEV2 : 0;
        }
        if (stream_kind == ORCD_STREAM_DATA) return POS_RAW;
        if (stream_kind == ORCD_STREAM_LENGTH) return POS_RLEV2;
        return -1;
    default: return -1;
    }
}

int load_entry_points(orcd_file *file, orcd_column *col) {
    col->index_loaded = true;
    const orcd_stripe_footer *ft = col->footer;
    if (col->column >= file->types.size() || col->column >= ft->columns.size()) return ORCD_OK;
    const int type_kind = file->types[col->column].kind;
    const int encoding = ft->columns[col->column].kind;
    const orcd_stream_info *index = nullptr;
    for (const auto &s : ft->streams)
        if (s.column == col->column && s.kind == ORCD_STREAM_ROW_INDEX) index = &s;
    if (!index || index->length == 0) return ORCD_OK;
    // Each row-group entry lists the positions of the column's streams in stream-kind order
    // (PRESENT, DATA, LENGTH, ... as in the ORC specification's per-type stream tables), which
    // need not be the order of the streams inside the stripe.
    struct Slot { uint32_t kind; int first; };
    std::vector<Slot> slots;
    int cursor = 0;
    for (uint32_t kind = 0; kind < 8; kind++) {
        if (kind == ORCD_STREAM_ROW_INDEX || kind == ORCD_STREAM_BLOOM_FILTER) continue;
        bool present = false;
        for (const auto &s : ft->streams) present = present || (s.column == col->column && s.kind == kind);
        if (!present) continue;
        int n = positions_of(type_kind, encoding, kind);
        if (n < 0) return ORCD_OK;  // a layout this reader does not know: no entry points
        // in a compressed file a position starts with the offset of a compression chunk in the
        // stream, and the byte offset that follows counts inside that chunk's decompressed bytes
        if (n > 0 && file->compression != 0) n++;
        if (n > 0) slots.push_back({kind, cursor});
        cursor += n;
    }
    // (a handle without the column's bytes has no chunk tables to translate such positions with)
    if (file->compression != 0 && !col->compressed) return ORCD_OK;
    std::vector<uint8_t> buf;
    const orcd_stripe_info &si = file->stripes[col->stripe];
    uint64_t index_at;
    if (!add_ok(si.offset, index->offset, index_at) || !file->src.read_at(index_at, index->length, buf))
        return ORCD_OUT_OF_SPEC;
    if (file->compression != 0) {
        std::vector<uint8_t> plain;
        const int prc = plain_section(file, buf.data(), buf.size(), plain);
        if (prc != ORCD_OK) return prc;
        buf.swap(plain);
    }
    // which of the column's non-ROW_INDEX streams (in file order) each stream kind is
    int stream_no[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    {
        int k = 0;
        for (const auto &s : ft->streams) {
            if (s.column != col->column || s.kind == ORCD_STREAM_ROW_INDEX) continue;
            if (s.kind < 8) stream_no[s.kind] = k;
            k++;
        }
    }
    Cursor c{buf.data(), buf.data() + buf.size()};
    uint32_t field, wire;
    uint64_t v;
    Cursor sub{nullptr, nullptr};
    std::vector<uint64_t> pos;
    // The values-to-skip that follows the byte offset of an RLE v2 position (src/proto.rs:89-99) says
    // where in its run a row group starts.  When every run holds 512 values -- the bulk of a long
    // column -- row group k of a stream that has a value for every row starts at value k * stride,
    // in run floor(k * stride / 512), (k * stride) mod 512 values in: an index whose skips are
    // exactly that tells the decoder which window of 512 values each entry point is the first of.
    const uint64_t stride = file->row_index_stride;
    bool regular[8];
    for (int k = 0; k < 8; k++) regular[k] = stride >= 512;
    uint64_t group = 0;
    while (c.next(field, wire, v, sub)) {
        if (field != 1 || wire != 2) continue;  // RowIndexEntry
        pos.clear();
        uint32_t f2, w2;
        uint64_t v2;
        Cursor s2{nullptr, nullptr};
        bool ok = true;
        while (sub.next(f2, w2, v2, s2))
            if (f2 == 1) read_repeated(w2, v2, s2, pos, ok);
        if (!sub.ok || !ok) return ORCD_INVALID_PROTO;
        if ((int)pos.size() != cursor) {  // unexpected shape: drop all entry points
            for (auto &e : col->entries) e.clear();
            return ORCD_OK;
        }
        for (const auto &sl : slots) {
            std::vector<uint64_t> &e = col->entries[sl.kind];
            uint64_t off = pos[sl.first];
            if (file->compression != 0) {
                // chunk offset -> where that chunk's bytes start in the decompressed stream
                const int k = stream_no[sl.kind];
                if (k < 0 || (size_t)k >= col->chunks.size()) continue;
                const auto &table = col->chunks[k];
                auto it = std::lower_bound(table.begin(), table.end(), std::make_pair(off, (uint64_t)0));
                if (it == table.end() || it->first != off) {  // not a chunk of this stream: no entry points
                    for (auto &x : col->entries) x.clear();
                    return ORCD_OK;
                }
                if (!add_ok(it->second, pos[sl.first + 1], off) || off > col->plain_len[k]) {
                    for (auto &x : col->entries) x.clear();
                    return ORCD_OK;
                }
            }
            const size_t before = e.size();
            if (e.empty() || off > e.back()) e.push_back(off);  // ascending, no duplicates
            // (positions_of: an RLE v2 position ends with its values-to-skip)
            const int n_pos = positions_of(type_kind, encoding, sl.kind) + (file->compression != 0 ? 1 : 0);
            const bool rle = positions_of(type_kind, encoding, sl.kind) == POS_RLEV2 && type_kind != 1;
            if (!rle || e.size() != before + 1 || e.size() != group + 1 ||
                pos[sl.first + n_pos - 1] != (group * stride) % 512)
                regular[sl.kind] = false;
        }
        group++;
    }
    if (!c.ok) return ORCD_INVALID_PROTO;
    for (const auto &sl : slots)
        if (regular[sl.kind] && col->entries[sl.kind].size() == group && group >= 2) col->group_values[sl.kind] = stride;
    return ORCD_OK;
}

}  // namespace

extern "C" {

static int open_common(orcd_file *f, orcd_file **out) {
    const int rc = parse_tail(f);
    if (rc != ORCD_OK) {
        orcd_file_close(f);
        return rc;
    }
    *out = f;
    return ORCD_OK;
}

int orcd_file_open(const char *path, orcd_file **out) try {
    if (!path || !out) return ORCD_INVALID_ARGUMENT;
    *out = nullptr;
    FILE *fp = fopen(path, "rb");
    if (!fp) return ORCD_OUT_OF_SPEC;  // io errors collapse to OutOfSpec (src/error.rs:29-33)
    orcd_file *f = new orcd_file();
    f->src.f = fp;
    if (fseeko(fp, 0, SEEK_END) != 0) {
        orcd_file_close(f);
        return ORCD_OUT_OF_SPEC;
    }
    f->src.len = (uint64_t)ftello(fp);
    return open_common(f, out);
} ORCD_CATCH

int orcd_file_open_memory(const uint8_t *bytes, size_t len, orcd_file **out) try {
    if (!bytes || !out) return ORCD_INVALID_ARGUMENT;
    *out = nullptr;
    orcd_file *f = new orcd_file();
    f->src.mem = bytes;
    f->src.len = len;
    return open_common(f, out);
} ORCD_CATCH

void orcd_file_close(orcd_file *f) {
    if (!f) return;
    if (f->src.f) fclose(f->src.f);
    delete f;
}

size_t orcd_file_num_stripes(const orcd_file *f) { return f ? f->stripes.size() : 0; }

int orcd_file_stripe(const orcd_file *f, size_t stripe, orcd_stripe_info *info) try {
    if (!f || !info || stripe >= f->stripes.size()) return ORCD_INVALID_ARGUMENT;
    *info = f->stripes[stripe];
    return ORCD_OK;
} ORCD_CATCH

int orcd_file_compression(const orcd_file *f) { return f ? f->compression : -1; }
uint64_t orcd_internal_file_compression_block(const orcd_file *f) { return f ? f->compression_block : 0; }
uint64_t orcd_file_number_of_rows(const orcd_file *f) { return f ? f->number_of_rows : 0; }
uint32_t orcd_file_row_index_stride(const orcd_file *f) { return f ? f->row_index_stride : 0; }
size_t orcd_file_num_types(const orcd_file *f) { return f ? f->types.size() : 0; }

int orcd_file_type_kind(const orcd_file *f, size_t type_id) try {
    if (!f || type_id >= f->types.size()) return -1;
    return f->types[type_id].kind;
} ORCD_CATCH

int orcd_read_stripe_footer(orcd_file *f, size_t stripe, orcd_stripe_footer **out) try {
    if (!f || !out) return ORCD_INVALID_ARGUMENT;
    *out = nullptr;
    if (stripe >= f->stripes.size()) return ORCD_INVALID_ARGUMENT;
    const orcd_stripe_info &si = f->stripes[stripe];
    std::vector<uint8_t> buf;
    uint64_t footer_at;
    if (!add3_ok(si.offset, si.index_length, si.data_length, footer_at) ||
        !f->src.read_at(footer_at, si.footer_length, buf))
        return ORCD_OUT_OF_SPEC;
    std::vector<uint8_t> plain;
    const int prc = plain_section(f, buf.data(), buf.size(), plain);
    if (prc != ORCD_OK) return prc;
    orcd_stripe_footer *ft = new orcd_stripe_footer();
    ft->stripe = stripe;
    const int rc = parse_stripe_footer(plain, ft);
    if (rc != ORCD_OK) {
        delete ft;
        return rc;
    }
    *out = ft;
    return ORCD_OK;
} ORCD_CATCH

void orcd_stripe_footer_free(orcd_stripe_footer *ft) { delete ft; }

size_t orcd_stripe_footer_num_streams(const orcd_stripe_footer *ft) { return ft ? ft->streams.size() : 0; }

int orcd_stripe_footer_stream(const orcd_stripe_footer *ft, size_t i, orcd_stream_info *info) try {
    if (!ft || !info || i >= ft->streams.size()) return ORCD_INVALID_ARGUMENT;
    *info = ft->streams[i];
    return ORCD_OK;
} ORCD_CATCH

// The column's region read as it is in the file (src/read/mod.rs:129-144).
// (ext: bytes [ext_at, ext_at + ext_len) of the file that the caller holds already -- the region is
// taken from there, without a copy, and must lie inside)
static int read_column_region(orcd_file *f, size_t stripe, const orcd_stripe_footer *ft, uint32_t column,
                              orcd_column **out, const uint8_t *ext = nullptr, uint64_t ext_at = 0,
                              uint64_t ext_len = 0) {
    if (!f || !ft || !out) return ORCD_INVALID_ARGUMENT;
    *out = nullptr;
    if (stripe >= f->stripes.size()) return ORCD_INVALID_ARGUMENT;
    // first non-ROW_INDEX stream of the column and the total length of all of them
    bool found = false;
    uint64_t start = 0, length = 0;
    for (const auto &s : ft->streams) {
        if (s.column != column || s.kind == ORCD_STREAM_ROW_INDEX) continue;
        if (!found) {
            found = true;
            start = s.offset;
        }
        if (!add_ok(length, s.length, length)) return ORCD_OUT_OF_SPEC;
    }
    if (!found) return ORCD_INVALID_COLUMN;
    uint64_t region_at;
    if (!add_ok(f->stripes[stripe].offset, start, region_at)) return ORCD_OUT_OF_SPEC;
    std::unique_ptr<orcd_column> col(new orcd_column());
    if (ext) {
        if (region_at < ext_at || region_at - ext_at > ext_len || length > ext_len - (region_at - ext_at))
            return ORCD_INVALID_ARGUMENT;
        col->ext = ext + (region_at - ext_at);
        col->ext_len = (size_t)length;
    } else if (!f->src.read_at(region_at, length, col->data)) {
        return ORCD_OUT_OF_SPEC;
    }
    col->column = column;
    col->number_of_rows = f->stripes[stripe].number_of_rows;
    col->stripe = stripe;
    col->footer = ft;
    *out = col.release();
    return ORCD_OK;
}

// Every stream of a compressed file is a chain of compression chunks of its own
// (src/read/column.rs:38-55 wraps the stream's slice in a Decompressor): they are decompressed
// here, back to back, and take the place of the region in `data`.
static int inflate_column(orcd_file *f, orcd_column *col) {
    std::vector<uint8_t> plain;
    uint64_t start2 = 0;
    col->plain_off.clear();
    col->plain_len.clear();
    col->chunks.clear();
    for (const auto &s : col->footer->streams) {
        if (s.column != col->column || s.kind == ORCD_STREAM_ROW_INDEX) continue;
        if (start2 > col->region_len() || s.length > col->region_len() - start2) return ORCD_OUT_OF_SPEC;
        while (plain.size() & 15) plain.push_back(0);  // streams start 16-byte aligned
        col->plain_off.push_back(plain.size());
        col->chunks.emplace_back();
        const int rc = inflate_chunks(f->compression, col->region() + start2, (size_t)s.length, f->compression_block, plain, &col->chunks.back());
        if (rc != ORCD_OK) return rc;
        col->plain_len.push_back(plain.size() - col->plain_off.back());
        start2 += s.length;
    }
    col->data.swap(plain);
    col->ext = nullptr;
    col->ext_len = 0;
    col->compressed = true;
    col->raw = false;
    col->raw_chunks.clear();
    return ORCD_OK;
}

int orcd_read_stripe_column(orcd_file *f, size_t stripe, const orcd_stripe_footer *ft,
                            uint32_t column, orcd_column **out) try {
    orcd_column *col = nullptr;
    const int rc = read_column_region(f, stripe, ft, column, &col);
    if (rc != ORCD_OK) return rc;
    if (f->compression != 0) {
        const int irc = inflate_column(f, col);
        if (irc != ORCD_OK) {
            delete col;
            return irc;
        }
    }
    *out = col;
    return ORCD_OK;
} ORCD_CATCH

// ---- (not part of include/orcd.h) the column calls' way through a compressed file ----
// The region as it is, and the table of its compression chunks (headers walked, nothing
// decompressed): what happens to the chunks next is up to the caller -- inflate them on the
// device and report the sizes (orcd_internal_column_set_plain), or leave it to the host
// (orcd_internal_column_inflate_host).
int orcd_internal_read_column_raw(orcd_file *f, size_t stripe, const orcd_stripe_footer *ft, uint32_t column,
                                  orcd_column **out) {
    return orcd_internal_column_raw_from(f, stripe, ft, column, nullptr, 0, 0, out);
}

// (region_bytes: file[region_at, region_at + region_len) in the caller's memory, kept by the caller
// for as long as the handle is raw; null: the column's region is read from the file)
int orcd_internal_column_raw_from(orcd_file *f, size_t stripe, const orcd_stripe_footer *ft, uint32_t column,
                                  const uint8_t *region_bytes, uint64_t region_at, uint64_t region_len,
                                  orcd_column **out) try {
    orcd_column *col = nullptr;
    const int rc = read_column_region(f, stripe, ft, column, &col, region_bytes, region_at, region_len);
    if (rc != ORCD_OK) return rc;
    std::unique_ptr<orcd_column> hold(col);
    if (f->compression == 0) return ORCD_INVALID_ARGUMENT;
    uint64_t start = 0;
    uint32_t k = 0;
    for (const auto &s : ft->streams) {
        if (s.column != column || s.kind == ORCD_STREAM_ROW_INDEX) continue;
        if (start > col->region_len() || s.length > col->region_len() - start) return ORCD_OUT_OF_SPEC;
        const uint8_t *src = col->region() + start;
        const size_t n = (size_t)s.length;
        for (size_t pos = 0; pos < n;) {  // (the walk of inflate_chunks)
            if (n - pos < 3) return ORCD_DECOMPRESSION;
            const uint32_t h = (uint32_t)src[pos] | ((uint32_t)src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16);
            const size_t len = h >> 1;
            if (len > n - pos - 3) return ORCD_DECOMPRESSION;
            orcd_internal_chunk c;
            c.header = pos;
            c.at = start + pos + 3;
            c.len = (uint32_t)len;
            c.stream = k;
            c.original = h & 1;
            c.pad = 0;
            col->raw_chunks.push_back(c);
            pos += 3 + len;
        }
        start += s.length;
        k++;
    }
    col->raw = true;
    col->raw_streams = k;
    *out = hold.release();
    return ORCD_OK;
} ORCD_CATCH

int orcd_internal_column_raw(const orcd_column *col, const uint8_t **bytes, size_t *len,
                             const orcd_internal_chunk **chunks, size_t *n_chunks, size_t *n_streams) {
    if (!col || !col->raw) return ORCD_INVALID_ARGUMENT;
    *bytes = col->region();
    *len = col->region_len();
    *chunks = col->raw_chunks.data();
    *n_chunks = col->raw_chunks.size();
    *n_streams = col->raw_streams;
    return ORCD_OK;
}

int orcd_internal_column_inflate_host(orcd_file *f, orcd_column *col) try {
    if (!f || !col || !col->raw) return ORCD_INVALID_ARGUMENT;
    return inflate_column(f, col);
} ORCD_CATCH

// The chunks were decompressed elsewhere: stream k's bytes start at plain_off[k] of the caller's
// buffer and are plain_len[k] long, chunk j (in raw_chunks order) starts chunk_off[j] bytes into
// its stream.  The handle keeps the tables (ROW_INDEX positions need them) and drops the bytes.
int orcd_internal_column_set_plain(orcd_column *col, const uint64_t *plain_off, const uint64_t *plain_len,
                                   const uint64_t *chunk_off) try {
    if (!col || !col->raw) return ORCD_INVALID_ARGUMENT;
    col->plain_off.assign(plain_off, plain_off + col->raw_streams);
    col->plain_len.assign(plain_len, plain_len + col->raw_streams);
    col->chunks.assign(col->raw_streams, {});
    for (size_t j = 0; j < col->raw_chunks.size(); j++)
        col->chunks[col->raw_chunks[j].stream].push_back({col->raw_chunks[j].header, chunk_off[j]});
    col->compressed = true;
    col->raw = false;
    col->has_data = false;
    std::vector<uint8_t>().swap(col->data);
    col->ext = nullptr;
    col->ext_len = 0;
    std::vector<orcd_internal_chunk>().swap(col->raw_chunks);
    return ORCD_OK;
} ORCD_CATCH

// Where a stream's decompressed bytes lie (offset from the start of the decompressed column).
int orcd_internal_column_plain_range(const orcd_column *col, int kind, uint64_t *off, uint64_t *len) {
    if (!col || !col->compressed) return ORCD_INVALID_ARGUMENT;
    size_t k = 0;
    for (const auto &s : col->footer->streams) {
        if (s.column != col->column || s.kind == ORCD_STREAM_ROW_INDEX) continue;
        if ((int)s.kind == kind) {
            if (k >= col->plain_off.size()) return ORCD_OUT_OF_SPEC;
            *off = col->plain_off[k];
            *len = col->plain_len[k];
            return ORCD_OK;
        }
        k++;
    }
    return ORCD_INVALID_KIND;
}

void orcd_column_free(orcd_column *col) { delete col; }

// (not part of include/orcd.h: for the column calls of orcd_api.hip) the bytes a column handle
// holds -- of a compressed file: its streams decompressed, which is where orcd_column_get_stream
// points into.
int orcd_internal_column_bytes(const orcd_column *col, const uint8_t **bytes, size_t *len) {
    if (!col || !bytes || !len || !col->has_data) return ORCD_INVALID_ARGUMENT;
    *bytes = col->data.data();
    *len = col->data.size();
    return ORCD_OK;
}

int orcd_stripe_column_open(orcd_file *f, size_t stripe, const orcd_stripe_footer *ft, uint32_t column,
                            orcd_column **out) try {
    if (!f || !ft || !out) return ORCD_INVALID_ARGUMENT;
    *out = nullptr;
    if (stripe >= f->stripes.size()) return ORCD_INVALID_ARGUMENT;
    bool found = false;
    for (const auto &s : ft->streams) found = found || (s.column == column && s.kind != ORCD_STREAM_ROW_INDEX);
    if (!found) return ORCD_INVALID_COLUMN;
    orcd_column *col = new orcd_column();
    col->has_data = false;
    col->column = column;
    col->number_of_rows = f->stripes[stripe].number_of_rows;
    col->stripe = stripe;
    col->footer = ft;
    *out = col;
    return ORCD_OK;
} ORCD_CATCH

int orcd_column_region(const orcd_column *col, uint64_t *offset, uint64_t *len) try {
    if (!col || !offset || !len) return ORCD_INVALID_ARGUMENT;
    // what read_stripe_column reads: from the column's first non-ROW_INDEX stream, as many bytes
    // as its streams hold together (src/read/mod.rs:129-144)
    bool found = false;
    uint64_t start = 0, length = 0;
    for (const auto &s : col->footer->streams) {
        if (s.column != col->column || s.kind == ORCD_STREAM_ROW_INDEX) continue;
        if (!found) {
            found = true;
            start = s.offset;
        }
        if (!add_ok(length, s.length, length)) return ORCD_OUT_OF_SPEC;
    }
    if (!found) return ORCD_INVALID_COLUMN;
    *offset = start;
    *len = length;
    return ORCD_OK;
} ORCD_CATCH

int orcd_column_stream_range(const orcd_column *col, int kind, uint64_t *offset, uint64_t *len) try {
    if (!col || !offset || !len) return ORCD_INVALID_ARGUMENT;
    uint64_t region = 0, region_len = 0;
    const int rc = orcd_column_region(col, &region, &region_len);
    if (rc != ORCD_OK) return rc;
    // the place orcd_column_get_stream finds the stream at: a running offset over the column's
    // streams inside its region (src/read/column.rs:38-55)
    uint64_t start = 0;
    for (const auto &s : col->footer->streams) {
        if (s.column != col->column || s.kind == ORCD_STREAM_ROW_INDEX) continue;
        if ((int)s.kind == kind) {
            if (start > region_len || s.length > region_len - start) return ORCD_OUT_OF_SPEC;
            *offset = region + start;  // inside the region, whose end was formed without overflow
            *len = s.length;
            return ORCD_OK;
        }
        if (!add_ok(start, s.length, start)) return ORCD_OUT_OF_SPEC;
    }
    return ORCD_INVALID_KIND;
} ORCD_CATCH

int orcd_file_read_at(orcd_file *f, uint64_t offset, uint64_t len, uint8_t *dst) try {
    if (!f || (len && !dst)) return ORCD_INVALID_ARGUMENT;
    return f->src.read_into(offset, len, dst) ? ORCD_OK : ORCD_OUT_OF_SPEC;
} ORCD_CATCH

int orcd_column_get_stream(const orcd_column *col, int kind, const uint8_t **bytes, size_t *len) try {
    if (!col || !bytes || !len || !col->has_data) return ORCD_INVALID_ARGUMENT;
    // running offset over the column's non-ROW_INDEX streams (src/read/column.rs:38-55)
    uint64_t start = 0;
    size_t k = 0;
    for (const auto &s : col->footer->streams) {
        if (s.column != col->column || s.kind == ORCD_STREAM_ROW_INDEX) continue;
        if (col->compressed && (int)s.kind == kind) {
            if (k >= col->plain_off.size()) return ORCD_OUT_OF_SPEC;
            *bytes = col->data.data() + col->plain_off[k];
            *len = (size_t)col->plain_len[k];
            return ORCD_OK;
        }
        k++;
        if ((int)s.kind == kind) {
            if (start > col->data.size() || s.length > col->data.size() - start) return ORCD_OUT_OF_SPEC;
            *bytes = col->data.data() + start;
            *len = (size_t)s.length;
            return ORCD_OK;
        }
        if (!add_ok(start, s.length, start)) return ORCD_OUT_OF_SPEC;
    }
    return ORCD_INVALID_KIND;
} ORCD_CATCH

int orcd_column_encoding(const orcd_column *col, int *kind, uint32_t *dictionary_size) try {
    if (!col || col->column >= col->footer->columns.size()) return ORCD_INVALID_COLUMN;
    const Encoding &e = col->footer->columns[col->column];
    if (kind) *kind = e.kind;
    if (dictionary_size) *dictionary_size = e.has_dict ? e.dictionary_size : 0;
    return ORCD_OK;
} ORCD_CATCH

uint64_t orcd_column_number_of_rows(const orcd_column *col) { return col ? col->number_of_rows : 0; }

int orcd_column_entry_points(orcd_file *f, orcd_column *col, int kind, const uint64_t **entries,
                             size_t *n_entries) try {
    if (!f || !col || !entries || !n_entries || kind < 0 || kind >= 8) return ORCD_INVALID_ARGUMENT;
    if (!col->index_loaded) {
        col->index_status = load_entry_points(f, col);
        if (col->index_status != ORCD_OK) {
            for (auto &e : col->entries) e.clear();
            for (auto &g : col->group_values) g = 0;
        }
    }
    if (col->index_status != ORCD_OK) return col->index_status;
    *entries = col->entries[kind].data();
    *n_entries = col->entries[kind].size();
    return ORCD_OK;
} ORCD_CATCH

void orcd_internal_parallel_run(int max_helpers, void (*fn)(void *, int), void *arg) {
    if (max_helpers <= 0) {
        fn(arg, 0);
        return;
    }
    HelperPool &pool = helper_pool();
    PoolJob job{fn, arg, max_helpers};
    {
        std::lock_guard<std::mutex> lock(pool.m);
        pool.ensure_threads();
        pool.jobs.push_back(&job);
    }
    pool.work.notify_all();
    fn(arg, 0);
    std::unique_lock<std::mutex> lock(pool.m);
    job.closed = true;
    pool.jobs.erase(std::find(pool.jobs.begin(), pool.jobs.end(), &job));
    pool.done.wait(lock, [&] { return job.running == 0; });
}

void orcd_internal_pool_bind(const void *cpu_set, size_t cpu_set_bytes) {
    if (!cpu_set || cpu_set_bytes < sizeof(cpu_set_t)) return;
    HelperPool &pool = helper_pool();
    std::lock_guard<std::mutex> lock(pool.m);
    if (pool.have_bind) return;  // (the first context's device decides: a process drives one GPU)
    memcpy(&pool.bind, cpu_set, sizeof(cpu_set_t));
    pool.have_bind = true;
}

int orcd_helper_threads(int *budget, int *in_use, int *peak) {
    HelperBudget &b = helper_budget();
    if (budget) *budget = b.budget;
    if (in_use) *in_use = b.in_use.load();
    if (peak) *peak = b.peak.load();
    return ORCD_OK;
}

int orcd_internal_numa_cpus(const char *pci_bus_id, void *cpu_set, size_t cpu_set_bytes) {
    if (!pci_bus_id || !cpu_set || cpu_set_bytes < sizeof(cpu_set_t)) return 0;
    if (const char *e = getenv("ORCD_NUMA_BIND"))
        if (*e == '0') return 0;
    char path[256], line[4096];
    std::string id(pci_bus_id);
    for (auto &ch : id) ch = (char)tolower((unsigned char)ch);
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", id.c_str());
    FILE *fp = fopen(path, "r");
    if (!fp) return 0;
    int node = -1;
    const int got = fscanf(fp, "%d", &node);
    fclose(fp);
    if (got != 1 || node < 0) return 0;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    fp = fopen(path, "r");
    if (!fp) return 0;
    const bool have = fgets(line, sizeof(line), fp) != nullptr;
    fclose(fp);
    if (!have) return 0;
    cpu_set_t *set = (cpu_set_t *)cpu_set;
    CPU_ZERO(set);
    int n_cpus = 0;
    for (char *p = line; *p;) {  // "0-47,96-143"
        char *end;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            p = end;
        }
        for (long k = a; k <= b && k < CPU_SETSIZE; k++) {
            CPU_SET((int)k, set);
            n_cpus++;
        }
        if (*p == ',') p++;
        else break;
    }
    return n_cpus > 0;
}

uint64_t orcd_column_values_per_group(orcd_file *f, orcd_column *col, int kind) try {
    const uint64_t *e;
    size_t n;
    if (orcd_column_entry_points(f, col, kind, &e, &n) != ORCD_OK) return 0;
    return col->group_values[kind];
} catch (...) { return 0; }

}  // extern "C"

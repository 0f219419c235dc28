// Inflating the compression chunks of a zlib-compressed ORC file on the device.
//
// Every stream of such a file is a chain of chunks (src/read/decompress/mod.rs:9-17: a 3-byte
// header, then that many bytes), each a raw deflate stream of its own (mod.rs:60-80 hands them to
// flate2 one at a time) that decompresses to at most the file's compression block size.  A stripe
// holds hundreds to thousands of them, independent of each other: a workgroup of two wavefronts
// inflates one chunk.
//
// Inside a chunk deflate is serial -- a token's position in the bit stream is known only when the
// token before it has been decoded -- so each wavefront works as one and uses its lanes where
// there is something to share out.  The first wavefront turns bits into tokens: lane l decodes
// the token that would start l bits ahead (one round of table looks for 64 candidates, of which
// the chain of real tokens is then followed through registers), and the lanes build the Huffman
// tables of a block.  The second turns tokens into bytes: literals in a row by as many lanes,
// a match by all of them, finished output to memory.  In LDS: the last 32 KiB of output (all a
// match may refer to), a tile of the compressed bytes, a ring of tokens between the two, and per
// code a table that resolves codes of up to 12 (literal/length) or 10 (distance) bits in one look
// (longer ones, which are rare by construction, take the canonical walk over the per-length
// counts, one token at a time).
//
// The checks are those of zlib's inflate, so that the same chunks are accepted and refused as by
// the host path: block type 3, stored lengths that do not match, more than 286 / 30 symbols, code
// length sets that are over-subscribed or incomplete (but for a single one-bit code), a repeat
// with nothing to repeat or past the end, no end-of-block code, the symbols 286, 287, 30 and 31,
// a distance beyond the output so far, input that ends before the last block does, and output
// beyond the room given.
#include "orcd_internal.h"

namespace orcd {

namespace {

constexpr uint32_t INF_WIN = 32768;   // bytes of output kept (the most a distance can be)
constexpr uint32_t INF_IN = 1024;     // compressed bytes held at a time
constexpr uint32_t LIT_FAST = 12;     // literal/length codes up to this many bits resolve in one look
constexpr uint32_t DIST_FAST = 10;    // ... and distance (and code-length) codes
constexpr uint32_t INF_FLUSH = 4096;  // finished output goes to memory in pieces of at least this

template <uint32_t FAST>
struct HuffTable {
    uint16_t fast[1u << FAST];  // by the next FAST bits: symbol << 4 | length, 0: longer or no code
    uint16_t count[16];             // codes of each length
    uint16_t first_code[16];        // scratch of the build: the first code of each length
    uint16_t first_slot[16];        // ... and where its symbols start in `sorted`
    uint16_t sorted[288];           // symbols by (length, symbol)
};

struct Lds {
    uint8_t window[INF_WIN];
    uint8_t in[INF_IN];
    HuffTable<LIT_FAST> lit;
    HuffTable<DIST_FAST> dist;
    uint8_t lens[352];  // code-length code lengths at 0, the block's two length lists from 32 on
};

// Length symbol 257 + s: 3..10 one by one, then four symbols to each number of extra bits, 258 last.
__device__ __forceinline__ void length_code(uint32_t s, uint32_t &base, uint32_t &extra) {
    if (s < 8) {
        base = 3 + s;
        extra = 0;
    } else if (s == 28) {
        base = 258;
        extra = 0;
    } else {
        extra = (s >> 2) - 1;
        base = 3 + ((4 + (s & 3)) << extra);
    }
}
// Distance symbol d: 1..4 one by one, then two symbols to each number of extra bits.
__device__ __forceinline__ void distance_code(uint32_t d, uint32_t &base, uint32_t &extra) {
    if (d < 4) {
        base = 1 + d;
        extra = 0;
    } else {
        extra = (d >> 1) - 1;
        base = 1 + ((2 + (d & 1)) << extra);
    }
}
__constant__ uint8_t CLEN_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// Builds the decoding tables of one code from lens[0, n_sym) (in LDS).  `one_bit_ok`: an
// incomplete set is accepted when its longest code has one bit (zlib's rule for the literal/length
// and distance codes; never for the code-length code).  False: the set is not a prefix code.
template <uint32_t FAST>
__device__ bool build_table(HuffTable<FAST> &t, const uint8_t *lens, uint32_t n_sym, bool one_bit_ok, uint32_t lane) {
    for (uint32_t i = lane; i < (1u << FAST) / 2; i += 64) ((uint32_t *)t.fast)[i] = 0;
    if (lane < 16) {  // lane l counts the codes of l bits
        uint32_t n = 0;
        for (uint32_t s = 0; s < n_sym; s++) n += (lens[s] == lane);
        t.count[lane] = (uint16_t)(lane ? n : 0);
    }
    lds_fence();
    uint32_t longest = 0, code = 0, slot = 0;
    int32_t left = 1;
    bool over = false;
    for (uint32_t l = 1; l <= 15; l++) {
        const uint32_t n = uni(t.count[l]);
        if (n) longest = l;
        left = (left << 1) - (int32_t)n;
        if (left < 0) over = true;
        if (left < 0) left = 0;  // (keeps the shifts small; the set is refused below)
        if (lane == 0) {
            t.first_code[l] = (uint16_t)code;
            t.first_slot[l] = (uint16_t)slot;
        }
        code = (code + n) << 1;
        slot += n;
    }
    lds_fence();
    if (over) return false;
    if (longest == 0) return true;  // no code at all: every look-up fails
    if (left > 0 && !(one_bit_ok && longest == 1)) return false;
    if (lane >= 1 && lane < 16 && t.count[lane]) {  // lane l hands out the codes of l bits, in symbol order
        uint32_t c = t.first_code[lane], at = t.first_slot[lane];
        for (uint32_t s = 0; s < n_sym; s++) {
            if (lens[s] != lane) continue;
            t.sorted[at++] = (uint16_t)s;
            if (lane <= FAST) {  // the code arrives lowest bit first: every index that ends in it, reversed
                const uint32_t rev = __brev(c) >> (32 - lane);
                for (uint32_t i = rev; i < (1u << FAST); i += 1u << lane) t.fast[i] = (uint16_t)((s << 4) | lane);
            }
            c++;
        }
    }
    lds_fence();
    return true;
}

// The symbol the low bits of `bits` start with; `used`: its length, 0 when they start no code.
template <uint32_t FAST>
__device__ __forceinline__ uint32_t decode_symbol(const HuffTable<FAST> &t, uint64_t bits, uint32_t &used) {
    const uint32_t e = uni(t.fast[(uint32_t)bits & ((1u << FAST) - 1)]);
    if (e) {
        used = e & 15;
        return e >> 4;
    }
    uint32_t code = 0, first = 0, index = 0;
    for (uint32_t l = 1; l <= 15; l++) {
        code |= (uint32_t)(bits >> (l - 1)) & 1;
        const uint32_t n = uni(t.count[l]);
        if (code < first + n) {
            used = l;
            return uni(t.sorted[index + (code - first)]);
        }
        index += n;
        first = (first + n) << 1;
        code <<= 1;
    }
    used = 0;
    return 0;
}

}  // namespace

// ---- two wavefronts to a chunk ----------------------------------------------------------------
// The first turns bits into tokens (the header fields, the tables, the 64 candidates at a time)
// and puts them into a ring of Q_SLOTS words in LDS; the second takes them out in their order and
// turns them into bytes (literals in a row by as many lanes at once, matches by all lanes, the
// window, the flushes).  Either ends the other through `state`; both give up waiting after
// Q_PATIENCE looks, so that neither can hang whatever the other does.
constexpr uint32_t Q_SLOTS = 256;
constexpr uint32_t Q_RUNNING = 0, Q_DONE = 1, Q_FAILED = 2;
constexpr uint32_t Q_PATIENCE = 1u << 22;
struct TokenQueue {
    uint32_t slot[Q_SLOTS];  // a token as the lanes pack it: kind << 6 | literal or length << 8 | distance - 1 << 17
    uint32_t tail, head;     // tokens put in / taken out so far
    uint32_t state;
};

__global__ __launch_bounds__(128) void inflate_chunks_pair_kernel(const uint8_t *__restrict__ src, uint64_t src_total,
                                                            const InflateChunk *__restrict__ chunks, uint32_t n_chunks,
                                                            uint8_t *__restrict__ dst, uint32_t *__restrict__ sizes) {
    __shared__ __align__(16) Lds lds;
    __shared__ TokenQueue q;
    const uint32_t lane = threadIdx.x & 63;
    const bool writes = threadIdx.x >= 64;  // the second wavefront: tokens to bytes
    const uint32_t chunk = blockIdx.x;
    if (chunk >= n_chunks) return;
    if (threadIdx.x == 0) {
        q.tail = 0;
        q.head = 0;
        q.state = Q_RUNNING;
    }
    __syncthreads();
    const InflateChunk ck = chunks[chunk];
    const uint32_t src_len = ck.src_len, cap = ck.dst_cap;
    const uintptr_t in_addr = (uintptr_t)src + ck.src_off;          // the chunk's first byte
    const uintptr_t src_end = (uintptr_t)src + src_total;
    uint8_t *const out = dst + ck.dst_off;

    // ---- the bit reader: `n` unread bits in `buf`, the word at chunk offset `pos` comes next ----
    uint64_t buf = 0;
    uint32_t n = 0;
    int64_t pos = 0;          // multiple of 4 as an address (may start below 0: the word the chunk starts in)
    uintptr_t tile_at = 0;    // address of lds.in[0], a multiple of 16
    bool tile_ok = false;
    auto tile_fill = [&](uintptr_t addr) {
        tile_at = addr & ~(uintptr_t)15;
        __builtin_amdgcn_wave_barrier();
        const uintptr_t a = tile_at + (uintptr_t)lane * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (a < src_end) v = *(const uint4 *)a;  // (the 16 bytes around the buffer's ends are read whole)
        ((uint4 *)lds.in)[lane] = v;
        lds_fence();
        tile_ok = true;
    };
    auto next_word = [&]() -> uint32_t {  // the 4 bytes at chunk offset pos
        const uintptr_t a = in_addr + pos;
        if (!tile_ok || a < tile_at || a + 4 > tile_at + INF_IN) tile_fill(a);
        return uni(*(const uint32_t *)(lds.in + (a - tile_at)));
    };
    auto seek = [&](uint64_t byte) {  // the bit reader starts over at chunk offset `byte`
        const uint32_t skew = (uint32_t)((in_addr + byte) & 3);
        pos = (int64_t)byte - skew;
        const uint32_t w = next_word();
        buf = (uint64_t)(w >> (8 * skew));
        n = 32 - 8 * skew;
        pos += 4;
    };
    auto refill = [&]() {  // at least 33 bits afterwards
        if (n <= 32) {
            buf |= (uint64_t)next_word() << n;
            n += 32;
            pos += 4;
        }
    };
    auto take = [&](uint32_t k) -> uint32_t {  // k <= 32 bits
        const uint32_t v = (uint32_t)(buf & ((1ull << k) - 1));
        buf >>= k;
        n -= k;
        return v;
    };
    auto consumed = [&]() -> int64_t { return 8 * pos - (int64_t)n; };  // bits of the chunk read so far
    const int64_t limit = 8 * (int64_t)src_len;

    // ---- output: `wpos` bytes so far, the last INF_WIN of them in lds.window, `flushed` in memory ----
    uint32_t wpos = 0, flushed = 0;
    auto flush = [&](uint32_t upto) {
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = flushed + lane; i < upto; i += 64) out[i] = lds.window[i & (INF_WIN - 1)];
        flushed = upto;
    };

    // A match: `len` bytes that repeat what lies `dist` bytes back.  Every source byte was there
    // before the match began: of a match that overlaps itself, byte i repeats byte i mod dist.
    auto copy_match = [&](uint32_t at, uint32_t len, uint32_t dist) -> bool {  // (room for it: checked by the caller)
        if (dist > at) return false;
        lds_fence();
        if (dist >= len) {
            for (uint32_t i = lane; i < len; i += 64)
                lds.window[(at + i) & (INF_WIN - 1)] = lds.window[(at - dist + i) & (INF_WIN - 1)];
        } else {
            const float inv = 1.0f / (float)dist;  // (i, dist < 512: the quotient is off by one at most)
            for (uint32_t i = lane; i < len; i += 64) {
                int32_t r = (int32_t)i - (int32_t)((uint32_t)((float)i * inv) * dist);
                if (r < 0) r += (int32_t)dist;
                if (r >= (int32_t)dist) r -= (int32_t)dist;
                lds.window[(at + i) & (INF_WIN - 1)] = lds.window[(at - dist + (uint32_t)r) & (INF_WIN - 1)];
            }
        }
        lds_fence();
        return true;
    };

    if (writes) {
        // ---- the second wavefront: tokens, in their order, to bytes ----
        uint32_t head = 0;
        bool fine = true;
        uint32_t idle = 0;
        for (;;) {
            const uint32_t state = __hip_atomic_load(&q.state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t tail = __hip_atomic_load(&q.tail, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (tail == head) {
                if (state != Q_RUNNING) {
                    fine = state == Q_DONE;
                    break;
                }
                if (++idle > Q_PATIENCE) {  // (never seen: the other wavefront always ends, and says so)
                    fine = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            idle = 0;
            const uint32_t count = tail - head < 64 ? tail - head : 64;
            const uint32_t tok = lane < count ? q.slot[(head + lane) & (Q_SLOTS - 1)] : 0xFFu;
            const uint64_t literal = __ballot(lane < count && ((tok >> 6) & 3) == 0);
            uint32_t k = 0;
            while (k < count) {
                const uint64_t rest = ~(literal >> k);
                uint32_t run = rest ? (uint32_t)__builtin_ctzll(rest) : 64;  // literals in a row from token k on
                if (run > count - k) run = count - k;
                if (run) {
                    if (run > cap - wpos) {
                        fine = false;
                        break;
                    }
                    if (lane >= k && lane < k + run) lds.window[(wpos + lane - k) & (INF_WIN - 1)] = (uint8_t)(tok >> 8);
                    wpos += run;
                    k += run;
                } else {
                    const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)tok, (int)k);
                    const uint32_t len = (t >> 8) & 0x1FF;
                    if (len > cap - wpos || !copy_match(wpos, len, (t >> 17) + 1)) {
                        fine = false;
                        break;
                    }
                    wpos += len;
                    k++;
                }
                if (wpos - flushed >= INF_FLUSH) flush(wpos);
            }
            if (!fine) break;
            head += count;
            __hip_atomic_store(&q.head, head, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (!fine) __hip_atomic_store(&q.state, Q_FAILED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (fine) flush(wpos);
        if (lane == 0) sizes[chunk] = fine ? wpos : 0xFFFFFFFFu;
        return;
    }

    // ---- the first wavefront: bits to tokens ----
    // (it is the one the chunk waits for: where it shares a SIMD with a token-taking wavefront, of
    // this chunk or another, it goes first)
    __builtin_amdgcn_s_setprio(1);
    uint32_t tail = 0;
    // room for `count` tokens (at most 64); false: the other wavefront has given up
    uint32_t head_seen = 0;  // (the other side's count when last looked at: it only grows)
    auto room = [&](uint32_t count) -> bool {
        if (tail + count - head_seen <= Q_SLOTS) return true;
        for (uint32_t idle = 0;; idle++) {
            head_seen = __hip_atomic_load(&q.head, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (tail + count - head_seen <= Q_SLOTS) return true;
            if (__hip_atomic_load(&q.state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != Q_RUNNING || idle > Q_PATIENCE)
                return false;
            __builtin_amdgcn_s_sleep(1);
        }
    };
    auto publish = [&](uint32_t count) {
        lds_fence();
        tail += count;
        __hip_atomic_store(&q.tail, tail, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto push_one = [&](uint32_t tok) -> bool {
        if (!room(1)) return false;
        if (lane == 0) q.slot[tail & (Q_SLOTS - 1)] = tok;
        publish(1);
        return true;
    };
    const uintptr_t base16 = in_addr & ~(uintptr_t)15;            // bit 0 of the chunk is bit skew_bits from here
    const uint32_t skew_bits = (uint32_t)(in_addr & 15) * 8;

    const uint32_t clen_slot = CLEN_ORDER[lane < 19 ? lane : 0];  // where lane i puts the i-th code-length length
    bool ok = true, last = false;
    seek(0);
    while (ok && !last) {
        refill();
        last = take(1) != 0;
        const uint32_t type = take(2);
        if (type == 3) {
            ok = false;
            break;
        }
        if (type == 0) {  // stored: to the next byte, LEN, ~LEN, the bytes
            take(n & 7);
            refill();
            const uint32_t len = take(16), nlen = take(16);
            const int64_t at = consumed() / 8;
            if ((len ^ nlen) != 0xFFFF || consumed() > limit || at + len > (int64_t)src_len) {
                ok = false;
                break;
            }
            for (uint32_t done = 0; done < len && ok; done += 64) {  // as literals, a wavefront's worth at a time
                const uint32_t piece = len - done < 64 ? len - done : 64;
                if (!room(piece)) {
                    ok = false;
                    break;
                }
                if (lane < piece)
                    q.slot[(tail + lane) & (Q_SLOTS - 1)] = (uint32_t)*(const uint8_t *)(in_addr + at + done + lane) << 8;
                publish(piece);
            }
            if (!ok) break;
            seek((uint64_t)(at + len));
            continue;
        }
        // ---- the block's two codes ----
        if (type == 1) {
            for (uint32_t i = lane; i < 288; i += 64) lds.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
            lds_fence();
            (void)build_table(lds.lit, lds.lens, 288, true, lane);
            if (lane < 32) lds.lens[lane] = 5;
            lds_fence();
            (void)build_table(lds.dist, lds.lens, 32, true, lane);
        } else {
            const uint32_t nlen = take(5) + 257, ndist = take(5) + 1, ncode = take(4) + 4;
            if (nlen > 286 || ndist > 30) {
                ok = false;
                break;
            }
            if (lane < 19) lds.lens[lane] = 0;
            lds_fence();
            for (uint32_t i = 0; i < ncode; i++) {
                refill();
                const uint32_t l = take(3);
                if (lane == i) lds.lens[clen_slot] = (uint8_t)l;
            }
            lds_fence();
            if (!build_table(lds.dist, lds.lens, 19, false, lane)) {  // (the distance table's room, free until below)
                ok = false;
                break;
            }
            // the code lengths of both codes, as one list
            uint32_t have = 0, prev = 0;
            const uint32_t want = nlen + ndist;
            while (have < want) {
                refill();
                uint32_t used;
                const uint32_t sym = decode_symbol(lds.dist, buf, used);
                if (!used) {
                    ok = false;
                    break;
                }
                take(used);
                if (sym < 16) {
                    if (lane == 0) lds.lens[32 + have] = (uint8_t)sym;  // (kept clear of the code-length lengths)
                    prev = sym;
                    have++;
                    continue;
                }
                uint32_t rep, val = 0;
                if (sym == 16) {
                    if (have == 0) {
                        ok = false;
                        break;
                    }
                    val = prev;
                    rep = 3 + take(2);
                } else if (sym == 17) {
                    rep = 3 + take(3);
                } else {
                    rep = 11 + take(7);
                }
                if (have + rep > want) {
                    ok = false;
                    break;
                }
                for (uint32_t i = lane; i < rep; i += 64) lds.lens[32 + have + i] = (uint8_t)val;
                have += rep;
                prev = val;
            }
            if (!ok || consumed() > limit) {
                ok = false;
                break;
            }
            lds_fence();
            if (uni(lds.lens[32 + 256]) == 0) {  // no end-of-block code
                ok = false;
                break;
            }
            if (!build_table(lds.lit, lds.lens + 32, nlen, true, lane) ||
                !build_table(lds.dist, lds.lens + 32 + nlen, ndist, true, lane)) {
                ok = false;
                break;
            }
        }
        // ---- the block's symbols ----
        // Sixty-four at a time where that works: lane l decodes the token -- a literal, the end
        // of the block, or a length with its distance -- that would start l bits from here (two
        // table looks for all lanes together instead of two for every token), then the wavefront
        // follows the chain of tokens that really are there: the first starts at bit 0, the next
        // where it ends.  A lane whose bits start no short code, or nothing valid, says so; when
        // the chain arrives at such a place the one-token-at-a-time reader takes a look, and
        // reports what is wrong if something is.
        uint32_t bitpos = (uint32_t)consumed();
        bool eob = false;
        while (ok && !eob) {
            const uint32_t abit0 = skew_bits + bitpos;
            {
                const uintptr_t lo = base16 + 4 * (uintptr_t)(abit0 >> 5);
                if (!tile_ok || lo < tile_at || lo + 24 > tile_at + INF_IN) tile_fill(lo);
            }
            const uint32_t tile_off = (uint32_t)(tile_at - base16);
            uint32_t packed = 0;  // bits of the token | kind << 6 | literal or length << 8 | distance - 1 << 17
            {
                const uint32_t ab = abit0 + lane;
                const uint32_t *wp = (const uint32_t *)(lds.in + (4 * (ab >> 5) - tile_off));
                const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2], sh = ab & 31;
                const uint64_t view = ((((uint64_t)w1 << 32) | w0) >> sh) | (((uint64_t)w2 << 32) << (32 - sh));
                const uint32_t e = lds.lit.fast[(uint32_t)view & ((1u << LIT_FAST) - 1)];
                uint32_t used = e & 15;
                const uint32_t sym = e >> 4;
                if (e && sym < 256) {
                    packed = used | (sym << 8);
                } else if (e && sym == 256) {
                    packed = used | (1u << 6);
                } else if (e && sym < 286) {
                    uint32_t len, dist, extra;
                    length_code(sym - 257, len, extra);
                    len += (uint32_t)(view >> used) & ((1u << extra) - 1);
                    used += extra;
                    const uint32_t d = lds.dist.fast[(uint32_t)(view >> used) & ((1u << DIST_FAST) - 1)];
                    if (d && (d >> 4) < 30) {
                        used += d & 15;
                        distance_code(d >> 4, dist, extra);
                        dist += (uint32_t)(view >> used) & ((1u << extra) - 1);
                        used += extra;  // at most 15 + 5 + 15 + 13 bits
                        packed = used | (2u << 6) | (len << 8) | ((dist - 1) << 17);
                    }
                }
            }
            // The chain: which lanes' tokens are real.  A lone wavefront issues an instruction every
            // five cycles or so, dependent or not, so what a token costs is the instructions spent
            // on it: here a lane read, an add and the chain's bit, sixteen tokens at a stretch
            // without a branch.  A lane that could not make out its bits steps past the end, and so
            // stops the walk; it and the end of the block are dealt with afterwards.
            const uint32_t step = (packed & 63) ? (packed & 63) : 64u;
            uint64_t chain = 0;  // the lanes whose tokens are real, but for the end of the block
            uint32_t s = 0;
            while (s < 64) {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    chain |= s < 64 ? 1ull << s : 0ull;
                    s += (uint32_t)__builtin_amdgcn_readlane((int)step, (int)(s & 63));
                }
            }
            {   // (steps taken after the 64 bits were through have added to s: where the chain really ends)
                const uint32_t last = 63 - (uint32_t)__builtin_clzll(chain);
                s = last + (uint32_t)__builtin_amdgcn_readlane((int)step, (int)last);
            }
            {
                const uint64_t stops = __ballot((packed & 63) == 0 || ((packed >> 6) & 3) == 1) & chain;
                if (stops) {  // the chain ends there: before a lane without a token, after the end of the block
                    const uint32_t e = (uint32_t)__builtin_ctzll(stops);
                    const uint32_t tok = (uint32_t)__builtin_amdgcn_readlane((int)packed, (int)e);
                    chain &= (1ull << e) - 1;
                    s = e + (tok & 63);
                    eob = (tok & 63) != 0;
                }
            }
            if (chain) {
                const uint32_t count = (uint32_t)__popcll(chain);
                if (!room(count)) {
                    ok = false;
                    break;
                }
                if ((chain >> lane) & 1)
                    q.slot[(tail + (uint32_t)__popcll(chain & ((1ull << lane) - 1))) & (Q_SLOTS - 1)] = packed;
                publish(count);
            }
            if (!ok) break;
            bitpos += s;
            if ((int64_t)bitpos > limit) {
                ok = false;
                break;
            }
            if (eob || s) continue;
            // the token at bitpos is none the lanes could make out: one at a time
            seek((uint64_t)(bitpos >> 3));
            take(bitpos & 7);
            refill();
            uint32_t used;
            const uint32_t sym = decode_symbol(lds.lit, buf, used);
            if (!used) {
                ok = false;
                break;
            }
            take(used);
            if (sym < 256) {
                if (!push_one(sym << 8)) {
                    ok = false;
                    break;
                }
            } else if (sym == 256) {
                eob = true;
            } else {
                if (sym >= 286) {
                    ok = false;
                    break;
                }
                uint32_t len, dist, extra;
                length_code(sym - 257, len, extra);
                len += take(extra);
                refill();
                const uint32_t dsym = decode_symbol(lds.dist, buf, used);
                if (!used || dsym >= 30) {
                    ok = false;
                    break;
                }
                take(used);
                distance_code(dsym, dist, extra);
                dist += take(extra);
                if (!push_one((2u << 6) | (len << 8) | ((dist - 1) << 17))) {
                    ok = false;
                    break;
                }
            }
            if (consumed() > limit) {
                ok = false;
                break;
            }
            bitpos = (uint32_t)consumed();
        }
        if (!ok) break;
        seek((uint64_t)(bitpos >> 3));  // the next block's header, one field at a time
        take(bitpos & 7);
    }
    // (the other wavefront writes what the chunk came to, when it has seen this)
    __hip_atomic_store(&q.state, ok ? Q_DONE : Q_FAILED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}


hipError_t launch_inflate_chunks(hipStream_t st, const uint8_t *src, uint64_t src_total, const InflateChunk *chunks,
                                 uint32_t n_chunks, uint8_t *dst, uint32_t *sizes) {
    if (n_chunks == 0) return hipSuccess;
    inflate_chunks_pair_kernel<<<n_chunks, 128, 0, st>>>(src, src_total, chunks, n_chunks, dst, sizes);
    return hipGetLastError();
}

// Chunks that were not full in their slots, moved to where the stream wants them: piece k of
// `moves` is copied by one workgroup.
__global__ __launch_bounds__(256) void inflate_compact_kernel(const uint8_t *__restrict__ from, uint8_t *__restrict__ to,
                                                              const InflateMove *__restrict__ moves) {
    const InflateMove m = moves[blockIdx.x];
    for (uint64_t i = threadIdx.x; i < m.len; i += 256) to[m.to + i] = from[m.from + i];
}

hipError_t launch_inflate_compact(hipStream_t st, const uint8_t *from, uint8_t *to, const InflateMove *moves,
                                  uint32_t n_moves) {
    if (n_moves == 0) return hipSuccess;
    inflate_compact_kernel<<<n_moves, 256, 0, st>>>(from, to, moves);
    return hipGetLastError();
}

}  // namespace orcd

/*
 * orc_oracle.c -- CPU oracle for the ORC stream decode math (TEST INFRASTRUCTURE ONLY).
 *
 * A value-at-a-time restatement, in plain C, of the decode algorithms of the `orc-format` crate:
 * one header read per run, the run's packed bytes copied into a scratch buffer, then one
 * `unpack` per value -- the same structure as src/read/decode/rle_v2.rs, so that timing this file
 * is a fair stand-in for the crate's CPU path.  See orc_oracle.h for how it is pinned.
 */
#include "orc_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---- a bounded cursor standing in for `R: std::io::Read` ---------------------------------- */

typedef struct {
    const uint8_t *p;
    size_t len;
    size_t pos;
} reader;

static int read_u8(reader *r, uint8_t *v) { /* decode/mod.rs:17-21 */
    if (r->pos >= r->len) return ORC_ORACLE_OUT_OF_SPEC;
    *v = r->p[r->pos++];
    return ORC_ORACLE_OK;
}

/* reader.take(n).read_to_end(scratch) followed by a length check (rle_v2.rs:147-149). */
static int read_into(reader *r, size_t n, uint8_t **scratch, size_t *cap) {
    if (r->len - r->pos < n) return ORC_ORACLE_OUT_OF_SPEC;
    if (*cap < n + 8) {
        size_t ncap = n + 8 > 4160 ? n + 8 : 4160;
        uint8_t *ns = (uint8_t *)realloc(*scratch, ncap);
        if (!ns) return ORC_ORACLE_OUT_OF_SPEC;
        *scratch = ns;
        *cap = ncap;
    }
    memcpy(*scratch, r->p + r->pos, n);
    r->pos += n;
    return ORC_ORACLE_OK;
}

/* ---- header field decoding ------------------------------------------------------------------ */

/* rle_v2.rs:21-36 lists only the byte-aligned codes; this is the full table of the ORC v1
 * specification ("5 bit width encoding table"). */
static uint32_t decode_bit_width(uint32_t code) {
    if (code <= 23) return code + 1;
    switch (code) {
    case 24: return 26;
    case 25: return 28;
    case 26: return 30;
    case 27: return 32;
    case 28: return 40;
    case 29: return 48;
    case 30: return 56;
    default: return 64;
    }
}

static uint32_t closest_fixed_bits(uint32_t n) {
    if (n == 0) return 1;
    if (n <= 24) return n;
    if (n <= 26) return 26;
    if (n <= 28) return 28;
    if (n <= 30) return 30;
    if (n <= 32) return 32;
    if (n <= 40) return 40;
    if (n <= 48) return 48;
    if (n <= 56) return 56;
    return 64;
}

static uint32_t direct_bit_width(uint8_t header) { /* rle_v2.rs:38-41 */
    return decode_bit_width((header & 0x3E) >> 1);
}

static uint32_t delta_bit_width(uint8_t header) { /* rle_v2.rs:60-63: code 0 means "no deltas" */
    uint32_t code = (header & 0x3E) >> 1;
    return code == 0 ? 0 : decode_bit_width(code);
}

static uint32_t run_length(uint8_t header, uint8_t header1) { /* rle_v2.rs:65-69 */
    return 1u + ((((uint32_t)header & 1u) << 8) | header1);
}

int orc_oracle_unsigned_varint(const uint8_t *bytes, size_t len, uint64_t *value, size_t *used) {
    /* rle_v2.rs:71-89 */
    uint64_t v = 0;
    size_t j = 0;
    for (;;) {
        if (j > 9) return ORC_ORACLE_OUT_OF_SPEC;
        if (j >= len) return ORC_ORACLE_OUT_OF_SPEC;
        uint8_t b = bytes[j];
        v |= (uint64_t)(b & 0x7F) << (j * 7);
        j++;
        if ((b >> 7) == 0) break;
    }
    *value = v;
    if (used) *used = j;
    return ORC_ORACLE_OK;
}

static int read_varint(reader *r, uint64_t *v) {
    size_t used = 0;
    int rc = orc_oracle_unsigned_varint(r->p + r->pos, r->len - r->pos, v, &used);
    if (rc) return rc;
    r->pos += used;
    return ORC_ORACLE_OK;
}

int64_t orc_oracle_zigzag(uint64_t z) { /* rle_v2.rs:92-98 */
    return (z & 1) ? (int64_t)~(z >> 1) : (int64_t)(z >> 1);
}

/* Value `index` of a big-endian bit stream of `num_bits`-wide fields: bits
 * [index*num_bits, (index+1)*num_bits), most significant first.  Same contract as
 * rle_v2.rs:105-123, written from the definition so that it also holds for widths that are not
 * a whole number of bytes. */
uint64_t orc_oracle_unpack(const uint8_t *bytes, uint32_t num_bits, size_t index) {
    if (num_bits == 0) return 0;
    size_t start = (size_t)num_bits * index;
    size_t byte_start = start / 8;
    size_t byte_end = (start + num_bits + 7) / 8;
    /* gather the covering bytes, most significant first (at most 9 of them) */
    unsigned __int128 acc = 0;
    for (size_t i = byte_start; i < byte_end; i++) acc = (acc << 8) | bytes[i];
    uint32_t tail = (uint32_t)(byte_end * 8 - (start + num_bits));
    acc >>= tail;
    uint64_t mask = num_bits == 64 ? ~0ull : ((1ull << num_bits) - 1);
    return (uint64_t)acc & mask;
}

/* ---- one run -------------------------------------------------------------------------------- */

typedef struct {
    uint32_t kind, bit_width, length;
    /* ShortRepeat */
    uint64_t value;
    /* Delta */
    uint64_t base;      /* first value (already zigzag-decoded for signed streams) */
    int64_t delta_base; /* signed second-value delta, or the fixed delta when bit_width == 0 */
    /* PatchedBase */
    int64_t pb_base;
    uint32_t patch_width, gap_width, patch_len, patch_fixed_bits;
    size_t patch_offset; /* offset of the patch list inside scratch */
} run_state;

/* UnsignedRleV2Run::try_new / SignedRleV2Run::try_new (rle_v2.rs:441-461, 638-658): read the
 * header, then pull the run's payload into scratch. */
static int run_try_new(reader *r, int is_signed, run_state *run, uint8_t **scratch, size_t *cap) {
    uint8_t header, header1;
    int rc = read_u8(r, &header);
    if (rc) return rc;
    run->kind = header >> 6;
    switch (run->kind) {
    case ORC_RUN_SHORT_REPEAT: { /* rle_v2.rs:283-301 */
        uint32_t width = 1 + ((header >> 3) & 7);
        run->length = 3 + (header & 7);
        run->bit_width = width * 8;
        rc = read_into(r, width, scratch, cap);
        if (rc) return rc;
        uint64_t v = 0;
        for (uint32_t i = 0; i < width; i++) v = (v << 8) | (*scratch)[i];
        run->value = v;
        return ORC_ORACLE_OK;
    }
    case ORC_RUN_DIRECT: { /* rle_v2.rs:135-157 */
        rc = read_u8(r, &header1);
        if (rc) return rc;
        run->bit_width = direct_bit_width(header);
        run->length = run_length(header, header1);
        size_t additional = ((size_t)run->bit_width * run->length + 7) / 8;
        return read_into(r, additional, scratch, cap);
    }
    case ORC_RUN_DELTA: { /* rle_v2.rs:195-222, 343-366 */
        rc = read_u8(r, &header1);
        if (rc) return rc;
        run->bit_width = delta_bit_width(header);
        run->length = run_length(header, header1);
        uint64_t base, db;
        rc = read_varint(r, &base);
        if (rc) return rc;
        run->base = is_signed ? (uint64_t)orc_oracle_zigzag(base) : base;
        rc = read_varint(r, &db);
        if (rc) return rc;
        run->delta_base = orc_oracle_zigzag(db);
        if (run->bit_width != 0 && run->length < 2) return ORC_ORACLE_OUT_OF_SPEC;
        size_t additional =
            run->length < 2 ? 0 : ((size_t)(run->length - 2) * run->bit_width + 7) / 8;
        return read_into(r, additional, scratch, cap);
    }
    default: { /* PatchedBase: `todo!()` in rle_v2.rs:459,656; decoded here per the ORC v1 spec */
        uint8_t h2, h3;
        rc = read_u8(r, &header1);
        if (rc) return rc;
        rc = read_u8(r, &h2);
        if (rc) return rc;
        rc = read_u8(r, &h3);
        if (rc) return rc;
        run->bit_width = direct_bit_width(header);
        run->length = run_length(header, header1);
        uint32_t base_bytes = 1 + ((h2 >> 5) & 7);
        run->patch_width = decode_bit_width(h2 & 0x1F);
        run->gap_width = 1 + ((h3 >> 5) & 7);
        run->patch_len = h3 & 0x1F;
        if (run->patch_len == 0) return ORC_ORACLE_OUT_OF_SPEC;
        if (run->patch_width + run->gap_width > 64) return ORC_ORACLE_OUT_OF_SPEC;
        run->patch_fixed_bits = closest_fixed_bits(run->patch_width + run->gap_width);
        size_t data_bytes = ((size_t)run->bit_width * run->length + 7) / 8;
        size_t patch_bytes = ((size_t)run->patch_fixed_bits * run->patch_len + 7) / 8;
        rc = read_into(r, base_bytes + data_bytes + patch_bytes, scratch, cap);
        if (rc) return rc;
        uint64_t b = 0;
        for (uint32_t i = 0; i < base_bytes; i++) b = (b << 8) | (*scratch)[i];
        uint64_t sign = 1ull << (base_bytes * 8 - 1);
        run->pb_base = (b & sign) ? -(int64_t)(b & ~sign) : (int64_t)b;
        /* keep the packed values at the start of scratch */
        memmove(*scratch, *scratch + base_bytes, data_bytes + patch_bytes);
        run->patch_offset = data_bytes;
        return ORC_ORACLE_OK;
    }
    }
}

/* Emit up to `want` values of the run, one `next()` at a time (rle_v2.rs:169-175, 240-264,
 * 318-323, 382-408, 586-588, 618-620). */
static void run_emit(const run_state *run, const uint8_t *scratch, int is_signed, size_t want,
                     int64_t *out) {
    size_t n = run->length < want ? run->length : want;
    switch (run->kind) {
    case ORC_RUN_SHORT_REPEAT: {
        int64_t v = is_signed ? orc_oracle_zigzag(run->value) : (int64_t)run->value;
        for (size_t i = 0; i < n; i++) out[i] = v;
        break;
    }
    case ORC_RUN_DIRECT:
        for (size_t i = 0; i < n; i++) {
            uint64_t u = orc_oracle_unpack(scratch, run->bit_width, i);
            out[i] = is_signed ? orc_oracle_zigzag(u) : (int64_t)u;
        }
        break;
    case ORC_RUN_DELTA: {
        /* two's-complement wrapping arithmetic; the ORC readers subtract when delta_base < 0 */
        uint64_t acc = run->base;
        for (size_t i = 0; i < n; i++) {
            if (i == 0) {
                /* base */
            } else if (i == 1 || run->bit_width == 0) {
                acc += (uint64_t)run->delta_base;
            } else {
                uint64_t d = orc_oracle_unpack(scratch, run->bit_width, i - 2);
                if (run->delta_base < 0) acc -= d; else acc += d;
            }
            out[i] = (int64_t)acc;
        }
        break;
    }
    default: { /* PatchedBase */
        const uint8_t *patches = scratch + run->patch_offset;
        uint64_t patch_mask =
            run->patch_width == 64 ? ~0ull : ((1ull << run->patch_width) - 1);
        uint32_t pidx = 0;
        uint64_t next_patch_pos = 0, cur_patch = 0;
        int have_patch = 0;
        /* adjustGapAndPatch of the ORC readers: entries with gap 255 and patch 0 only extend
         * the gap */
        #define NEXT_PATCH(origin)                                                              \
            do {                                                                                \
                uint64_t gap_total = 0;                                                         \
                have_patch = 0;                                                                 \
                while (pidx < run->patch_len) {                                                 \
                    uint64_t e = orc_oracle_unpack(patches, run->patch_fixed_bits, pidx);       \
                    uint64_t gap = run->patch_width == 64 ? 0 : (e >> run->patch_width);        \
                    uint64_t pv = e & patch_mask;                                               \
                    pidx++;                                                                     \
                    gap_total += gap;                                                           \
                    if (gap == 255 && pv == 0) continue;                                        \
                    cur_patch = pv;                                                             \
                    next_patch_pos = (origin) + gap_total;                                      \
                    have_patch = 1;                                                             \
                    break;                                                                      \
                }                                                                               \
            } while (0)
        NEXT_PATCH(0);
        for (size_t i = 0; i < run->length; i++) {
            uint64_t u = orc_oracle_unpack(scratch, run->bit_width, i);
            if (have_patch && i == next_patch_pos) {
                if (run->bit_width < 64) u |= cur_patch << run->bit_width;
                NEXT_PATCH(i);
            }
            if (i < n) out[i] = (int64_t)((uint64_t)run->pb_base + u);
        }
        #undef NEXT_PATCH
        break;
    }
    }
}

/* ---- RLE v2 iterators ----------------------------------------------------------------------- */

int orc_oracle_rlev2_i64(const uint8_t *stream, size_t len, size_t num_values, int is_signed,
                         int64_t *out, size_t *consumed) {
    reader r = {stream, len, 0};
    uint8_t *scratch = NULL;
    size_t cap = 0;
    size_t remaining = num_values; /* SignedRleV2RunIter.length, rle_v2.rs:677-710 */
    int rc = ORC_ORACLE_OK;
    while (remaining != 0) {
        run_state run;
        memset(&run, 0, sizeof run);
        rc = run_try_new(&r, is_signed, &run, &scratch, &cap);
        if (rc) break;
        run_emit(&run, scratch, is_signed, remaining, out + (num_values - remaining));
        remaining -= run.length < remaining ? run.length : remaining;
    }
    free(scratch);
    if (consumed) *consumed = r.pos;
    return rc;
}

int orc_oracle_rlev2_runs(const uint8_t *stream, size_t len, size_t num_values,
                          orc_oracle_run *runs, size_t max_runs, size_t *n_runs) {
    reader r = {stream, len, 0};
    uint8_t *scratch = NULL;
    size_t cap = 0, count = 0, remaining = num_values;
    int rc = ORC_ORACLE_OK;
    while (remaining != 0) {
        run_state run;
        memset(&run, 0, sizeof run);
        size_t start = r.pos;
        rc = run_try_new(&r, 0, &run, &scratch, &cap);
        if (rc) break;
        if (count < max_runs) {
            runs[count].byte_offset = start;
            runs[count].byte_length = r.pos - start;
            runs[count].kind = run.kind;
            runs[count].bit_width = run.bit_width;
            runs[count].length = run.length;
            runs[count].reserved = 0;
        }
        count++;
        remaining -= run.length < remaining ? run.length : remaining;
    }
    free(scratch);
    *n_runs = count;
    return rc;
}

/* ---- byte-RLE and booleans ------------------------------------------------------------------ */

/* boolean_rle.rs:44-62 reads a control byte: >= 0 is a run of (control + 3) copies of the next
 * byte, < 0 is (-control) literal bytes.  The run iterator of the crate over-counts runs by one
 * byte and never leaves a literal run (boolean_rle.rs:107-113,128); the lengths here are the ORC
 * specification's. */
/* Integer RLE version 1 (ORC v1 specification, "Integer Run Length Encoding, version 1"; the
 * reference lists it as not implemented, README.md:23-25).  A control byte 0..127 starts a run of
 * control + 3 values: a signed delta byte, then the base as a varint (zigzag for signed streams,
 * like every signed varint of rle_v2.rs:100-102), value i = base + i * delta (wrapping).  A control
 * byte 128..255 starts 256 - control literal values, a varint each. */
int orc_oracle_rlev1_i64(const uint8_t *stream, size_t len, size_t num_values, int is_signed,
                         int64_t *out, size_t *consumed) {
    reader r = {stream, len, 0};
    size_t n = 0;
    while (n < num_values) {
        uint8_t control;
        int rc = read_u8(&r, &control);
        if (rc) return rc;
        if (control < 128) {
            uint8_t d;
            uint64_t v;
            if ((rc = read_u8(&r, &d))) return rc;
            if ((rc = read_varint(&r, &v))) return rc;
            const uint64_t base = is_signed ? (uint64_t)orc_oracle_zigzag(v) : v;
            const uint64_t delta = (uint64_t)(int64_t)(int8_t)d;
            const size_t count = (size_t)control + 3;
            for (size_t i = 0; i < count && n < num_values; i++) out[n++] = (int64_t)(base + (uint64_t)i * delta);
        } else {
            const size_t count = 256 - (size_t)control;
            for (size_t i = 0; i < count; i++) {   /* the whole group is read, as a run is */
                uint64_t v;
                if ((rc = read_varint(&r, &v))) return rc;
                if (n < num_values) out[n++] = is_signed ? orc_oracle_zigzag(v) : (int64_t)v;
            }
        }
    }
    if (consumed) *consumed = r.pos;
    return ORC_ORACLE_OK;
}

int orc_oracle_byte_rle(const uint8_t *stream, size_t len, size_t num_bytes, uint8_t *out,
                        size_t *consumed) {
    reader r = {stream, len, 0};
    size_t produced = 0;
    int rc = ORC_ORACLE_OK;
    while (produced < num_bytes) {
        uint8_t control, value;
        rc = read_u8(&r, &control);
        if (rc) break;
        if ((int8_t)control >= 0) {
            size_t n = (size_t)control + 3;
            rc = read_u8(&r, &value);
            if (rc) break;
            for (size_t i = 0; i < n && produced < num_bytes; i++) out[produced++] = value;
        } else {
            size_t n = (size_t)(-(int)(int8_t)control);
            if (r.len - r.pos < n) {
                rc = ORC_ORACLE_OUT_OF_SPEC;
                break;
            }
            for (size_t i = 0; i < n; i++) {
                if (produced < num_bytes) out[produced++] = r.p[r.pos + i];
            }
            r.pos += n;
        }
    }
    if (consumed) *consumed = r.pos;
    return rc;
}

int orc_oracle_boolean(const uint8_t *stream, size_t len, size_t num_bits, uint8_t *out,
                       uint64_t *set_bits) {
    size_t num_bytes = (num_bits + 7) / 8;
    uint8_t *bytes = (uint8_t *)malloc(num_bytes ? num_bytes : 1);
    if (!bytes) return ORC_ORACLE_OUT_OF_SPEC;
    int rc = orc_oracle_byte_rle(stream, len, num_bytes, bytes, NULL);
    uint64_t ones = 0;
    if (rc == ORC_ORACLE_OK) {
        for (size_t i = 0; i < num_bits; i++) { /* mask = 128 >> position, boolean_rle.rs:98 */
            uint8_t bit = (bytes[i >> 3] >> (7 - (i & 7))) & 1;
            out[i] = bit;
            ones += bit;
        }
    }
    free(bytes);
    if (set_bits) *set_bits = ones;
    return rc;
}

/* ---- floats --------------------------------------------------------------------------------- */

int orc_oracle_float(const uint8_t *stream, size_t len, size_t elem_size, size_t num_values,
                     void *out) {
    if (elem_size != 4 && elem_size != 8) return ORC_ORACLE_UNSUPPORTED;
    uint8_t *o = (uint8_t *)out;
    size_t pos = 0;
    for (size_t i = 0; i < num_values; i++) { /* float.rs:73-84 */
        if (len - pos < elem_size) return ORC_ORACLE_DECODE_FLOAT;
        memcpy(o + i * elem_size, stream + pos, elem_size); /* from_le_bytes on a LE host */
        pos += elem_size;
    }
    return ORC_ORACLE_OK;
}

/* ---- strings -------------------------------------------------------------------------------- */

static int utf8_valid(const uint8_t *s, size_t n) { /* std::str::from_utf8 acceptance rules */
    size_t i = 0;
    while (i < n) {
        uint8_t c = s[i];
        if (c < 0x80) {
            i++;
        } else if (c >= 0xC2 && c <= 0xDF) {
            if (i + 1 >= n || (s[i + 1] & 0xC0) != 0x80) return 0;
            i += 2;
        } else if (c >= 0xE0 && c <= 0xEF) {
            if (i + 2 >= n) return 0;
            uint8_t c1 = s[i + 1], c2 = s[i + 2];
            if ((c1 & 0xC0) != 0x80 || (c2 & 0xC0) != 0x80) return 0;
            if (c == 0xE0 && c1 < 0xA0) return 0;
            if (c == 0xED && c1 > 0x9F) return 0;
            i += 3;
        } else if (c >= 0xF0 && c <= 0xF4) {
            if (i + 3 >= n) return 0;
            uint8_t c1 = s[i + 1], c2 = s[i + 2], c3 = s[i + 3];
            if ((c1 & 0xC0) != 0x80 || (c2 & 0xC0) != 0x80 || (c3 & 0xC0) != 0x80) return 0;
            if (c == 0xF0 && c1 < 0x90) return 0;
            if (c == 0xF4 && c1 > 0x8F) return 0;
            i += 4;
        } else {
            return 0;
        }
    }
    return 1;
}

int orc_oracle_string_direct(const uint8_t *length_stream, size_t length_len, const uint8_t *data,
                             size_t data_len, size_t num_values, int check_utf8, int64_t *offsets,
                             uint8_t *bytes_out) {
    int64_t *lengths = (int64_t *)malloc((num_values ? num_values : 1) * sizeof(int64_t));
    if (!lengths) return ORC_ORACLE_OUT_OF_SPEC;
    int rc = orc_oracle_rlev2_i64(length_stream, length_len, num_values, 0, lengths, NULL);
    size_t pos = 0; /* the cursor of `Values` (variable_length.rs:15-23) */
    for (size_t i = 0; rc == ORC_ORACLE_OK && i < num_values; i++) {
        uint64_t l = (uint64_t)lengths[i];
        offsets[i] = (int64_t)pos;
        if (l > data_len - pos) {
            rc = ORC_ORACLE_OUT_OF_SPEC;
            break;
        }
        if (check_utf8 && !utf8_valid(data + pos, (size_t)l)) {
            rc = ORC_ORACLE_INVALID_UTF8;
            break;
        }
        if (bytes_out) memcpy(bytes_out + pos, data + pos, (size_t)l);
        pos += (size_t)l;
    }
    if (rc == ORC_ORACLE_OK) offsets[num_values] = (int64_t)pos;
    free(lengths);
    return rc;
}

int orc_oracle_string_dictionary(const int64_t *dict_offsets, const uint8_t *dict_bytes,
                                 size_t dict_size, const int64_t *indices, size_t num_values,
                                 int64_t *offsets_out, uint8_t *bytes_out) {
    size_t pos = 0;
    for (size_t i = 0; i < num_values; i++) {
        uint64_t idx = (uint64_t)indices[i];
        if (idx >= dict_size) return ORC_ORACLE_OUT_OF_SPEC; /* deserialize.rs:162 */
        size_t b = (size_t)dict_offsets[idx], e = (size_t)dict_offsets[idx + 1];
        offsets_out[i] = (int64_t)pos;
        if (bytes_out) memcpy(bytes_out + pos, dict_bytes + b, e - b);
        pos += e - b;
    }
    offsets_out[num_values] = (int64_t)pos;
    return ORC_ORACLE_OK;
}

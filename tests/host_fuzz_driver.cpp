// Standalone driver for the host reader (orcd_host.cpp) meant to be built with
// -fsanitize=address,undefined: opens the given ORC files, then many damaged copies of each,
// and touches everything the reader offers.  Exit status 0 = no sanitizer report.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "orcd.h"
#include "../orc_format_amd/csrc/orcd_column_internal.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static unsigned long checksum = 0;

static int walk(const std::vector<uint8_t> &image) {
    orcd_file *f = nullptr;
    int rc = orcd_file_open_memory(image.data(), image.size(), &f);
    if (rc != ORCD_OK) return rc;
    const size_t n_types = orcd_file_num_types(f);
    for (size_t t = 0; t < n_types && t < 64; t++) checksum += orcd_file_type_kind(f, t);
    for (size_t stripe = 0; stripe < orcd_file_num_stripes(f) && stripe < 8; stripe++) {
        orcd_stripe_info si;
        orcd_file_stripe(f, stripe, &si);
        orcd_stripe_footer *ft = nullptr;
        if (orcd_read_stripe_footer(f, stripe, &ft) != ORCD_OK) continue;
        for (size_t i = 0; i < orcd_stripe_footer_num_streams(ft) && i < 256; i++) {
            orcd_stream_info info;
            orcd_stripe_footer_stream(ft, i, &info);
            checksum += info.kind;
        }
        for (uint32_t column = 0; column < n_types + 2 && column < 26; column++) {
            orcd_column *col = nullptr;
            if (orcd_read_stripe_column(f, stripe, ft, column, &col) != ORCD_OK) continue;
            int kind;
            uint32_t dict;
            orcd_column_encoding(col, &kind, &dict);
            for (int k = 0; k < 8; k++) {
                const uint8_t *bytes;
                size_t len;
                if (orcd_column_get_stream(col, k, &bytes, &len) == ORCD_OK && len)
                    checksum += bytes[0] + bytes[len - 1];
                const uint64_t *entries;
                size_t n_entries;
                if (orcd_column_entry_points(f, col, k, &entries, &n_entries) == ORCD_OK && n_entries)
                    checksum += entries[0] + entries[n_entries - 1];
            }
            orcd_column_free(col);
            // the way of the column calls through a compressed file: the region as it is with its
            // chunk table, then either the host's inflate or sizes reported from elsewhere (here:
            // every chunk claims a size of its own making) and the entry points on top of them
            for (int way = 0; way < 2 && orcd_file_compression(f) != 0; way++) {
                orcd_column *raw = nullptr;
                if (orcd_internal_read_column_raw(f, stripe, ft, column, &raw) != ORCD_OK) break;
                const uint8_t *bytes;
                size_t len, n_chunks, n_streams;
                const orcd_internal_chunk *chunks;
                if (orcd_internal_column_raw(raw, &bytes, &len, &chunks, &n_chunks, &n_streams) != ORCD_OK) return 98;
                for (size_t j = 0; j < n_chunks; j++) {
                    if (chunks[j].stream >= n_streams || chunks[j].at > len || chunks[j].len > len - chunks[j].at) return 97;
                    if (chunks[j].len) checksum += bytes[chunks[j].at] + bytes[chunks[j].at + chunks[j].len - 1];
                }
                if (way == 1) {  // the host's share of a split inflate: every chunk into a slot of its own
                    std::vector<orcd_internal_host_chunk> list;
                    for (size_t j = 0; j < n_chunks && list.size() < 64; j++)
                        if (!chunks[j].original && chunks[j].len) list.push_back({bytes + chunks[j].at, chunks[j].len, 1u << 16});
                    std::vector<uint8_t> slots(list.size() << 16);
                    std::vector<uint32_t> sizes(list.size());
                    if (!list.empty() && orcd_internal_inflate_list(orcd_file_compression(f), list.data(), list.size(),
                                                                    slots.data(), 1u << 16, sizes.data()) == ORCD_OK)
                        for (size_t j = 0; j < list.size(); j++)
                            if (sizes[j] != 0xFFFFFFFFu) {
                                if (sizes[j] > (1u << 16)) return 94;
                                if (sizes[j]) checksum += slots[(j << 16) + sizes[j] - 1];
                            }
                }
                if (way == 0) {
                    if (orcd_internal_column_inflate_host(f, raw) == ORCD_OK) {
                        uint64_t off, n;
                        for (int k = 0; k < 8; k++)
                            if (orcd_internal_column_plain_range(raw, k, &off, &n) == ORCD_OK) checksum += off + n;
                    }
                } else {
                    std::vector<uint64_t> plain_off(n_streams, 0), plain_len(n_streams, 0), chunk_off(n_chunks, 0);
                    uint64_t at = 0;
                    for (size_t j = 0; j < n_chunks; j++) {
                        const uint32_t s = chunks[j].stream;
                        if (j == 0 || chunks[j - 1].stream != s) plain_off[s] = at;
                        chunk_off[j] = plain_len[s];
                        const uint64_t size = chunks[j].original ? chunks[j].len : (uint64_t)chunks[j].len * 3 + (j % 5);
                        plain_len[s] += size;
                        at += size;
                    }
                    if (orcd_internal_column_set_plain(raw, plain_off.data(), plain_len.data(), chunk_off.data()) != ORCD_OK) return 96;
                    for (int k = 0; k < 8; k++) {
                        uint64_t off, n;
                        if (orcd_internal_column_plain_range(raw, k, &off, &n) == ORCD_OK) checksum += off + n;
                        const uint64_t *entries;
                        size_t n_entries;
                        if (orcd_column_entry_points(f, raw, k, &entries, &n_entries) == ORCD_OK && n_entries) {
                            checksum += entries[0] + entries[n_entries - 1];
                            if (orcd_internal_column_plain_range(raw, k, &off, &n) == ORCD_OK && entries[n_entries - 1] > n) return 95;
                        }
                    }
                }
                orcd_column_free(raw);
            }
            // the handle without bytes: region, stream places, a positioned read of each place
            orcd_column *bare = nullptr;
            if (orcd_stripe_column_open(f, stripe, ft, column, &bare) != ORCD_OK) continue;
            uint64_t off = 0, n = 0;
            if (orcd_column_region(bare, &off, &n) == ORCD_OK) checksum += off + n;
            for (int k = 0; k < 8; k++) {
                if (orcd_column_stream_range(bare, k, &off, &n) != ORCD_OK || n == 0 || n > (1u << 20)) continue;
                std::vector<uint8_t> got(n);
                if (orcd_file_read_at(f, si.offset + off, n, got.data()) == ORCD_OK) checksum += got[0] + got[n - 1];
                const uint64_t *entries;
                size_t n_entries;
                if (orcd_column_entry_points(f, bare, k, &entries, &n_entries) == ORCD_OK && n_entries)
                    checksum += entries[0] + entries[n_entries - 1];
            }
            const uint8_t *none;
            size_t none_len;
            if (orcd_column_get_stream(bare, 1, &none, &none_len) == ORCD_OK) return 99;  // it has no bytes
            orcd_column_free(bare);
        }
        orcd_stripe_footer_free(ft);
    }
    orcd_file_close(f);
    return ORCD_OK;
}

// ---- hand-made images whose lengths are chosen so that their sums wrap around 64 bits --------
static void put_varint(std::vector<uint8_t> &b, uint64_t v) {
    while (v >= 0x80) {
        b.push_back((uint8_t)(v | 0x80));
        v >>= 7;
    }
    b.push_back((uint8_t)v);
}
static void put_u(std::vector<uint8_t> &b, uint32_t field, uint64_t v) {
    put_varint(b, (uint64_t)field << 3);
    put_varint(b, v);
}
static void put_msg(std::vector<uint8_t> &b, uint32_t field, const std::vector<uint8_t> &m) {
    put_varint(b, ((uint64_t)field << 3) | 2);
    put_varint(b, m.size());
    b.insert(b.end(), m.begin(), m.end());
}

struct Crafted {
    uint64_t stripe_offset = 0, index_length = 0, data_length = 24;
    uint64_t stream_a = 16, stream_b = 8;   // DATA and LENGTH of column 1
    uint64_t extra_metadata_length = 0;     // PostScript.metadataLength beyond what is there
    uint64_t footer_length_override = 0;
};

static std::vector<uint8_t> craft(const Crafted &c) {
    std::vector<uint8_t> image(24, 0x5a);   // the stripe's data area
    std::vector<uint8_t> sf, st;
    put_u(st, 1, 1); put_u(st, 2, 1); put_u(st, 3, c.stream_a);
    put_msg(sf, 1, st);
    st.clear();
    put_u(st, 1, 2); put_u(st, 2, 1); put_u(st, 3, c.stream_b);
    put_msg(sf, 1, st);
    for (int k = 0; k < 2; k++) {
        std::vector<uint8_t> enc;
        put_u(enc, 1, 2);
        put_msg(sf, 2, enc);
    }
    image.insert(image.end(), sf.begin(), sf.end());
    std::vector<uint8_t> footer, si, ty;
    put_u(si, 1, c.stripe_offset); put_u(si, 2, c.index_length); put_u(si, 3, c.data_length);
    put_u(si, 4, sf.size()); put_u(si, 5, 5);
    put_msg(footer, 3, si);
    put_u(ty, 1, 12); put_u(ty, 2, 1);
    put_msg(footer, 4, ty);
    ty.clear();
    put_u(ty, 1, 7);
    put_msg(footer, 4, ty);
    put_u(footer, 6, 5);
    image.insert(image.end(), footer.begin(), footer.end());
    std::vector<uint8_t> ps;
    put_u(ps, 1, c.footer_length_override ? c.footer_length_override : footer.size());
    put_u(ps, 2, 0);
    put_u(ps, 5, c.extra_metadata_length);
    image.insert(image.end(), ps.begin(), ps.end());
    image.push_back((uint8_t)ps.size());
    return image;
}

// 0 when every crafted image is handled (opened or refused) the way it should be
static int crafted_images() {
    Crafted ok;
    if (walk(craft(ok)) != ORCD_OK) return 10;            // the builder itself makes a readable file
    {   // footer_length + metadata_length wraps: the footer would start before the tail buffer
        Crafted c;
        c.footer_length_override = 1000;
        c.extra_metadata_length = 0ull - 500;
        if (walk(craft(c)) == ORCD_OK) return 11;
        std::vector<uint8_t> big = craft(c);
        big.insert(big.begin(), 600, 0);
        if (walk(big) == ORCD_OK) return 12;
    }
    {   // the lengths of a column's streams wrap: the column buffer would be 8 bytes, the second
        // stream "2^64 - 8 bytes at offset 16" of it
        Crafted c;
        c.stream_b = 0ull - 8;
        std::vector<uint8_t> image = craft(c);
        orcd_file *f = nullptr;
        if (orcd_file_open_memory(image.data(), image.size(), &f) != ORCD_OK) return 13;
        orcd_stripe_footer *ft = nullptr;
        if (orcd_read_stripe_footer(f, 0, &ft) == ORCD_OK) {   // (refusing the footer is fine too)
            orcd_column *col = nullptr;
            if (orcd_read_stripe_column(f, 0, ft, 1, &col) == ORCD_OK) {
                const uint8_t *bytes;
                size_t len;
                if (orcd_column_get_stream(col, 2, &bytes, &len) == ORCD_OK) return 14;
                orcd_column_free(col);
            }
            orcd_column *bare = nullptr;
            if (orcd_stripe_column_open(f, 0, ft, 1, &bare) == ORCD_OK) {
                uint64_t off, n;
                if (orcd_column_region(bare, &off, &n) == ORCD_OK) return 15;
                if (orcd_column_stream_range(bare, 2, &off, &n) == ORCD_OK) return 16;
                orcd_column_free(bare);
            }
            orcd_stripe_footer_free(ft);
        }
        orcd_file_close(f);
        walk(image);
    }
    {   // offset + index_length + data_length of a stripe wraps to a small number
        Crafted c;
        c.stripe_offset = 0ull - 10;
        c.index_length = 5;
        c.data_length = 29;
        std::vector<uint8_t> image = craft(c);
        orcd_file *f = nullptr;
        if (orcd_file_open_memory(image.data(), image.size(), &f) != ORCD_OK) return 17;
        orcd_stripe_footer *ft = nullptr;
        if (orcd_read_stripe_footer(f, 0, &ft) == ORCD_OK) return 18;
        orcd_file_close(f);
    }
    return 0;
}

int main(int argc, char **argv) {
    const int crafted = crafted_images();
    if (crafted) {
        printf("crafted image case %d\n", crafted);
        return crafted;
    }
    int opened = 0, refused = 0;
    for (int a = 1; a < argc; a++) {
        FILE *fp = fopen(argv[a], "rb");
        if (!fp) return 2;
        std::vector<uint8_t> image;
        uint8_t buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, fp)) > 0) image.insert(image.end(), buf, buf + got);
        fclose(fp);
        if (walk(image) != ORCD_OK) return 3;  // the undamaged file must open
        const size_t n = image.size();
        for (int c = 0; c < 400; c++) {
            std::vector<uint8_t> m = image;
            switch (c % 5) {
            case 0:
                for (int k = 0; k < 1 + (int)(rnd() % 5); k++) m[rnd() % n] = (uint8_t)rnd();
                break;
            case 1:
                for (int k = 0; k < 1 + (int)(rnd() % 5); k++) m[n - 1 - rnd() % (n < 600 ? n : 600)] = (uint8_t)rnd();
                break;
            case 2:
                m.resize(rnd() % n);
                break;
            case 3: {
                const size_t cut = rnd() % (n / 2 + 1), tail = 1 + rnd() % (n < 400 ? n : 400);
                std::vector<uint8_t> t(m.begin(), m.begin() + cut);
                t.insert(t.end(), image.end() - tail, image.end());
                m.swap(t);
                break;
            }
            default: {
                size_t p = n - 1 - rnd() % (n < 500 ? n : 500);
                for (size_t k = p; k < n && k < p + 11; k++) m[k] = 0xFF;
            }
            }
            if (walk(m) == ORCD_OK) opened++;
            else refused++;
        }
    }
    printf("opened %d refused %d checksum %lu\n", opened, refused, checksum);
    return 0;
}

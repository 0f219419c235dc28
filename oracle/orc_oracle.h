/*
 * orc_oracle.h -- CPU oracle for the ORC stream decode math.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C, value-at-a-time restatement of the decode
 * algorithms of the `orc-format` crate (the files of src/read/decode).  It exists so that tests can
 * check the HIP decoder bit-for-bit and so that bench.py can time a CPU baseline.  Nothing in
 * the shipped decoder may call it.
 *
 * Parity pin: every known-answer vector of the crate's own unit tests
 * (src/read/decode/rle_v2.rs:764-815, src/read/decode/boolean_rle.rs:164-208) and the decoded
 * columns of its integration tests (tests/it/main.rs:26-316) are asserted against this oracle in
 * tests/test_oracle.py; inputs the crate cannot decode (non-aligned bit widths, PatchedBase,
 * multi-run byte-RLE) are pinned against the Apache ORC C++ reader through pyarrow.orc.
 *
 * Where the crate is partial or deviates from the ORC v1 specification the oracle follows the
 * specification (full bit-width table, exact byte-RLE run/literal lengths, delta sign test).
 */
#ifndef ORC_ORACLE_H
#define ORC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes: 1..7 follow the variant order of `enum Error` (src/error.rs:6-21). */
enum {
    ORC_ORACLE_OK = 0,
    ORC_ORACLE_OUT_OF_SPEC = 1,
    ORC_ORACLE_INVALID_UTF8 = 2,
    ORC_ORACLE_DECODE_FLOAT = 5,
    ORC_ORACLE_UNSUPPORTED = 100
};

/* Run kinds, numbered as the two header bits (src/read/decode/rle_v2.rs:418-429). */
enum { ORC_RUN_SHORT_REPEAT = 0, ORC_RUN_DIRECT = 1, ORC_RUN_PATCHED_BASE = 2, ORC_RUN_DELTA = 3 };

/* One decoded run header, as the run iterators expose it (rle_v2.rs:432-439, 629-636). */
typedef struct {
    uint64_t byte_offset; /* offset of the header byte in the stream */
    uint64_t byte_length; /* header + payload bytes */
    uint32_t kind;        /* ORC_RUN_* */
    uint32_t bit_width;   /* packed width in bits (bytes*8 for ShortRepeat) */
    uint32_t length;      /* number of values in the run */
    uint32_t reserved;
} orc_oracle_run;

/* SignedRleV2Iter / UnsignedRleV2Iter (rle_v2.rs:517-562, 713-758): decode exactly `num_values`
 * values from `stream`; trailing bytes are ignored, a run may overshoot `num_values` and is then
 * truncated.  `consumed` (nullable) receives the bytes read including the last, whole run. */
int orc_oracle_rlev2_i64(const uint8_t *stream, size_t len, size_t num_values, int is_signed,
                         int64_t *out, size_t *consumed);

/* Integer RLE version 1 (ORC v1 specification; not in the reference, README.md:23-25): the same
 * contract as orc_oracle_rlev2_i64 -- exactly `num_values` values, trailing bytes ignored, the
 * last run or literal group is read whole and truncated. */
int orc_oracle_rlev1_i64(const uint8_t *stream, size_t len, size_t num_values, int is_signed,
                         int64_t *out, size_t *consumed);

/* SignedRleV2RunIter / UnsignedRleV2RunIter (rle_v2.rs:480-514, 677-710): list the run headers
 * needed to cover `num_values`.  Writes at most `max_runs` entries, returns the count in `n_runs`. */
int orc_oracle_rlev2_runs(const uint8_t *stream, size_t len, size_t num_values,
                          orc_oracle_run *runs, size_t max_runs, size_t *n_runs);

/* BooleanRleRunIter expanded to bytes (boolean_rle.rs:14-63), ORC byte-RLE semantics. */
int orc_oracle_byte_rle(const uint8_t *stream, size_t len, size_t num_bytes, uint8_t *out,
                        size_t *consumed);

/* BooleanIter (boolean_rle.rs:65-158): `num_bits` booleans, most significant bit first, one
 * byte (0/1) per value in `out`.  `set_bits` (nullable) receives the number of ones. */
int orc_oracle_boolean(const uint8_t *stream, size_t len, size_t num_bits, uint8_t *out,
                       uint64_t *set_bits);

/* FloatIter<f32|f64> (float.rs:34-91): `num_values` little-endian IEEE values of `elem_size`
 * (4 or 8) bytes copied bit-for-bit. */
int orc_oracle_float(const uint8_t *stream, size_t len, size_t elem_size, size_t num_values,
                     void *out);

/* Values::next over an unsigned RLE v2 LENGTH stream (variable_length.rs:5-28 driven as in
 * tests/it/deserialize.rs:106-140): offsets[0..num_values] (exclusive prefix sum, offsets[n] =
 * total bytes), the value bytes copied to `bytes_out` (nullable), UTF-8 checked per value when
 * `check_utf8` is non-zero. */
int orc_oracle_string_direct(const uint8_t *length_stream, size_t length_len, const uint8_t *data,
                             size_t data_len, size_t num_values, int check_utf8, int64_t *offsets,
                             uint8_t *bytes_out);

/* Dictionary gather (tests/it/deserialize.rs:142-185): out string i = dictionary entry
 * indices[i]; offsets_out has num_values + 1 entries. */
int orc_oracle_string_dictionary(const int64_t *dict_offsets, const uint8_t *dict_bytes,
                                 size_t dict_size, const int64_t *indices, size_t num_values,
                                 int64_t *offsets_out, uint8_t *bytes_out);

/* Primitives exposed for the known-answer tests (rle_v2.rs:71-123). */
int64_t orc_oracle_zigzag(uint64_t z);
uint64_t orc_oracle_unpack(const uint8_t *bytes, uint32_t num_bits, size_t index);
int orc_oracle_unsigned_varint(const uint8_t *bytes, size_t len, uint64_t *value, size_t *used);

#ifdef __cplusplus
}
#endif
#endif

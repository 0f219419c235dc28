"""A writer of synthetic, uncompressed ORC files with one string column (struct<s:string>,
DIRECT_V2): what BASELINE's string configuration looks like as a file, stripe by stripe -- DATA
bytes plus an unsigned RLE v2 LENGTH stream of Direct runs of 512 lengths -- so that the whole path
from `read_metadata` to the decoded column can be timed and checked against pyarrow.  Test and
bench infrastructure; nothing in the product path uses it.

Layout (ORC specification v1; the reference reads it in src/read/mod.rs:46-159):
    "ORC" | stripe 0 | stripe 1 | ... | Footer | PostScript | len(PostScript)
    stripe = DATA stream | LENGTH stream | StripeFooter        (no index streams: rowIndexStride 0)
"""
import numpy as np


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _u(field, v):
    return _varint(field << 3) + _varint(v)


def _msg(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def length_stream(lengths, width):
    """Unsigned RLE v2: Direct runs of 512 values of `width` bits (the last run shorter)."""
    codes = {1: 0, 2: 1, 3: 2, 4: 3, 5: 4, 6: 5, 7: 6, 8: 7, 16: 15}
    code = codes[width]
    lengths = np.asarray(lengths, dtype=np.uint64)
    assert int(lengths.max(initial=0)) < (1 << width)
    def runs(chunk_values, n):
        # (k, n) values -> k runs of n values each, header included
        k = chunk_values.shape[0]
        bits = ((chunk_values[:, :, None] >> np.arange(width - 1, -1, -1, dtype=np.uint64)) & 1).astype(np.uint8)
        packed = np.packbits(bits.reshape(k, n * width), axis=1)
        header = np.array([0x40 | (code << 1) | ((n - 1) >> 8), (n - 1) & 0xFF], dtype=np.uint8)
        return np.concatenate([np.broadcast_to(header, (k, 2)), packed], axis=1).tobytes()
    full = lengths.size // 512
    out = b""
    for beg in range(0, full, 4096):  # (bounded temporaries)
        part = lengths[beg * 512:min(full, beg + 4096) * 512]
        out += runs(part.reshape(-1, 512), 512)
    rest = lengths[full * 512:]
    if rest.size:
        out += runs(rest.reshape(1, -1), rest.size)
    return out


def chunked(buf, block):
    """ORC compression chunks of at most `block` plain bytes each: a 3-byte little-endian header
    (length << 1 | stored-as-is) and raw deflate, or the bytes themselves where that is no
    shorter (src/read/decompress/mod.rs:9-17)."""
    import zlib
    out = bytearray()
    for beg in range(0, len(buf), block):
        plain = bytes(buf[beg:beg + block])
        z = zlib.compressobj(6, zlib.DEFLATED, -15)
        packed = z.compress(plain) + z.flush()
        original = len(packed) >= len(plain)
        body = plain if original else packed
        h = (len(body) << 1) | (1 if original else 0)
        out += bytes([h & 0xFF, (h >> 8) & 0xFF, (h >> 16) & 0xFF]) + body
    return bytes(out)


def make_stripe(rng, n_strings, min_len=8, max_len=56, width=6, compression_block=0):
    """One stripe's bytes and what it holds: (stripe bytes, lengths uint64[n], data uint8[...],
    data_length, footer_length).  Printable ASCII, lengths U[min_len, max_len]."""
    lengths = rng.integers(min_len, max_len + 1, size=n_strings, dtype=np.uint64)
    total = int(lengths.sum())
    data = rng.integers(0x20, 0x7F, size=total, dtype=np.uint8)
    length_bytes = length_stream(lengths, width)
    data_bytes = data.tobytes()
    if compression_block:
        data_bytes = chunked(data_bytes, compression_block)
        length_bytes = chunked(length_bytes, compression_block)
    footer = (_msg(1, _u(1, 1) + _u(2, 1) + _u(3, len(data_bytes))) +    # DATA of column 1
              _msg(1, _u(1, 2) + _u(2, 1) + _u(3, len(length_bytes))) +  # LENGTH of column 1
              _msg(2, _u(1, 0)) +                                        # column 0: DIRECT
              _msg(2, _u(1, 2)))                                         # column 1: DIRECT_V2
    if compression_block:
        footer = chunked(footer, compression_block)
    stripe = data_bytes + length_bytes + footer
    return stripe, lengths, data, len(data_bytes) + len(length_bytes), len(footer)


def write_string_file(path, n_stripes, strings_per_stripe, seed=46, distinct=2, **kw):
    """Writes the file; stripe i holds the strings of template i % distinct (making a few
    templates and repeating them keeps the generation of a multi-gigabyte file short).  Returns the
    templates as a list of (lengths, data) and the file size.  compression_block=N: CompressionKind
    ZLIB with chunks of N plain bytes (streams, stripe footers and the file footer)."""
    rng = np.random.default_rng(seed)
    block = kw.get("compression_block", 0)
    templates = [make_stripe(rng, strings_per_stripe, **kw) for _ in range(min(distinct, n_stripes))]
    infos = []
    pos = 3
    with open(path, "wb") as f:
        f.write(b"ORC")
        for i in range(n_stripes):
            stripe, _, _, data_length, footer_length = templates[i % len(templates)]
            f.write(stripe)
            infos.append(_msg(3, _u(1, pos) + _u(2, 0) + _u(3, data_length) + _u(4, footer_length) +
                              _u(5, strings_per_stripe)))
            pos += len(stripe)
        footer = (_u(1, 3) + _u(2, pos - 3) + b"".join(infos) +
                  _msg(4, _u(1, 12) + _msg(2, _varint(1)) + _msg(3, b"s")) +   # struct<s: ...>
                  _msg(4, _u(1, 7)) +                                          # string
                  _u(6, n_stripes * strings_per_stripe) + _u(8, 0))
        if block:
            footer = chunked(footer, block)
        ps = (_u(1, len(footer)) + _u(2, 1 if block else 0) + (_u(3, max(block, 64)) if block else b"") +
              _msg(4, _varint(0) + _varint(12)) + _u(5, 0) + _u(6, 1) + _msg(8000, b"ORC"))
        f.write(footer + ps + bytes([len(ps)]))
        size = pos + len(footer) + len(ps) + 1
    return [(t[1], t[2]) for t in templates], size

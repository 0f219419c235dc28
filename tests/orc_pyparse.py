"""Minimal pure-Python reader of the ORC file tail and stripe footers (test helper).

Independent of the C++ host library, so that tests can (a) hand raw stream bytes to the oracle
and (b) cross-check the host library's own parsing.  Handles uncompressed and zlib-compressed files (python's zlib, raw deflate chunks).
Field numbers are those of the ORC protobuf schema (reference: src/proto.rs:89-278).
"""
from collections import namedtuple

PRESENT, DATA, LENGTH, DICTIONARY_DATA, DICTIONARY_COUNT, SECONDARY, ROW_INDEX, BLOOM_FILTER = range(8)

Stream = namedtuple("Stream", "kind column length offset")  # offset: absolute file offset
Stripe = namedtuple("Stripe", "offset index_length data_length footer_length number_of_rows")


def _varint(buf, pos):
    v = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if b < 0x80:
            return v, pos
        shift += 7


def fields(buf):
    """Yields (field_number, wire_type, value) of one protobuf message."""
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield fno, wt, v


def _packed_varints(v, wt):
    if wt == 0:
        return [v]
    out = []
    pos = 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(x)
    return out


def inflate_chunks(buf):
    """ORC compression chunks (decompress/mod.rs:9-17, 60-80) -> plain bytes and the chunk table
    [(offset of the chunk header in buf, offset of its bytes in the result)]."""
    import zlib
    out, table, pos = bytearray(), [], 0
    while pos < len(buf):
        h = buf[pos] | (buf[pos + 1] << 8) | (buf[pos + 2] << 16)
        table.append((pos, len(out)))
        n = h >> 1
        chunk = bytes(buf[pos + 3:pos + 3 + n])
        assert len(chunk) == n
        out += chunk if h & 1 else zlib.decompressobj(-15).decompress(chunk)
        pos += 3 + n
    return bytes(out), table


class OrcFile:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.bytes = f.read()
        b = self.bytes
        ps_len = b[-1]
        ps = b[-1 - ps_len:-1]
        self.footer_length = self.metadata_length = 0
        self.compression = 0
        for fno, wt, v in fields(ps):
            if fno == 1:
                self.footer_length = v
            elif fno == 2:
                self.compression = v
            elif fno == 5:
                self.metadata_length = v
        assert self.compression in (0, 1), "test helper reads uncompressed and zlib files only"
        footer = self._plain(b[-1 - ps_len - self.footer_length:-1 - ps_len])
        self.stripes = []
        self.types = []
        self.number_of_rows = 0
        self.row_index_stride = 0
        for fno, wt, v in fields(footer):
            if fno == 3:
                d = {1: 0, 2: 0, 3: 0, 4: 0, 5: 0}
                for f2, _, v2 in fields(v):
                    d[f2] = v2
                self.stripes.append(Stripe(d[1], d[2], d[3], d[4], d[5]))
            elif fno == 4:
                kind, subtypes = 0, []
                for f2, w2, v2 in fields(v):
                    if f2 == 1:
                        kind = v2
                    elif f2 == 2:
                        subtypes += _packed_varints(v2, w2)
                self.types.append((kind, subtypes))
            elif fno == 6:
                self.number_of_rows = v
            elif fno == 8:
                self.row_index_stride = v

    def _plain(self, buf):
        return inflate_chunks(buf)[0] if self.compression else buf

    def stripe_footer(self, stripe):
        s = self.stripes[stripe]
        start = s.offset + s.index_length + s.data_length
        buf = self._plain(self.bytes[start:start + s.footer_length])
        streams, encodings = [], []
        offset = s.offset
        for fno, wt, v in fields(buf):
            if fno == 1:
                d = {1: 0, 2: 0, 3: 0}
                for f2, _, v2 in fields(v):
                    d[f2] = v2
                streams.append(Stream(d[1], d[2], d[3], offset))
                offset += d[3]
            elif fno == 2:
                d = {1: 0, 2: None}
                for f2, _, v2 in fields(v):
                    d[f2] = v2
                encodings.append((d[1], d[2]))
        return streams, encodings

    def stream(self, stripe, column, kind):
        """Bytes of one stream (decompressed), or None when the column has no such stream."""
        streams, _ = self.stripe_footer(stripe)
        for st in streams:
            if st.column == column and st.kind == kind:
                return self._plain(self.bytes[st.offset:st.offset + st.length])
        return None

    def chunk_table(self, stripe, column, kind):
        """The chunk table of a compressed stream (see inflate_chunks)."""
        streams, _ = self.stripe_footer(stripe)
        for st in streams:
            if st.column == column and st.kind == kind:
                return inflate_chunks(self.bytes[st.offset:st.offset + st.length])[1]
        return None

    def row_index_positions(self, stripe, column):
        """List (one per row group) of the position lists of the column's ROW_INDEX stream."""
        buf = self.stream(stripe, column, ROW_INDEX)
        if buf is None:
            return None
        groups = []
        for fno, wt, v in fields(buf):
            if fno == 1:
                pos = []
                for f2, w2, v2 in fields(v):
                    if f2 == 1:
                        pos += _packed_varints(v2, w2)
                groups.append(pos)
        return groups

"""Test-side writers for ORC RLE v2 / byte-RLE streams (numpy; ORC v1 specification layouts).

Used to build synthetic streams with chosen sub-encodings, widths and run lengths -- including
the ones no writer library would choose -- for parity tests of the decoder against the oracle.
"""
import numpy as np

WIDTHS = list(range(1, 25)) + [26, 28, 30, 32, 40, 48, 56, 64]
_CODE = {w: i for i, w in enumerate(WIDTHS)}


def width_code(w):
    return _CODE[w]


def pack_bits(values, width):
    """values (uint64 array) packed MSB-first, `width` bits each, zero-padded to a whole byte."""
    v = np.asarray(values, dtype=np.uint64)
    if width == 0 or v.size == 0:
        return b""
    shifts = np.arange(width - 1, -1, -1, dtype=np.uint64)
    bits = ((v[:, None] >> shifts[None, :]) & np.uint64(1)).astype(np.uint8)
    return np.packbits(bits.reshape(-1)).tobytes()


def zigzag_encode(v):
    v = np.asarray(v, dtype=np.int64)
    return ((v << np.int64(1)) ^ (v >> np.int64(63))).view(np.uint64)


def varint(u):
    u = int(u) & (2**64 - 1)
    out = bytearray()
    while True:
        b = u & 0x7F
        u >>= 7
        if u:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def svarint(i):
    i = int(i)
    return varint(((i << 1) ^ (i >> 63)) & (2**64 - 1))


def short_repeat(value, count, signed):
    """value repeated count (3..10) times; value is the logical (signed) value."""
    assert 3 <= count <= 10
    u = int(zigzag_encode([value])[0]) if signed else int(value) & (2**64 - 1)
    nbytes = max(1, (u.bit_length() + 7) // 8)
    return bytes([((nbytes - 1) << 3) | (count - 3)]) + u.to_bytes(nbytes, "big")


def direct(values, width, signed):
    """Direct run; values are logical values that must fit `width` bits after zigzag."""
    n = len(values)
    assert 1 <= n <= 512
    u = zigzag_encode(values) if signed else np.asarray(values).astype(np.uint64)
    h = (1 << 14) | (width_code(width) << 9) | (n - 1)
    return bytes([h >> 8, h & 0xFF]) + pack_bits(u, width)


def delta(base, delta_base, deltas, width, signed):
    """Delta run: base, base + delta_base, then +/- |deltas| (width 0: fixed delta, then
    `deltas` is an int count of further values)."""
    if width == 0:
        n = 2 + int(deltas)
        payload = b""
        code = 0
    else:
        n = 2 + len(deltas)
        payload = pack_bits(np.asarray(deltas, dtype=np.uint64), width)
        code = width_code(width)
        assert code != 0, "width 1 cannot be written for a Delta run"
    assert 1 <= n <= 512
    h = (3 << 14) | (code << 9) | (n - 1)
    first = svarint(base) if signed else varint(base)
    return bytes([h >> 8, h & 0xFF]) + first + svarint(delta_base) + payload


def patched_base(base, values, width, patches, patch_width, gap_width):
    """PatchedBase run.  `values` are the reduced (value - base) low `width` bits; `patches` is
    a list of (gap, patch) entries as stored (gap < 2**gap_width)."""
    n = len(values)
    assert 1 <= n <= 512 and 1 <= len(patches) <= 31 and 1 <= gap_width <= 8
    mag = abs(int(base))
    base_bytes = max(1, (mag.bit_length() + 1 + 7) // 8)
    b = mag | ((1 << (base_bytes * 8 - 1)) if base < 0 else 0)
    h = (2 << 14) | (width_code(width) << 9) | (n - 1)
    third = ((base_bytes - 1) << 5) | width_code(patch_width)
    fourth = ((gap_width - 1) << 5) | len(patches)
    total = patch_width + gap_width
    cfb = next(w for w in WIDTHS if w >= total)
    entries = [(g << patch_width) | p for g, p in patches]
    return (bytes([h >> 8, h & 0xFF, third, fourth]) + b.to_bytes(base_bytes, "big") +
            pack_bits(np.asarray(values, dtype=np.uint64), width) +
            pack_bits(np.asarray(entries, dtype=np.uint64), cfb))


def rlev1_run(base, delta, count, signed):
    """Integer RLE v1 run: count (3..130) values base, base + delta, ... (delta -128..127)."""
    assert 3 <= count <= 130 and -128 <= delta <= 127
    return bytes([count - 3, delta & 0xFF]) + (svarint(base) if signed else varint(base))


def rlev1_literals(values, signed):
    """Integer RLE v1 literal group of 1..128 values."""
    assert 1 <= len(values) <= 128
    return bytes([256 - len(values)]) + b"".join(svarint(v) if signed else varint(v) for v in values)


def rlev1(values, signed):
    """Integer RLE v1 encoding of `values` the way the ORC writers do it: runs of at least three
    values with a constant delta in -128..127, everything else as literal groups.  Returns the
    bytes and the start offset of every run / literal group."""
    v = [int(x) for x in values]
    out, starts, lit = bytearray(), [], []

    def flush():
        while lit:
            starts.append(len(out))
            out.extend(rlev1_literals(lit[:128], signed))
            del lit[:128]
    i, n = 0, len(v)
    while i < n:
        j = i + 1
        if j < n and -128 <= v[j] - v[i] <= 127:
            d = v[j] - v[i]
            while j + 1 < n and v[j + 1] - v[j] == d and j + 1 - i < 130:
                j += 1
        if j - i + 1 >= 3 and j < n and -128 <= v[i + 1] - v[i] <= 127:
            flush()
            starts.append(len(out))
            out.extend(rlev1_run(v[i], v[i + 1] - v[i], j - i + 1, signed))
            i = j + 1
        else:
            lit.append(v[i])
            i += 1
    flush()
    return bytes(out), starts


def byte_rle(data, rng=None, max_literal=128):
    """Byte-RLE encoding of `data` (bytes): runs of 3..130 equal bytes, literals of 1..128."""
    data = bytes(data)
    out = bytearray()
    i, n = 0, len(data)
    lit = bytearray()

    def flush():
        while lit:
            chunk = lit[:max_literal]
            out.append(256 - len(chunk))
            out.extend(chunk)
            del lit[:max_literal]

    while i < n:
        j = i
        while j < n and data[j] == data[i] and j - i < 130:
            j += 1
        if j - i >= 3:
            flush()
            out.append(j - i - 3)
            out.append(data[i])
            i = j
        else:
            lit.append(data[i])
            i += 1
    flush()
    return bytes(out)


def random_stream(rng, n_runs, signed, kinds=(0, 1, 2, 3), widths=None, max_len=512, min_len=1):
    """A random RLE v2 stream.  Returns (bytes, expected int64 values, run start offsets)."""
    widths = widths or WIDTHS
    chunks, expect, starts = [], [], []
    pos = 0
    for _ in range(n_runs):
        kind = int(rng.choice(kinds))
        w = int(rng.choice(widths))
        if kind == 0:
            cnt = int(rng.integers(3, 11))
            nb = int(rng.integers(1, 9))
            u = int(rng.integers(0, 2**(8 * nb) - 1, dtype=np.uint64, endpoint=True))
            if signed:
                v = (u >> 1) ^ -(u & 1)
                buf = short_repeat(v, cnt, True)
                vals = np.full(cnt, v, dtype=np.int64)
            else:
                buf = short_repeat(u, cnt, False)
                vals = np.full(cnt, u, dtype=np.uint64).view(np.int64)
        elif kind == 1:
            L = int(rng.integers(min_len, max_len + 1))
            u = rng.integers(0, 2**w - 1, size=L, dtype=np.uint64, endpoint=True)
            if signed:
                vals = ((u >> np.uint64(1)).view(np.int64)) ^ -((u & np.uint64(1)).view(np.int64))
                buf = direct(vals, w, True)
            else:
                vals = u.view(np.int64)
                buf = direct(u, w, False)
        elif kind == 3:
            L = int(rng.integers(max(min_len, 2), max_len + 1))
            neg = bool(rng.integers(0, 2))
            base = int(rng.integers(-2**40, 2**40)) if signed else int(rng.integers(0, 2**41))
            if w == 1 or rng.integers(0, 4) == 0:
                db = int(rng.integers(-1000, 1000))
                buf = delta(base, db, L - 2, 0, signed)
                vals = (np.int64(base) + np.arange(L, dtype=np.int64) * np.int64(db))
            else:
                d = rng.integers(0, 2**w - 1, size=L - 2, dtype=np.uint64, endpoint=True)
                db = int(rng.integers(1, 1000)) * (-1 if neg else 1)
                buf = delta(base, db, d, w, signed)
                steps = np.concatenate([[np.uint64(base & (2**64 - 1))],
                                        [np.uint64(db & (2**64 - 1))],
                                        (np.uint64(0) - d) if neg else d]).astype(np.uint64)
                with np.errstate(over="ignore"):
                    vals = np.cumsum(steps, dtype=np.uint64).view(np.int64)
        else:
            L = int(rng.integers(max(min_len, 2), max_len + 1))
            w = min(w, 56)
            pw = int(rng.choice([x for x in WIDTHS if x + w <= 64 and x + 8 <= 64]))
            gw = int(rng.integers(1, 9))
            base = int(rng.integers(-2**30, 2**30))
            low = rng.integers(0, 2**w - 1, size=L, dtype=np.uint64, endpoint=True)
            full = low.copy()
            patches = []
            p = 0
            last = 0
            for _k in range(int(rng.integers(1, 6))):
                gap = int(rng.integers(0, min(2**gw, 255)))
                p = last + gap
                if p >= L or (patches and gap == 0):
                    break
                pv = int(rng.integers(1, 2**pw - 1, dtype=np.uint64, endpoint=True))
                patches.append((gap, pv))
                full[p] = np.uint64(int(low[p]) | (pv << w))
                last = p
            if not patches:
                patches = [(0, 1)]
                full[0] = np.uint64(int(low[0]) | (1 << w))
            buf = patched_base(base, low, w, patches, pw, gw)
            with np.errstate(over="ignore"):
                vals = (full + np.uint64(base & (2**64 - 1))).view(np.int64)
        starts.append(pos)
        pos += len(buf)
        chunks.append(buf)
        expect.append(np.asarray(vals, dtype=np.int64))
    return b"".join(chunks), np.concatenate(expect), starts

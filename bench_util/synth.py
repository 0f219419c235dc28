"""Synthetic ORC streams built directly in GPU memory with torch tensor ops (bench / test
plumbing -- not part of the decoder).

Each generator returns the stream bytes, the run-header entry points a writer would record in
the ROW_INDEX (one per `runs_per_entry` runs) and the values the stream encodes, all as CUDA
tensors.  Streams are laid out exactly as the ORC v1 specification says; the small-size tests
check them against the CPU oracle.
"""
import torch

MASK64 = (1 << 64) - 1


def _splitmix64(seed, start, count, device):
    """splitmix64 of the counters start+1 .. start+count as int64 bit patterns."""
    idx = torch.arange(start + 1, start + count + 1, dtype=torch.int64, device=device)
    z = idx * _i64(0x9E3779B97F4A7C15) + _i64(seed)
    z = (z ^ _lsr(z, 30)) * _i64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _i64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def _i64(u):
    u &= MASK64
    return u - (1 << 64) if u >= (1 << 63) else u


def _lsr(x, k):
    """logical shift right of int64 bit patterns"""
    return (x >> k) & ((1 << (64 - k)) - 1)


def pack_bits(values, width):
    """values: int64 tensor [rows, n] of non-negative numbers < 2**width (width <= 56).
    Returns uint8 [rows, ceil(n*width/8)]: every row packed MSB-first, zero padded."""
    rows, n = values.shape
    nbytes = (n * width + 7) // 8
    # pad so that the "next value" gather below stays in range
    v = torch.nn.functional.pad(values, (0, 2))
    j = torch.arange(nbytes, device=values.device, dtype=torch.int64)
    bit = j * 8
    i = bit // width                      # first value overlapping byte j
    off = bit - i * width                 # bits of value i already consumed
    out = torch.zeros((rows, nbytes), dtype=torch.int64, device=values.device)
    # a byte overlaps at most ceil(8/width)+1 values; accumulate them MSB-first
    remaining = torch.full_like(j, 8)
    cur_i, cur_off = i.clone(), off.clone()
    for _ in range(8 // width + 2):
        take = torch.minimum(remaining, width - cur_off)          # bits taken from value cur_i
        vi = v[:, cur_i.clamp(max=n + 1)]                          # [rows, nbytes]
        chunk = (vi >> (width - cur_off - take)) & ((torch.ones_like(take) << take) - 1)
        out = (out << take) | chunk
        remaining = remaining - take
        cur_i = cur_i + 1
        cur_off = torch.zeros_like(cur_off)
        if int(remaining.max()) == 0:
            break
    return out.to(torch.uint8)


def direct_w17(n_runs, seed=42, runs_per_entry=20, device="cuda", chunk_runs=1 << 15):
    """Signed int64 column, every run Direct, 512 values of width 17 (header 61 ff): zigzag value
    u ~ U[0, 2^17) with bit 16 forced so that the width is exactly 17.
    Returns (stream uint8 [n_runs * 1090], entries int64, values int64 [n_runs * 512])."""
    run_bytes = 2 + 1088
    stream = torch.empty((n_runs, run_bytes), dtype=torch.uint8, device=device)
    values = torch.empty((n_runs, 512), dtype=torch.int64, device=device)
    stream[:, 0] = 0x61
    stream[:, 1] = 0xFF
    for r0 in range(0, n_runs, chunk_runs):
        r1 = min(n_runs, r0 + chunk_runs)
        u = _splitmix64(seed, r0 * 512, (r1 - r0) * 512, device)
        u = (u & 0xFFFF) | 0x10000
        u = u.view(r1 - r0, 512)
        stream[r0:r1, 2:] = pack_bits(u, 17)
        values[r0:r1] = (u >> 1) ^ -(u & 1)                        # zigzag decode
    entries = torch.arange(0, n_runs, runs_per_entry, dtype=torch.int64, device=device) * run_bytes
    return stream.view(-1), entries, values.view(-1)


def delta_monotonic(n_runs, width=10, seed=43, runs_per_entry=20, device="cuda",
                    chunk_runs=1 << 15, t0=1_600_000_000_000):
    """Monotonic signed int64 column (timestamp-like): t[i] = t[i-1] + d, d ~ U[1, 1000] (seed),
    every run Delta with 512 values and `width`-bit deltas (10: not byte aligned, 16: aligned).
    The second value of each run uses d in [64, 1000] so that every header has the same size
    (2 header bytes, 6-byte zigzag base, 2-byte delta base).
    Returns (stream, entries, values)."""
    assert width >= 10
    payload = (510 * width + 7) // 8
    run_bytes = 2 + 6 + 2 + payload
    stream = torch.empty((n_runs, run_bytes), dtype=torch.uint8, device=device)
    values = torch.empty((n_runs, 512), dtype=torch.int64, device=device)
    code = {10: 9, 16: 15, 24: 23, 32: 27}[width]  # (the 5-bit width code of the ORC v1 table)
    stream[:, 0] = 0xC0 | (code << 1) | 1
    stream[:, 1] = 0xFF
    base = t0
    for r0 in range(0, n_runs, chunk_runs):
        r1 = min(n_runs, r0 + chunk_runs)
        z = _splitmix64(seed, r0 * 512, (r1 - r0) * 512, device)
        d = (_lsr(z, 11) % 1000 + 1).view(r1 - r0, 512)
        d[:, 1] = d[:, 1] % 937 + 64
        t = torch.cumsum(d.view(-1), 0).view(r1 - r0, 512) + base
        base = int(t[-1, -1])
        values[r0:r1] = t
        zz = t[:, 0] << 1                                          # zigzag of a positive base
        for k in range(6):                                         # 6-byte LEB128
            stream[r0:r1, 2 + k] = (((zz >> (7 * k)) & 0x7F) | (0x80 if k < 5 else 0)).to(torch.uint8)
        zd = d[:, 1] << 1
        stream[r0:r1, 8] = ((zd & 0x7F) | 0x80).to(torch.uint8)
        stream[r0:r1, 9] = ((zd >> 7) & 0x7F).to(torch.uint8)
        stream[r0:r1, 10:] = pack_bits(d[:, 2:].contiguous(), width)
    entries = torch.arange(0, n_runs, runs_per_entry, dtype=torch.int64, device=device) * run_bytes
    return stream.view(-1), entries, values.view(-1)


def string_lengths_w6(n_runs, seed=46, runs_per_entry=20, device="cuda", chunk_runs=1 << 15):
    """Unsigned LENGTH stream of a direct-encoded string column: lengths ~ U[8, 56], Direct
    runs of 512 values of width 6.  Returns (stream, entries, lengths int64)."""
    run_bytes = 2 + 384
    stream = torch.empty((n_runs, run_bytes), dtype=torch.uint8, device=device)
    lengths = torch.empty((n_runs, 512), dtype=torch.int64, device=device)
    stream[:, 0] = 0x40 | (5 << 1) | 1
    stream[:, 1] = 0xFF
    for r0 in range(0, n_runs, chunk_runs):
        r1 = min(n_runs, r0 + chunk_runs)
        z = _splitmix64(seed, r0 * 512, (r1 - r0) * 512, device)
        ln = (_lsr(z, 11) % 49 + 8).view(r1 - r0, 512)
        lengths[r0:r1] = ln
        stream[r0:r1, 2:] = pack_bits(ln, 6)
    entries = torch.arange(0, n_runs, runs_per_entry, dtype=torch.int64, device=device) * run_bytes
    return stream.view(-1), entries, lengths.view(-1)


def byte_rle_mix(n_blocks, seed=45, run_fraction=0.3, blocks_per_entry=10, device="cuda"):
    """Byte-RLE stream (PRESENT / boolean DATA shape): a random mix of literal groups (control
    0x80 + 128 random bytes) and runs (control 0x7f: 130 copies of 0xff, "all valid").
    Returns (stream uint8, entries int64 = offsets of every `blocks_per_entry`-th control byte,
    decoded bytes uint8)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    is_run = torch.rand(n_blocks, device=device, generator=g) < run_fraction
    size_in = torch.where(is_run, 2, 129)
    size_out = torch.where(is_run, 130, 128)
    in_off = torch.cumsum(size_in, 0) - size_in
    out_off = torch.cumsum(size_out, 0) - size_out
    total_in = int(size_in.sum())
    total_out = int(size_out.sum())
    stream = torch.zeros(total_in, dtype=torch.uint8, device=device)
    decoded = torch.full((total_out,), 0xFF, dtype=torch.uint8, device=device)
    stream[in_off] = torch.where(is_run, 0x7F, 0x80).to(torch.uint8)
    stream[in_off[is_run] + 1] = 0xFF
    lit_in = in_off[~is_run] + 1
    lit_out = out_off[~is_run]
    j = torch.arange(128, device=device)
    for b0 in range(0, lit_in.numel(), 1 << 20):   # in pieces: the index tensors are 8x the data
        b1 = min(lit_in.numel(), b0 + (1 << 20))
        payload = torch.randint(0, 256, (b1 - b0, 128), dtype=torch.uint8, device=device, generator=g)
        stream[(lit_in[b0:b1, None] + j).view(-1)] = payload.view(-1)
        decoded[(lit_out[b0:b1, None] + j).view(-1)] = payload.view(-1)
    entries = in_off[::blocks_per_entry].contiguous()
    return stream, entries, decoded


def unpack_bools(packed, num_bits):
    """MSB-first bits of `packed` as one uint8 (0/1) per value."""
    sh = torch.arange(7, -1, -1, device=packed.device, dtype=torch.uint8)
    out = torch.empty(packed.numel() * 8, dtype=torch.uint8, device=packed.device)
    step = 1 << 26
    for b0 in range(0, packed.numel(), step):
        b1 = min(packed.numel(), b0 + step)
        out[b0 * 8:b1 * 8] = ((packed[b0:b1, None] >> sh) & 1).view(-1)
    return out[:num_bits]


def mixed_short_runs(n_runs, seed=44, runs_per_entry=64, device="cuda", chunk=1 << 20):
    """Low-cardinality unsigned column as a writer emits it for values U{0..15} that repeat in
    bursts: a random mix of ShortRepeat runs (3..10 copies, one value byte) and Direct runs of
    width 4 and 1..64 values.  Returns (stream uint8, entries int64, values int64, entry_values
    int64 = number of values before each entry point)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    is_sr = torch.rand(n_runs, device=device, generator=g) < 0.5
    count = torch.randint(3, 11, (n_runs,), device=device, generator=g)      # ShortRepeat copies
    dlen = torch.randint(1, 65, (n_runs,), device=device, generator=g)       # Direct values
    n_out = torch.where(is_sr, count, dlen)
    n_in = torch.where(is_sr, 2, 2 + (dlen + 1) // 2)
    in_off = torch.cumsum(n_in, 0) - n_in
    out_off = torch.cumsum(n_out, 0) - n_out
    total_in, total_out = int(n_in.sum()), int(n_out.sum())
    stream = torch.zeros(total_in, dtype=torch.uint8, device=device)
    values = torch.zeros(total_out, dtype=torch.int64, device=device)
    # headers: ShortRepeat 00 | (bytes-1)<<3 | (count-3); Direct 01 | code(4 bits)=3 <<1 | len-1 (9 bits)
    h0 = torch.where(is_sr, count - 3, 0x40 | (3 << 1) | ((dlen - 1) >> 8))
    stream[in_off] = h0.to(torch.uint8)
    sr_val = torch.randint(0, 16, (n_runs,), device=device, generator=g)
    stream[in_off + 1] = torch.where(is_sr, sr_val, (dlen - 1) & 0xFF).to(torch.uint8)
    j10 = torch.arange(10, device=device)
    j64 = torch.arange(64, device=device)
    j32 = torch.arange(32, device=device)
    for r0 in range(0, n_runs, chunk):
        r1 = min(n_runs, r0 + chunk)
        sr = is_sr[r0:r1]
        # ShortRepeat values
        o, c, v = out_off[r0:r1][sr], count[r0:r1][sr], sr_val[r0:r1][sr]
        m = j10[None, :] < c[:, None]
        values[(o[:, None] + j10)[m]] = v[:, None].expand(-1, 10)[m]
        # Direct values and payload bytes
        o, ln, io = out_off[r0:r1][~sr], dlen[r0:r1][~sr], in_off[r0:r1][~sr]
        nib = torch.randint(0, 16, (o.numel(), 64), device=device, generator=g)
        m = j64[None, :] < ln[:, None]
        nib = nib * m
        values[(o[:, None] + j64)[m]] = nib[m]
        packed = (nib[:, 0::2] << 4) | nib[:, 1::2]
        mb = j32[None, :] < ((ln + 1) // 2)[:, None]
        stream[(io[:, None] + 2 + j32)[mb]] = packed[mb].to(torch.uint8)
    entries = in_off[::runs_per_entry].contiguous()
    return stream, entries, values, out_off[::runs_per_entry].contiguous()

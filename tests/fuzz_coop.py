"""Differential fuzz of the header passes against the CPU oracle: random RLE v2 streams of every run
kind and random byte-RLE / boolean streams, with and without entry points, cut, corrupted and with garbage tails
(python3 tests/fuzz_coop.py [rounds]; not collected by pytest: a longer soak than the suite affords)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rle_encode as enc
import oracle_ctypes
from orc_format_amd import Context
from orc_format_amd.error import OrcError

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
oracle = oracle_ctypes.load()
ctx = Context(0)
rng = np.random.default_rng(12345)
bad = 0
for it in range(rounds):
    signed = bool(rng.integers(0, 2))
    kinds = [(0, 1), (0, 1, 2, 3), (0, 3), (1,), (2, 3), (0,)][int(rng.integers(0, 6))]
    n_runs = int(rng.choice([3, 70, 500, 3000, 20000]))
    max_len = int(rng.choice([3, 10, 60, 512]))
    stream, vals, starts = enc.random_stream(rng, n_runs, signed, kinds=kinds, widths=[1, 2, 4, 7, 8, 13, 24, 32], max_len=max_len)
    mode = int(rng.integers(0, 5))
    s = stream
    n = vals.size
    if mode == 1 and len(s) > 4:      # cut
        s = s[:int(rng.integers(1, len(s)))]
    elif mode == 2:                   # a flipped byte
        b = bytearray(s); k = int(rng.integers(0, len(b))); b[k] ^= 1 << int(rng.integers(0, 8)); s = bytes(b)
    elif mode == 3:                   # garbage tail, more values wanted
        s = s + bytes(rng.integers(0, 256, size=int(rng.integers(1, 3000)), dtype=np.uint8)); n = vals.size + int(rng.integers(1, 5000))
    elif mode == 4:                   # fewer values
        n = max(1, vals.size - int(rng.integers(0, vals.size)))
    entries = None
    if mode in (0, 4) and len(starts) > 8 and rng.integers(0, 2):
        step = int(rng.integers(1, max(2, len(starts) // 4)))
        entries = np.array(starts[::step], dtype=np.uint64)
    rc, ref, _ = oracle.rlev2(s, n, signed)
    try:
        out = ctx.rlev2(s, n, signed, entries)
        ok = rc == 0 and np.array_equal(out, ref[:n])
        got = 0
    except OrcError as e:
        got = e.code
        ok = got == rc
    if not ok:
        bad += 1
        print("MISMATCH round", it, "mode", mode, "kinds", kinds, "n_runs", n_runs, "max_len", max_len, "rc oracle", rc, "gpu", got, flush=True)
    if it % 50 == 49:
        print("round", it + 1, "mismatches", bad, flush=True)

def byte_rle_starts(stream):
    starts, pos = [], 0
    while pos < len(stream):
        starts.append(pos)
        c = stream[pos]
        pos += 2 if c < 128 else 1 + (256 - c)
    return starts


# byte-RLE / boolean streams the same way
for it in range(rounds):
    n_bytes = int(rng.choice([5, 200, 5000, 150000]))
    style = int(rng.integers(0, 3))
    if style == 0:
        data = bytes(rng.integers(0, 256, size=n_bytes, dtype=np.uint8))
    elif style == 1:
        data = np.packbits(rng.random(n_bytes * 8) < rng.choice([0.5, 0.9, 0.99])).tobytes()
    else:
        data = b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 200)) for _ in range(max(1, n_bytes // 60)))
    stream = enc.byte_rle(data, max_literal=int(rng.choice([3, 17, 128])))
    n = len(data) * 8
    mode = int(rng.integers(0, 5))
    s = stream
    if mode == 1 and len(s) > 2:
        s = s[:int(rng.integers(1, len(s)))]
    elif mode == 2:
        b = bytearray(s); k = int(rng.integers(0, len(b))); b[k] ^= 1 << int(rng.integers(0, 8)); s = bytes(b)
    elif mode == 3:
        n = n + int(rng.integers(1, 4000))
    elif mode == 4:
        n = max(1, n - int(rng.integers(0, n)))
    entries = None
    starts = byte_rle_starts(stream)
    if mode in (0, 4) and len(starts) > 8 and rng.integers(0, 2):
        step = int(rng.integers(1, max(2, len(starts) // 4)))
        entries = np.array(starts[::step], dtype=np.uint64)
    rc, ref, ones_ref = oracle.boolean(s, n)
    try:
        out, ones = ctx.boolean(s, n, entries)
        ok = rc == 0 and np.array_equal(out.astype(bool), ref) and ones == ones_ref
        got = 0
    except OrcError as e:
        got = e.code
        ok = got == rc
    if not ok:
        bad += 1
        print("MISMATCH boolean round", it, "mode", mode, "style", style, "bytes", n_bytes, "rc oracle", rc, "gpu", got, flush=True)
    if it % 100 == 99:
        print("boolean round", it + 1, "mismatches", bad, flush=True)
print("done:", rounds, "rounds of each,", bad, "mismatches")
sys.exit(1 if bad else 0)

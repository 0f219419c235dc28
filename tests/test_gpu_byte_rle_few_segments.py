"""Byte-RLE / boolean streams of few segments (a stripe's PRESENT stream with an entry point per row
group, or no entry points at all): a lane per segment walks the control bytes out of LDS, 2 to 64
segments to a wavefront, or from memory where the segments do not fit.  Values and verdicts are
the oracle's (boolean_rle.rs:28-62, run and literal lengths as the ORC specification has them)."""
import numpy as np
import pytest

import rle_encode as enc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from gpu_backend import GpuBackend
    return GpuBackend()


def run_starts(stream):
    starts, pos = [], 0
    while pos < len(stream):
        starts.append(pos)
        c = stream[pos]
        pos += 2 if c < 128 else 1 + (256 - c)
    return starts


def present_like(rng, n_rows, null_rate):
    bits = rng.random(n_rows) >= null_rate
    return bits, enc.byte_rle(np.packbits(bits).tobytes())


@pytest.mark.parametrize("n_seg", [1, 2, 5, 63, 64, 65, 430, 3000])
@pytest.mark.parametrize("null_rate", [0.1, 0.5])
def test_present_streams(gpu, oracle, n_seg, null_rate):
    rng = np.random.default_rng(n_seg + int(null_rate * 10))
    bits, stream = present_like(rng, 1_200_003, null_rate)
    starts = run_starts(stream)
    n_seg = min(n_seg, len(starts) - 1)  # (half nulls: hardly a repeat, literals of 128 bytes all through)
    picks = [0] + sorted(rng.choice(np.arange(1, len(starts)), size=n_seg - 1, replace=False).tolist())
    entries = np.array([starts[i] for i in picks], dtype=np.uint64)
    rc, ref, ones_ref = oracle.boolean(stream, bits.size)
    assert rc == 0 and (ref == bits).all()
    rc, out, ones = gpu.boolean(stream, bits.size, entries if n_seg > 1 else None)
    assert rc == 0 and (out == bits).all() and ones == ones_ref
    packed = np.packbits(bits)
    rc, raw = gpu.byte_rle(stream, packed.size, entries if n_seg > 1 else None)
    assert rc == 0 and (raw == packed).all()


def test_segments_of_very_different_sizes(gpu, oracle):
    """One segment of 200 KB among segments of a few hundred bytes: the wavefront that holds it walks
    from memory, its neighbours out of LDS."""
    rng = np.random.default_rng(3)
    small = [bytes(rng.integers(0, 256, size=int(rng.integers(1, 5)), dtype=np.uint8)) + bytes([7]) * int(rng.integers(3, 9))
             for _ in range(4000)]
    big = bytes(rng.integers(0, 256, size=200_000, dtype=np.uint8))
    data = b"".join(small[:2000]) + big + b"".join(small[2000:])
    stream = enc.byte_rle(data)
    starts = run_starts(stream)
    # an entry point every 40 runs in the small parts, none inside the big one
    s_small = enc.byte_rle(b"".join(small[:2000]))
    n_before = len(run_starts(s_small))
    s_big = len(run_starts(enc.byte_rle(big)))
    picks = list(range(0, n_before - 40, 40)) + list(range(n_before + s_big + 40, len(starts), 40))
    entries = np.array([starts[i] for i in picks], dtype=np.uint64)
    want = np.frombuffer(data, dtype=np.uint8)
    rc, ref = oracle.byte_rle(stream, want.size)
    assert rc == 0 and (ref == want).all()
    rc, raw = gpu.byte_rle(stream, want.size, entries)
    assert rc == 0 and (raw == want).all()


def test_errors_are_the_same(gpu, oracle):
    rng = np.random.default_rng(4)
    bits, stream = present_like(rng, 400_000, 0.2)
    starts = run_starts(stream)
    entries = np.array(starts[::5], dtype=np.uint64)
    assert 64 <= entries.size <= 16_384
    # cut inside the last run or literal
    cut = stream[:-1]
    rc_ref, _, _ = oracle.boolean(cut, bits.size)
    assert rc_ref != 0
    rc, _, _ = gpu.boolean(cut, bits.size, entries)
    assert rc == rc_ref
    # more values wanted than the stream holds
    rc_ref, _, _ = oracle.boolean(stream, bits.size + 64)
    rc, _, _ = gpu.boolean(stream, bits.size + 64, entries)
    assert rc == rc_ref and rc != 0
    # an entry point that is no control byte of the chain: the segment before it does not end there
    bad = entries.copy()
    k = 20
    c = stream[int(bad[k])]
    bad[k] += 1 if c < 128 else 2   # inside the run's two bytes, or inside the literal
    rc, _, _ = gpu.boolean(stream, bits.size, bad)
    if rc == 0:  # the walk fell in with the chain again before the next entry point: then the values are right
        rc, out, _ = gpu.boolean(stream, bits.size, bad)
        assert (out == bits).all()

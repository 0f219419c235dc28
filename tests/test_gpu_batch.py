"""orcd_rlev2_decode_batch: several streams that lie in one buffer decoded as one, against the
per-stream decode and the oracle; a slice that does not hold its values is refused."""
import numpy as np
import pytest

import rle_encode as enc

pytestmark = pytest.mark.gpu

OUT_OF_SPEC = 1


def _blob(rng, n_slices, signed, with_entries, gaps=True, max_len=512):
    """Random streams laid into one buffer with junk between them."""
    parts, slices, expect = [], [], []
    pos = 0
    for _ in range(n_slices):
        if gaps:
            junk = bytes(rng.integers(0, 256, size=int(rng.integers(0, 40)), dtype=np.uint8))
            parts.append(junk)
            pos += len(junk)
        n_runs = int(rng.integers(0, 30))
        if n_runs == 0:
            slices.append((pos, 0, 0, None))
            continue
        stream, vals, starts = enc.random_stream(rng, n_runs, signed, max_len=max_len)
        ent = np.array(starts[::3], dtype=np.uint64) if with_entries else None
        slices.append((pos, len(stream), len(vals), ent))
        parts.append(stream)
        pos += len(stream)
        expect.append(vals)
    parts.append(b"\x00" * 7)
    return b"".join(parts), slices, (np.concatenate(expect) if expect else np.empty(0, np.int64))


@pytest.mark.parametrize("signed", [False, True])
@pytest.mark.parametrize("with_entries", [False, True])
def test_batch_equals_per_stream(oracle, signed, with_entries):
    from orc_format_amd import default_context
    ctx = default_context()
    rng = np.random.default_rng(40 + signed + 2 * with_entries)
    for case in range(12):
        blob, slices, expect = _blob(rng, int(rng.integers(1, 20)), signed, with_entries,
                                     max_len=int(rng.choice([8, 64, 512])))
        got = ctx.rlev2_batch(blob, slices, signed)
        assert (got == expect).all(), case
        # each slice on its own through the oracle gives the same pieces
        pos = 0
        for off, ln, nv, _ in slices:
            if nv == 0:
                continue
            rc, ref, _ = oracle.rlev2(blob[off:off + ln], nv, signed)
            assert rc == 0 and (got[pos:pos + nv] == ref).all()
            pos += nv


def test_batch_refuses_slices_out_of_order():
    """Slices are given in the order they lie in the buffer, without overlap."""
    from orc_format_amd import default_context
    from orc_format_amd.error import OrcError
    ctx = default_context()
    rng = np.random.default_rng(45)
    blob, slices, expect = _blob(rng, 6, True, False)
    live = [s for s in slices if s[2]]
    assert len(live) >= 2
    assert (ctx.rlev2_batch(blob, live, True) == expect).all()
    for bad in (live[::-1], [live[0], live[0]]):
        with pytest.raises(OrcError) as ei:
            ctx.rlev2_batch(blob, bad, True)
        assert ei.value.code == 102


def test_batch_refuses_wrong_counts_and_damage(oracle):
    from orc_format_amd import default_context
    from orc_format_amd.error import OrcError
    ctx = default_context()
    rng = np.random.default_rng(44)
    blob, slices, expect = _blob(rng, 8, True, False)
    live = [i for i, s in enumerate(slices) if s[2]]
    assert len(live) >= 3
    # one value too few / too many in a middle slice: later slices would land in the wrong place
    for delta in (-1, 1):
        k = live[1]
        bad = list(slices)
        off, ln, nv, ent = bad[k]
        bad[k] = (off, ln, nv + delta, ent)
        with pytest.raises(OrcError) as ei:
            ctx.rlev2_batch(blob, bad, True)
        assert ei.value.code == OUT_OF_SPEC and ctx.bad_slice <= max(live)
    # a slice cut short
    k = live[-1]
    bad = list(slices)
    off, ln, nv, ent = bad[k]
    bad[k] = (off, ln - 1, nv, ent)
    with pytest.raises(OrcError) as ei:
        ctx.rlev2_batch(blob, bad, True)
    assert ei.value.code == OUT_OF_SPEC
    # a slice outside the buffer
    bad = list(slices)
    bad[k] = (len(blob) - 2, 50, 5, None)
    with pytest.raises(OrcError):
        ctx.rlev2_batch(blob, bad, True)
    # nothing to do
    assert ctx.rlev2_batch(blob, [(0, 0, 0, None)], True).size == 0
    # and the undamaged batch still decodes afterwards
    assert (ctx.rlev2_batch(blob, slices, True) == expect).all()


def test_batch_on_device_pointers():
    import torch
    from bench_util_path import synth
    from orc_format_amd import default_context
    ctx = default_context()
    cols = [synth.direct_w17(37 + c, seed=7 + c, device="cuda") for c in range(9)]
    pad = torch.zeros(13, dtype=torch.uint8, device="cuda")
    pieces, slices, pos = [], [], 0
    for stream, entries, _ in cols:
        pieces += [pad, stream]
        pos += 13
        slices.append((pos, stream.numel(), stream.numel() // 1090 * 512, entries.cpu().numpy().astype(np.uint64)))
        pos += stream.numel()
    blob = torch.cat(pieces)
    expect = torch.cat([c[2] for c in cols])
    out = torch.empty(expect.numel(), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.rlev2_batch((blob.data_ptr(), blob.numel()), slices, True, out_ptr=out.data_ptr(), device=True)
    ctx.synchronize()
    assert torch.equal(out, expect)


def test_boolean_batch_equals_per_stream(oracle):
    from orc_format_amd import default_context
    from orc_format_amd.error import OrcError
    ctx = default_context()
    rng = np.random.default_rng(50)
    for case in range(10):
        parts, slices, want = [], [], []
        pos = 0
        for _ in range(int(rng.integers(1, 12))):
            junk = bytes(rng.integers(0, 256, size=int(rng.integers(0, 9)), dtype=np.uint8))
            parts.append(junk)
            pos += len(junk)
            n_bits = int(rng.integers(0, 30000)) if rng.integers(0, 5) else 0
            if n_bits == 0:
                slices.append((pos, 0, 0, None))
                want.append(np.zeros(0, dtype=bool))
                continue
            bits = rng.random(n_bits) < rng.choice([0.02, 0.5, 0.98])
            stream = enc.byte_rle(np.packbits(bits).tobytes())
            slices.append((pos, len(stream), n_bits, None))
            want.append(bits)
            parts.append(stream)
            pos += len(stream)
        blob = b"".join(parts) + b"\x00" * 3
        got, counts = ctx.boolean_batch(blob, slices)
        for g, w, cnt in zip(got, want, counts):
            assert (g == w).all() and cnt == int(w.sum())
    # a stream that holds a byte less than it should
    live = [i for i, s_ in enumerate(slices) if s_[2] > 64]
    if live:
        k = live[0]
        off, ln, nv, ent = slices[k]
        bad = list(slices)
        bad[k] = (off, ln, nv + 8, ent)
        with pytest.raises(OrcError):
            ctx.boolean_batch(blob, bad)
        assert [int(x.sum()) for x in ctx.boolean_batch(blob, slices)[0]] == [int(w.sum()) for w in want]

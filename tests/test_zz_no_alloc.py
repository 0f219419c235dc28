"""orcd_rlev2_decode_noalloc (every table in the library's static device array, no allocation)
against orcd_rlev2_decode on the same device buffers: two calls in a row, the second on a smaller
input, for a column the single-launch path takes and for one that needs the header scan; and the
limits it states."""
import numpy as np
import pytest

import rle_encode as enc

pytestmark = pytest.mark.gpu


def _device(torch, host_u8):
    return torch.from_numpy(np.frombuffer(bytes(host_u8), dtype=np.uint8).copy()).cuda()


def _both(torch, ctx, stream, n, signed, entries):
    d_stream = _device(torch, stream)
    d_entries = torch.from_numpy(np.asarray(entries, dtype=np.uint64).view(np.int64).copy()).cuda() \
        if entries is not None and len(entries) else None
    want = torch.full((n,), -7, dtype=torch.int64, device="cuda")
    got = torch.full((n,), -9, dtype=torch.int64, device="cuda")
    ep, ne = (d_entries.data_ptr(), d_entries.numel()) if d_entries is not None else (0, 0)
    torch.cuda.synchronize()
    ctx.rlev2_device(d_stream.data_ptr(), d_stream.numel(), n, signed, want.data_ptr(), ep, ne)
    ctx.synchronize()
    ctx.rlev2_device_noalloc(d_stream.data_ptr(), d_stream.numel(), n, signed, got.data_ptr(), ep, ne)
    ctx.synchronize()
    return want, got


def test_noalloc_matches_decode_twice_in_a_row(oracle):
    import torch
    from orc_format_amd import Context
    rng = np.random.default_rng(2024)
    with Context(0) as ctx:
        # irregular column (every run kind): header scan, run table, value expansion from the static array
        big, big_vals, big_starts = enc.random_stream(rng, 6000, True)
        small, small_vals, small_starts = enc.random_stream(rng, 700, True)
        for stream, vals, starts, step in ((big, big_vals, big_starts, 50), (small, small_vals, small_starts, 7)):
            entries = np.array(starts[::step], dtype=np.uint64)
            want, got = _both(torch, ctx, stream, vals.size, True, entries)
            assert torch.equal(want, got)
            rc, ref, _ = oracle.rlev2(stream, vals.size, True)
            assert rc == 0 and (got.cpu().numpy() == ref).all()
            assert ctx.last_path() == 0
        # the same two without entry points, and cut short inside a run
        for stream, vals in ((big, big_vals), (small, small_vals)):
            n = vals.size - 3
            want, got = _both(torch, ctx, stream, n, True, None)
            assert torch.equal(want, got) and (got.cpu().numpy() == vals[:n]).all()
    with Context(0) as ctx:   # (a context that has not yet backed off from speculating)
        # regular column: the single-launch path, its flag word in the static array
        for n_runs in (300, 40):
            parts, vals, starts, pos = [], [], [], 0
            for r in range(n_runs):
                v = rng.integers(-(1 << 15), 1 << 15, size=512, dtype=np.int64)
                b = enc.direct(v, 17, True)
                starts.append(pos)
                pos += len(b)
                parts.append(b)
                vals.append(v)
            stream, vals = b"".join(parts), np.concatenate(vals)
            want, got = _both(torch, ctx, stream, vals.size, True, np.array(starts[::20], dtype=np.uint64))
            assert torch.equal(want, got) and (got.cpu().numpy() == vals).all()
            assert ctx.last_path() == 1


def test_noalloc_limits_and_errors():
    import torch
    from orc_format_amd import Context
    from orc_format_amd.error import OrcError
    rng = np.random.default_rng(7)
    with Context(0) as ctx:
        stream, vals, _ = enc.random_stream(rng, 100, False)
        d_stream = _device(torch, stream)
        out = torch.zeros(vals.size, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        with pytest.raises(OrcError) as e:   # more values than the static tables are sized for
            ctx.rlev2_device_noalloc(d_stream.data_ptr(), d_stream.numel(), (1 << 24) + 1, False, out.data_ptr())
        assert e.value.code == 102
        with pytest.raises(OrcError) as e:   # more entry points than the static tables are sized for
            ctx.rlev2_device_noalloc(d_stream.data_ptr(), d_stream.numel(), vals.size, False, out.data_ptr(),
                                     d_stream.data_ptr(), (1 << 16) + 1)
        assert e.value.code == 102
        with pytest.raises(OrcError) as e:   # a truncated stream is OutOfSpec, as in orcd_rlev2_decode
            ctx.rlev2_device_noalloc(d_stream.data_ptr(), d_stream.numel() - 3, vals.size, False, out.data_ptr())
        assert e.value.code == 1
        # and the array is fit for use afterwards
        ctx.rlev2_device_noalloc(d_stream.data_ptr(), d_stream.numel(), vals.size, False, out.data_ptr())
        ctx.synchronize()
        assert (out.cpu().numpy() == vals.view(np.int64)).all()


def test_noalloc_more_runs_than_the_table_holds():
    """A stream within the value limit but with more runs than the static run table holds is turned
    down, not decoded wrongly."""
    import torch
    from orc_format_amd import Context
    from orc_format_amd.error import OrcError
    n_runs = (1 << 19) + 10
    stream = bytes([0x00, 0x05]) * n_runs        # ShortRepeat, 3 x value 5, again and again
    with Context(0) as ctx:
        d_stream = _device(torch, stream)
        out = torch.zeros(3 * n_runs, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        with pytest.raises(OrcError) as e:
            ctx.rlev2_device_noalloc(d_stream.data_ptr(), d_stream.numel(), 3 * n_runs, False, out.data_ptr())
        assert e.value.code == 102
        ctx.rlev2_device(d_stream.data_ptr(), d_stream.numel(), 3 * n_runs, False, out.data_ptr())
        ctx.synchronize()
        assert bool((out == 5).all())


def test_noalloc_allocates_nothing(oracle):
    """What the entry point is for: on a context that has decoded nothing yet -- its buffers are all
    unallocated -- the device's free memory is the same after the calls as before, on the path with
    the header scan and on the single-launch path."""
    import torch
    from orc_format_amd import Context
    rng = np.random.default_rng(99)
    irregular, irr_vals, irr_starts = enc.random_stream(rng, 3000, True)
    parts, vals, starts, pos = [], [], [], 0
    for r in range(200):
        v = rng.integers(-(1 << 15), 1 << 15, size=512, dtype=np.int64)
        b = enc.direct(v, 17, True)
        starts.append(pos)
        pos += len(b)
        parts.append(b)
        vals.append(v)
    regular, reg_vals = b"".join(parts), np.concatenate(vals)
    cases = []
    for stream, want, ent in ((irregular, irr_vals, irr_starts[::40]), (regular, reg_vals, starts[::20])):
        d_stream = _device(torch, stream)
        d_ent = torch.from_numpy(np.asarray(ent, dtype=np.uint64).view(np.int64).copy()).cuda()
        out = torch.zeros(want.size, dtype=torch.int64, device="cuda")
        cases.append((d_stream, d_ent, out, want))
    with Context(0) as ctx:
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        paths = []
        for _ in range(2):
            for d_stream, d_ent, out, want in cases:
                ctx.rlev2_device_noalloc(d_stream.data_ptr(), d_stream.numel(), want.size, True, out.data_ptr(),
                                         d_ent.data_ptr(), d_ent.numel())
                paths.append(ctx.last_path())
        ctx.synchronize()
        torch.cuda.synchronize()
        free_after = torch.cuda.mem_get_info()[0]
        assert free_after == free_before, (free_before, free_after)
        assert paths[:2] == [0, 1]
        for d_stream, d_ent, out, want in cases:
            assert (out.cpu().numpy() == want).all()

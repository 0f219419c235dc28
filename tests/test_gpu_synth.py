"""The benchmark's stream generators at small sizes: what they claim to encode is what the oracle
decodes, and the HIP path (device pointers, entry points) agrees."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,kw,signed", [("direct_w17", {}, True),
                                            ("delta_monotonic", {"width": 10}, True),
                                            ("delta_monotonic", {"width": 16}, True),
                                            ("string_lengths_w6", {}, False)])
def test_rlev2_generators(oracle, name, kw, signed):
    import torch
    from bench_util import synth
    from orc_format_amd import default_context
    ctx = default_context()
    n_runs = 203
    stream, entries, values = getattr(synth, name)(n_runs, device="cuda", **kw)
    n = n_runs * 512
    rc, ref, consumed = oracle.rlev2(stream.cpu().numpy(), n, signed)
    assert rc == 0 and consumed == stream.numel()
    assert (ref == values.cpu().numpy()).all()
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    for ent in (entries, None):
        out.zero_()
        ctx.rlev2_device(stream.data_ptr(), stream.numel(), n, signed, out.data_ptr(),
                         ent.data_ptr() if ent is not None else 0, ent.numel() if ent is not None else 0)
        ctx.synchronize()
        assert torch.equal(out, values)


def test_mixed_short_runs_generator(oracle):
    import torch
    from bench_util import synth
    from orc_format_amd import default_context
    ctx = default_context()
    stream, entries, values, entry_values = synth.mixed_short_runs(5000, device="cuda", chunk=1500)
    n = values.numel()
    rc, ref, consumed = oracle.rlev2(stream.cpu().numpy(), n, False)
    assert rc == 0 and consumed == stream.numel()
    assert (ref == values.cpu().numpy()).all()
    rc, runs = oracle.runs(stream.cpu().numpy(), n)
    starts = {r[3] for r in runs}      # header offset of every run
    assert set(entries.cpu().tolist()) <= starts
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    for ent in (entries, None):
        for count in (n, n - 7, int(entry_values[3])):
            out.fill_(-1)
            ctx.rlev2_device(stream.data_ptr(), stream.numel(), count, False, out.data_ptr(),
                             ent.data_ptr() if ent is not None else 0, ent.numel() if ent is not None else 0)
            ctx.synchronize()
            assert torch.equal(out[:count], values[:count]) and (out[count:] == -1).all()


@pytest.mark.parametrize("run_fraction", [0.0, 0.3, 1.0])
def test_byte_rle_generator(oracle, run_fraction):
    import torch
    from bench_util import synth
    from orc_format_amd import default_context
    ctx = default_context()
    stream, entries, packed = synth.byte_rle_mix(777, run_fraction=run_fraction, device="cuda")
    host = stream.cpu().numpy()
    rc, ref = oracle.byte_rle(host, packed.numel())
    assert rc == 0 and (ref == packed.cpu().numpy()).all()
    for num_bits in (packed.numel() * 8, packed.numel() * 8 - 3, 8 * 1024 * 5 + 1, 7):
        want = synth.unpack_bools(packed, num_bits)
        rc, ref_b, ref_ones = oracle.boolean(host, num_bits)
        assert rc == 0 and (ref_b == want.cpu().numpy().astype(bool)).all()
        for ent in (entries, None):
            for skew in (0, 1, 8):
                buf = torch.full((num_bits + 32,), 0xEE, dtype=torch.uint8, device="cuda")
                bitmap = torch.full(((num_bits + 7) // 8 + 2,), 0xEE, dtype=torch.uint8, device="cuda")
                ones = ctx.boolean_device(stream.data_ptr(), stream.numel(), num_bits,
                                          buf.data_ptr() + skew, bitmap.data_ptr() + (skew & 1),
                                          ent.data_ptr() if ent is not None else 0,
                                          ent.numel() if ent is not None else 0)
                ctx.synchronize()
                assert ones == ref_ones
                assert torch.equal(buf[skew:skew + num_bits], want)
                assert (buf[:skew] == 0xEE).all() and (buf[skew + num_bits:] == 0xEE).all()
                bm = bitmap.cpu().numpy()
                nb = (num_bits + 7) // 8
                got_bits = np.unpackbits(bm[(skew & 1):(skew & 1) + nb], bitorder="little")[:num_bits]
                assert (got_bits == want.cpu().numpy()).all()
                assert bm[(skew & 1) + nb] == 0xEE


def test_long_streams_without_entry_points(oracle):
    """Streams long enough for the decoder to look for entry points of its own (stretches of
    equal-sized runs confirmed in parallel): uniform, uniform with breaks, and not uniform at all."""
    import torch
    from bench_util import synth
    from orc_format_amd import default_context
    ctx = default_context()
    a = synth.direct_w17(1500, seed=1, device="cuda")
    b = synth.delta_monotonic(900, width=10, seed=2, device="cuda")
    c = synth.direct_w17(37, seed=3, device="cuda")
    d = synth.delta_monotonic(1100, width=16, seed=4, device="cuda")
    m = synth.mixed_short_runs(40000, seed=5, device="cuda")
    cases = {
        "uniform": [a],
        "breaks": [a, b, c, d, c, a],
        "short runs": [(m[0], m[1], m[2])],
        "short then uniform": [(m[0], m[1], m[2]), a, b],
    }
    for name, parts in cases.items():
        signed = name in ("uniform", "breaks")
        stream = torch.cat([p[0] for p in parts])
        assert stream.numel() > (1 << 18)
        n = sum(int(p[2].numel()) for p in parts)
        rc, ref, _ = oracle.rlev2(stream.cpu().numpy(), n, signed)
        assert rc == 0
        want = torch.from_numpy(ref).cuda()
        out = torch.empty(n, dtype=torch.int64, device="cuda")
        for count in (n, n - 1000):
            out.fill_(-7)
            ctx.rlev2_device(stream.data_ptr(), stream.numel(), count, signed, out.data_ptr(), 0, 0)
            ctx.synchronize()
            assert torch.equal(out[:count], want[:count]), (name, count)
            assert (out[count:] == -7).all()
        # damaged in the middle of a uniform stretch: an error, not a crash and not wrong values
        bad = stream.clone()
        bad[stream.numel() // 2] ^= 0x40
        rc_bad, ref_bad, _ = oracle.rlev2(bad.cpu().numpy(), n, signed)
        from orc_format_amd.error import OrcError
        try:
            ctx.rlev2_device(bad.data_ptr(), bad.numel(), n, signed, out.data_ptr(), 0, 0)
            ctx.synchronize()
            assert rc_bad == 0 and torch.equal(out, torch.from_numpy(ref_bad).cuda()), name
        except OrcError:
            assert rc_bad != 0, name


def test_large_pageable_host_buffers():
    """Host buffers of tens of megabytes that are not pinned travel in chunks through pinned
    staging on several threads, in both directions; the result is the same column.  A second call
    (chunks still marked busy from the first) and a pinned input take the other ways through."""
    import torch
    from bench_util import synth
    from orc_format_amd import default_context
    ctx = default_context()
    n_runs = 40_000                       # 43.6 MB of stream, 164 MB of values
    stream, entries, values = synth.direct_w17(n_runs, seed=7, device="cuda")
    n = n_runs * 512
    want = values.cpu().numpy()
    h_stream = stream.cpu().numpy().copy()
    h_entries = entries.cpu().numpy().astype(np.uint64)
    assert h_stream.nbytes >= 32 << 20
    for _ in range(2):
        out = ctx.rlev2(h_stream, n, True, h_entries)
        assert (out == want).all()
    # an odd length: the last chunk is partial and unaligned
    cut = n - 512 * 3 - 1
    out = ctx.rlev2(h_stream, cut, True, h_entries)
    assert (out == want[:cut]).all()
    pinned = torch.from_numpy(h_stream).pin_memory()
    out = ctx.rlev2(pinned.numpy(), n, True, h_entries)
    assert (out == want).all()


def test_large_pageable_results_of_other_decoders():
    """The same chunked way back to the host for boolean values and string bytes."""
    import torch
    from bench_util import synth
    from orc_format_amd import default_context
    ctx = default_context()
    # 41 M boolean values: 41 MB of one-byte values go back
    stream, entries, decoded = synth.byte_rle_mix(40_000, seed=3, device="cuda")
    bits = synth.unpack_bools(decoded, decoded.numel() * 8).cpu().numpy().astype(bool)
    assert bits.size >= 32 << 20
    out, ones = ctx.boolean(stream.cpu().numpy(), bits.size, entries.cpu().numpy().astype(np.uint64))
    assert (out == bits).all() and ones == int(bits.sum())
    # 1.2 M strings, 39 MB of ASCII bytes in and out
    n_runs = 2400
    lstream, lentries, lengths = synth.string_lengths_w6(n_runs, seed=5, device="cuda")
    n = n_runs * 512
    total = int(lengths.sum())
    assert total >= 32 << 20
    data = torch.randint(32, 127, (total,), dtype=torch.uint8, device="cuda").cpu().numpy()
    offsets, got = ctx.string_direct(lstream.cpu().numpy(), data, n, check_utf8=True,
                                     length_entry_points=lentries.cpu().numpy().astype(np.uint64))
    want_off = np.concatenate([[0], np.cumsum(lengths.view(-1).cpu().numpy())])
    assert (offsets == want_off).all() and got.tobytes() == data.tobytes()


def test_decode_on_the_callers_stream():
    """orcd_ctx_set_stream: the context's work is queued on a stream of the caller's -- after
    what the caller queued there before (the generator's kernels, no synchronisation in between)
    and before what it queues afterwards."""
    import torch
    from bench_util import synth
    from orc_format_amd import Context
    ctx = Context(0)
    side = torch.cuda.Stream()
    ctx.set_stream(side.cuda_stream)
    try:
        with torch.cuda.stream(side):
            for seed in (1, 2, 3):
                stream, entries, values = synth.delta_monotonic(300, seed=seed, device="cuda", width=10)
                n = 300 * 512
                out = torch.empty(n, dtype=torch.int64, device="cuda")
                ctx.rlev2_device(stream.data_ptr(), stream.numel(), n, True, out.data_ptr(),
                                 entries.data_ptr(), entries.numel())
                doubled = out * 2          # queued behind the decode on the same stream
                side.synchronize()
                assert torch.equal(out, values) and torch.equal(doubled, values * 2)
    finally:
        ctx.set_stream(None)

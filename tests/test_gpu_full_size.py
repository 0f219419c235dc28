"""The decoders at the sizes the benchmark quotes (10^9-row integer column, 8 GB of string DATA,
4*10^9 booleans), checked through properties that need no CPU decode: the columns come from
device-side generators (bench_util/synth.py) that produce stream and values together, so the
decoded column must equal the generator's values; beside that, sortedness for the Delta column,
offsets that are the prefix sums of the lengths and end at the DATA length (past 4 GiB: every
offset beyond 32 bits), and the count of true values.  The same generators are checked against the
CPU oracle at small sizes in test_gpu_synth.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from orc_format_amd import Context
    c = Context(0)
    yield c
    c.synchronize()


def _free():
    import torch
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def test_integer_column_of_a_billion_rows(ctx):
    """BASELINE configs[1]: 10^9 signed values, Direct width 17, runs of 512 (rle_v2.rs:126-182,
    564-593), an entry point every 20 runs.  Decoded with and without entry points."""
    import torch
    from bench_util import synth
    n_runs = 1_000_000_000 // 512
    n = n_runs * 512
    stream, entries, expect = synth.direct_w17(n_runs, device="cuda")
    assert stream.numel() == n_runs * 1090
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    for use in (True, False):
        out.zero_()
        torch.cuda.synchronize()  # the context decodes on a stream of its own
        ctx.rlev2_device(stream.data_ptr(), stream.numel(), n, True, out.data_ptr(),
                         entries.data_ptr() if use else 0, entries.numel() if use else 0)
        ctx.synchronize()
        assert torch.equal(out, expect)
    # zigzag of 17-bit fields: every value within +-2^16
    assert int(out.max().item()) < (1 << 16) and int(out.min().item()) >= -(1 << 16)
    # stopping early: num_values inside a run, the rest of the output untouched
    out.fill_(-7)
    m = n - 512 * 3 - 100
    torch.cuda.synchronize()
    ctx.rlev2_device(stream.data_ptr(), stream.numel(), m, True, out.data_ptr(), entries.data_ptr(),
                     entries.numel())
    ctx.synchronize()
    assert torch.equal(out[:m], expect[:m]) and bool((out[m:] == -7).all())
    del stream, entries, expect, out
    _free()


def test_delta_column_of_a_billion_rows_is_sorted(ctx):
    """10^9 strictly increasing values as Delta runs of width 10 (rle_v2.rs:184-281): the decoded
    column is sorted whatever the generator says, and equals the generator's values."""
    import torch
    from bench_util import synth
    n_runs = 1_000_000_000 // 512
    n = n_runs * 512
    stream, entries, expect = synth.delta_monotonic(n_runs, width=10, device="cuda")
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.rlev2_device(stream.data_ptr(), stream.numel(), n, True, out.data_ptr(), entries.data_ptr(),
                     entries.numel())
    ctx.synchronize()
    step = 1 << 27
    for a in range(0, n - 1, step):  # in pieces: the differences of 10^9 values are 8 GB more
        b = min(n, a + step + 1)
        assert bool((out[a + 1:b] > out[a:b - 1]).all())
    assert torch.equal(out, expect)
    del stream, entries, expect, out
    _free()


def test_string_column_with_data_beyond_4_gib(ctx):
    """2.5*10^8 strings over 8 GB of DATA (variable_length.rs:5-28 driven by an unsigned RLE v2
    LENGTH stream): offsets are the prefix sums of the lengths, start at 0, never decrease, end at
    the DATA length (beyond 32 bits), and the copied bytes are the DATA bytes."""
    import torch
    from bench_util import synth
    n_runs = 250_000_000 // 512
    n = n_runs * 512
    lstream, entries, lengths = synth.string_lengths_w6(n_runs, device="cuda")
    want_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    want_off[0] = 0
    torch.cumsum(lengths, 0, out=want_off[1:])
    total = int(want_off[-1].item())
    assert total > (1 << 32)
    data = torch.empty(total, dtype=torch.uint8, device="cuda")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    for b0 in range(0, total, 1 << 28):
        b1 = min(total, b0 + (1 << 28))
        data[b0:b1] = torch.randint(32, 127, (b1 - b0,), dtype=torch.uint8, device="cuda", generator=gen)
    offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    out = torch.zeros(total, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.string_direct_device(lstream.data_ptr(), lstream.numel(), data.data_ptr(), total, n,
                             offsets.data_ptr(), out.data_ptr(), entries.data_ptr(), entries.numel(),
                             check_utf8=True)
    ctx.synchronize()
    assert int(offsets[0].item()) == 0 and int(offsets[-1].item()) == total
    assert torch.equal(offsets[1:] - offsets[:-1], lengths)
    assert torch.equal(offsets, want_off)
    assert torch.equal(out, data)
    # one ill-formed byte in the last gigabyte is found (tests/it/deserialize.rs:116-118)
    from orc_format_amd import error
    where = total - 12345
    data[where] = 0xFF
    torch.cuda.synchronize()
    with pytest.raises(error.OrcError) as caught:
        ctx.string_direct_device(lstream.data_ptr(), lstream.numel(), data.data_ptr(), total, n,
                                 offsets.data_ptr(), out.data_ptr(), entries.data_ptr(),
                                 entries.numel(), check_utf8=True)
        ctx.synchronize()
    assert caught.value.variant == "InvalidUtf8"
    del lstream, entries, lengths, want_off, data, offsets, out
    _free()


def test_four_billion_booleans(ctx):
    """4*10^9 booleans (boolean_rle.rs:14-158, runs and literals mixed): one byte per value equal
    to the bits of the generator's bytes, most significant first, and the count of true values."""
    import torch
    from bench_util import synth
    n_bits = 4_000_000_000
    n_blocks = n_bits // 8 // 128
    stream, entries, decoded = synth.byte_rle_mix(n_blocks, device="cuda")
    n_bytes = decoded.numel()
    n_bits = n_bytes * 8
    out = torch.empty(n_bits, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ones = ctx.boolean_device(stream.data_ptr(), stream.numel(), n_bits, out.data_ptr(), 0,
                              entries.data_ptr(), entries.numel())
    ctx.synchronize()
    want_ones = 0
    step = 1 << 26
    for a in range(0, n_bytes, step):
        b = min(n_bytes, a + step)
        want = synth.unpack_bools(decoded[a:b], (b - a) * 8)
        assert torch.equal(out[a * 8:b * 8], want)
        want_ones += int(want.sum(dtype=torch.int64).item())
    assert int(ones) == want_ones
    del stream, entries, decoded, out
    _free()


@pytest.mark.parametrize("runs_per_entry", [64, 512])
def test_half_a_billion_values_in_short_runs(ctx, runs_per_entry):
    """2.5*10^7 runs of ShortRepeat (3..10 copies, rle_v2.rs:283-329) and Direct width 4 (1..64
    values) -- about 25 runs under every 512 values, the many-runs path of the value kernel --
    with an entry point every 64 or every 512 runs; whole and stopped at an entry point's value."""
    import torch
    from bench_util import synth
    stream, entries, expect, entry_values = synth.mixed_short_runs(
        500_000_000 // 20, runs_per_entry=runs_per_entry, device="cuda")
    n = int(expect.numel())
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    cut = int(entry_values[entry_values.numel() // 2].item()) + 3
    for count in (n, cut):
        out.fill_(-7)
        torch.cuda.synchronize()
        ctx.rlev2_device(stream.data_ptr(), stream.numel(), count, False, out.data_ptr(),
                         entries.data_ptr(), entries.numel())
        ctx.synchronize()
        assert torch.equal(out[:count], expect[:count])
        assert bool((out[count:] == -7).all())
    del stream, entries, expect, entry_values, out
    _free()


def test_more_than_2_to_32_values(ctx):
    """A column with more values than 32 bits count: 8.5*10^6 Delta runs of 512 values with a fixed
    delta (width 0: header, base, delta base and nothing else, rle_v2.rs:184-281), 4 stream bytes
    each, 35 GB decoded.  Every run is 5, 8, 11, ...; the last values sit past index 2^32."""
    import torch
    n_runs = 8_500_000
    n = n_runs * 512
    assert n > (1 << 32)
    run = torch.tensor([0xC1, 0xFF, 0x0A, 0x06], dtype=torch.uint8, device="cuda")  # zigzag(5), zigzag(3)
    stream = run.repeat(n_runs)
    entries = torch.arange(0, n_runs, 32, dtype=torch.int64, device="cuda") * 4
    want_run = torch.arange(512, dtype=torch.int64, device="cuda") * 3 + 5
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    for use in (True, False):
        out.fill_(-7)
        torch.cuda.synchronize()
        ctx.rlev2_device(stream.data_ptr(), stream.numel(), n, True, out.data_ptr(),
                         entries.data_ptr() if use else 0, entries.numel() if use else 0)
        ctx.synchronize()
        rows = out.view(n_runs, 512)
        step = 1 << 19
        for a in range(0, n_runs, step):
            assert bool((rows[a:a + step] == want_run).all()), (use, a)
    del stream, entries, out, rows
    _free()


def test_more_than_2_to_32_booleans(ctx):
    """5*10^9 + 3 booleans from 4.9*10^6 byte-RLE runs of 130 bytes (boolean_rle.rs:40-62; the run
    emits header + 3 copies), alternately 0xff and 0x0f: bytes, bitmap and the count of ones."""
    import torch
    n_bits = 5_000_000_003
    n_bytes = (n_bits + 7) // 8
    n_runs = (n_bytes + 129) // 130
    n_runs += n_runs & 1
    stream = torch.tensor([0x7F, 0xFF, 0x7F, 0x0F], dtype=torch.uint8, device="cuda").repeat(n_runs // 2)
    entries = torch.arange(0, n_runs, 50, dtype=torch.int64, device="cuda") * 2
    out = torch.empty(n_bits, dtype=torch.uint8, device="cuda")
    bitmap = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ones = ctx.boolean_device(stream.data_ptr(), stream.numel(), n_bits, out.data_ptr(),
                              bitmap.data_ptr(), entries.data_ptr(), entries.numel())
    ctx.synchronize()
    # decoded byte k belongs to run k // 130; even runs are 0xff, odd runs 0x0f (0,0,0,0,1,1,1,1)
    want_ones = 0
    step = 130 * 2 * (1 << 18)  # whole pairs of runs
    pair = torch.cat([torch.ones(130 * 8, dtype=torch.uint8, device="cuda"),
                      torch.tensor([0, 0, 0, 0, 1, 1, 1, 1], dtype=torch.uint8, device="cuda").repeat(130)])
    pair_bm = torch.cat([torch.full((130,), 0xFF, dtype=torch.uint8, device="cuda"),
                         torch.full((130,), 0xF0, dtype=torch.uint8, device="cuda")])  # least significant bit first
    for a in range(0, n_bytes, step):
        b = min(n_bytes, a + step)
        bits = min(n_bits, b * 8) - a * 8
        want = pair.repeat((b - a + 259) // 260)[:bits]
        assert torch.equal(out[a * 8:a * 8 + bits], want), a
        want_ones += int(want.sum(dtype=torch.int64).item())
        whole = bits // 8
        assert torch.equal(bitmap[a:a + whole], pair_bm.repeat((b - a + 259) // 260)[:whole]), a
    assert int(ones) == want_ones
    del stream, entries, out, bitmap
    _free()

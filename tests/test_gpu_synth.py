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

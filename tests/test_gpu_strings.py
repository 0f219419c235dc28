"""String columns on the GPU: LENGTH streams of every run kind expanded straight into offsets,
the one-pass copy + UTF-8 validation at every source / destination alignment, and agreement with
both the oracle and Python's own UTF-8 decoder on well-formed and ill-formed values."""
import numpy as np
import pytest

import rle_encode as enc

pytestmark = pytest.mark.gpu

INVALID_UTF8 = 2
OUT_OF_SPEC = 1


@pytest.fixture(scope="module")
def gpu():
    from gpu_backend import GpuBackend
    return GpuBackend()


def _length_stream(lens, width=8):
    lens = np.asarray(lens, dtype=np.uint64)
    return b"".join(enc.direct(lens[i:i + 512], width, False) for i in range(0, len(lens), 512))


def _python_says_valid(data, lens):
    pos = 0
    for l in lens:
        try:
            bytes(data[pos:pos + int(l)]).decode("utf-8")
        except UnicodeDecodeError:
            return False
        pos += int(l)
    return True


CHARS = ["a", "Z", "\x00", "\x7f", "\u0080", "é", "߿", "ࠀ", "€", "퟿", "",
         "￿", "\U00010000", "\U0001f600", "\U0010ffff", "࿿", "က", "쿿", "\U00040000"]


def _random_text(rng, n_chars):
    return "".join(CHARS[int(k)] for k in rng.integers(0, len(CHARS), size=n_chars))


ILL_FORMED = [b"\x80", b"\xbf", b"\xc0\x80", b"\xc1\xbf", b"\xc2", b"\xc2\x41", b"\xe0\x80\x80",
              b"\xe0\x9f\xbf", b"\xed\xa0\x80", b"\xed\xbf\xbf", b"\xe1\x80", b"\xe1\x80\x41",
              b"\xf0\x80\x80\x80", b"\xf0\x8f\xbf\xbf", b"\xf4\x90\x80\x80", b"\xf5\x80\x80\x80",
              b"\xf8\x88\x80\x80\x80", b"\xff", b"\xfe", b"\xf1\x80\x80", b"\xf1\x80\x80\x41",
              b"\xc2\x80\x80", b"\xe1\x80\x80\x80", b"\xf1\x80\x80\x80\x80"]


def test_utf8_acceptance_matches_python_and_oracle(gpu, oracle):
    rng = np.random.default_rng(21)
    n_bad = n_good = 0
    for case in range(400):
        pieces = []
        for _ in range(int(rng.integers(1, 12))):
            pieces.append(_random_text(rng, int(rng.integers(0, 30))).encode("utf-8"))
            if rng.integers(0, 25) == 0:
                pieces.append(ILL_FORMED[int(rng.integers(0, len(ILL_FORMED)))])
        data = b"".join(pieces)
        if not data:
            continue
        if rng.integers(0, 8) == 0:       # cut somewhere: may split a character at the very end
            data = data[:int(rng.integers(1, len(data) + 1))]
        # value boundaries anywhere, also inside characters
        n = int(rng.integers(1, 8))
        cuts = np.sort(rng.integers(0, len(data) + 1, size=n - 1))
        if rng.integers(0, 2) == 0:       # ... or (mostly) on character boundaries
            starts = np.array([i for i in range(len(data)) if (data[i] & 0xC0) != 0x80] + [len(data)])
            cuts = np.sort(starts[rng.integers(0, len(starts), size=n - 1)])
        lens = np.diff(np.concatenate([[0], cuts, [len(data)]])).astype(np.uint64)
        ls = _length_stream(lens, 16)
        want = _python_says_valid(data, lens)
        rc_cpu, _, _ = oracle.string_direct(ls, data, n)
        rc_gpu, off, out = gpu.string_direct(ls, data, n)
        assert (rc_cpu == 0) == want, (case, data.hex())
        assert rc_gpu == rc_cpu, (case, rc_gpu, rc_cpu, data.hex(), lens.tolist())
        if want:
            assert out.tobytes() == data and off[-1] == len(data)
            n_good += 1
        else:
            assert rc_gpu == INVALID_UTF8
            n_bad += 1
    assert n_good > 50 and n_bad > 50


@pytest.mark.parametrize("tail", [b"\xc2", b"\xe1\x80", b"\xf1\x80\x80", b"\xe1", b"\xf1", b"\xf1\x80"])
def test_sequence_cut_by_the_end_of_the_data(gpu, oracle, tail):
    # every residue of the total length modulo 16, so that the end falls on and off a chunk edge
    for pad in range(0, 40):
        data = b"x" * pad + tail
        ls = _length_stream([len(data)], 8)
        rc_cpu, _, _ = oracle.string_direct(ls, data, 1)
        rc_gpu, _, _ = gpu.string_direct(ls, data, 1)
        assert rc_cpu == INVALID_UTF8 and rc_gpu == INVALID_UTF8, pad


def test_device_pointers_at_every_alignment(gpu, oracle):
    import torch
    from orc_format_amd import default_context
    from orc_format_amd.error import OrcError
    ctx = default_context()
    rng = np.random.default_rng(22)
    n = 5000
    # every value is made of whole characters
    vals = []
    chars = _random_text(rng, 3 * n)
    ci = 0
    for _ in range(n):
        k = int(rng.integers(0, 4))
        vals.append(chars[ci:ci + k].encode("utf-8"))
        ci += k
    lens = np.array([len(v) for v in vals], dtype=np.uint64)
    data = b"".join(vals)
    total = len(data)
    assert total > 4096
    ls = _length_stream(lens, 8)
    d_ls = torch.tensor(np.frombuffer(ls, dtype=np.uint8).copy(), device="cuda")
    want_off = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    for src_skew in (0, 1, 3, 4, 7, 8, 13, 15):
        for dst_skew in (src_skew, (src_skew + 5) % 16):
            d_src = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
            d_src[src_skew:src_skew + total] = torch.tensor(np.frombuffer(data, dtype=np.uint8).copy())
            d_dst = torch.full((total + 64,), 0xEE, dtype=torch.uint8, device="cuda")
            d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
            for check in (False, True):
                d_dst.fill_(0xEE)
                ctx.string_direct_device(d_ls.data_ptr(), d_ls.numel(), d_src.data_ptr() + src_skew, total,
                                         n, d_off.data_ptr(), d_dst.data_ptr() + dst_skew, check_utf8=check)
                ctx.synchronize()
                got = d_dst.cpu().numpy()
                assert got[dst_skew:dst_skew + total].tobytes() == data, (src_skew, dst_skew, check)
                assert (got[:dst_skew] == 0xEE).all() and (got[dst_skew + total:] == 0xEE).all()
                assert (d_off.cpu().numpy() == want_off).all()
            # validation only (no copy), and an ill-formed byte planted in the middle / at the ends
            ctx.string_direct_device(d_ls.data_ptr(), d_ls.numel(), d_src.data_ptr() + src_skew, total,
                                     n, d_off.data_ptr(), 0, check_utf8=True)
            for where in (0, total // 2, total - 1):
                saved = int(d_src[src_skew + where])
                d_src[src_skew + where] = 0xFF
                with pytest.raises(OrcError) as ei:
                    ctx.string_direct_device(d_ls.data_ptr(), d_ls.numel(), d_src.data_ptr() + src_skew,
                                             total, n, d_off.data_ptr(), 0, check_utf8=True)
                assert ei.value.code == INVALID_UTF8
                d_src[src_skew + where] = saved


def test_value_starting_on_a_continuation_byte(gpu, oracle):
    # the concatenation is well formed, a value boundary falls inside a character
    data = ("ab" + "€" + "cd").encode("utf-8") * 3000      # many tiles' worth? no: 21 KB
    data = data * 8                                            # 168 KB: three tiles
    n = len(data) // 7
    lens = np.full(n, 7, dtype=np.uint64)
    ls = _length_stream(lens, 4)
    assert gpu.string_direct(ls, data, n)[0] == 0
    for victim in (0, n // 3, n - 2):
        bad = lens.copy()
        bad[victim] = 3            # "ab" + first byte of the euro sign
        bad[victim + 1] = 11
        ls = _length_stream(bad, 4)
        rc_cpu, _, _ = oracle.string_direct(ls, data, n)
        rc_gpu, _, _ = gpu.string_direct(ls, data, n)
        assert rc_cpu == INVALID_UTF8 and rc_gpu == INVALID_UTF8, victim


def _length_runs(rng, n_runs, kinds):
    """A LENGTH stream of small non-negative values made of the given run kinds."""
    chunks, lens = [], []
    for _ in range(n_runs):
        kind = int(rng.choice(kinds))
        if kind == 0:
            v, cnt = int(rng.integers(0, 60)), int(rng.integers(3, 11))
            chunks.append(enc.short_repeat(v, cnt, False))
            lens.append(np.full(cnt, v, dtype=np.int64))
        elif kind == 1:
            w = int(rng.choice([1, 2, 3, 5, 8, 11]))
            L = int(rng.integers(1, 513))
            v = rng.integers(0, 2**w, size=L).astype(np.uint64)
            chunks.append(enc.direct(v, w, False))
            lens.append(v.astype(np.int64))
        elif kind == 3:
            L = int(rng.integers(2, 513))
            if rng.integers(0, 3) == 0:
                base, db = int(rng.integers(0, 50)), int(rng.integers(0, 3))
                chunks.append(enc.delta(base, db, L - 2, 0, False))
                lens.append(base + db * np.arange(L, dtype=np.int64))
            else:
                w = int(rng.choice([2, 3, 4]))
                d = rng.integers(0, 2**w, size=L - 2).astype(np.uint64)
                neg = bool(rng.integers(0, 2))
                db = int(rng.integers(1, 9))
                base = int(d.sum()) + db + int(rng.integers(0, 40)) if neg else int(rng.integers(0, 40))
                chunks.append(enc.delta(base, -db if neg else db, d, w, False))
                steps = np.concatenate([[base, -db if neg else db], -d.astype(np.int64) if neg else d.astype(np.int64)])
                lens.append(np.cumsum(steps))
        else:
            L = int(rng.integers(2, 513))
            w, pw = int(rng.choice([3, 5, 8])), int(rng.choice([2, 4]))
            base = int(rng.integers(0, 30))
            low = rng.integers(0, 2**w, size=L).astype(np.uint64)
            full = low.astype(np.int64)
            patches, last = [], 0
            for _k in range(int(rng.integers(1, 5))):
                gap = int(rng.integers(0 if not patches else 1, 200))
                if last + gap >= L:
                    break
                pv = int(rng.integers(1, 2**pw))
                patches.append((gap, pv))
                last += gap
                full[last] |= pv << w
            if not patches:
                patches, full[0] = [(0, 1)], full[0] | (1 << w)
            chunks.append(enc.patched_base(base, low, w, patches, pw, 8))
            lens.append(full + base)
    return b"".join(chunks), np.concatenate(lens)


def test_lengths_of_every_run_kind(gpu, oracle):
    """LENGTH streams mixing ShortRepeat, Direct, Delta and PatchedBase runs: the offsets must be
    the prefix sums either way (patched runs take the materialise-and-scan route)."""
    rng = np.random.default_rng(23)
    for kinds in ((0, 1, 3), (0, 1, 2, 3), (2,), (3,), (0,)):
        for case in range(10):
            stream, lens = _length_runs(rng, int(rng.integers(1, 40)), kinds)
            assert lens.min() >= 0
            total = int(lens.sum())
            data = bytes(rng.integers(32, 127, size=total, dtype=np.uint8))
            for n in (len(lens), max(1, len(lens) // 2)):
                rc_cpu, off_ref, out_ref = oracle.string_direct(stream, data, n)
                rc_gpu, off, out = gpu.string_direct(stream, data, n)
                assert rc_cpu == 0 and rc_gpu == 0, (kinds, case, rc_cpu, rc_gpu)
                assert (off_ref == np.concatenate([[0], np.cumsum(lens[:n])])).all()
                assert (off == off_ref).all(), (kinds, case, n)
                assert out.tobytes() == out_ref.tobytes()


def test_many_groups_and_tiles(gpu, oracle):
    """More than 4096 groups of 512 values (several tiles of the group-sum scan) and several tiles
    of string bytes."""
    rng = np.random.default_rng(24)
    n = 512 * 4096 * 2 + 777
    lens = rng.integers(0, 4, size=n).astype(np.uint64)
    ls = _length_stream(lens, 2)
    total = int(lens.sum())
    data = bytes(rng.integers(32, 127, size=total, dtype=np.uint8))
    rc, off, out = gpu.string_direct(ls, data, n)
    assert rc == 0
    assert (off == np.concatenate([[0], np.cumsum(lens.astype(np.int64))])).all()
    assert out.tobytes() == data
    # one byte short
    rc, _, _ = gpu.string_direct(ls, data[:-1], n)
    assert rc == OUT_OF_SPEC


def test_length_sums_that_overflow(gpu, oracle):
    """Lengths whose 64-bit sum wraps to something small must still be refused."""
    big = np.array([2**63, 2**63, 5], dtype=np.uint64)
    ls = enc.direct(big, 64, False)
    data = b"hello"
    rc_cpu, _, _ = oracle.string_direct(ls, data, 3)
    rc_gpu, _, _ = gpu.string_direct(ls, data, 3)
    assert rc_cpu == OUT_OF_SPEC and rc_gpu == OUT_OF_SPEC
    many = np.full(512, 2**55, dtype=np.uint64)          # 512 * 2^55 = 2^64
    ls = enc.direct(many, 56, False)
    rc_gpu, _, _ = gpu.string_direct(ls, data, 512)
    assert rc_gpu == OUT_OF_SPEC


@pytest.mark.parametrize("width", [1, 2, 3, 5, 6, 7, 8, 11, 13, 15, 16, 17, 24])
def test_full_runs_of_every_width_four_windows_to_a_wavefront(gpu, oracle, width):
    """LENGTH streams of full Direct runs (what writers make of string lengths): windows of one run
    each, summed four to a wavefront when the fields are at most 16 bits wide -- every residue of
    the run's start modulo 16 bytes comes by with the odd widths -- and one by one otherwise; the
    number of values wanted ends inside a run, at a run's end and inside a group of four."""
    rng = np.random.default_rng(300 + width)
    n_runs = 41
    lens = rng.integers(0, min(3, 2**width), size=512 * n_runs - 100).astype(np.uint64)   # the last run is short
    ls = _length_stream(lens, width)
    data = bytes(rng.integers(32, 127, size=int(lens.sum()), dtype=np.uint8))
    want = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    for n in (len(lens), 512 * 40, 512 * 38 + 1, 512 * 37 - 1, 512 * 4, 700, 512, 5):
        rc_gpu, off, out = gpu.string_direct(ls, data[:int(want[n])], n)
        assert rc_gpu == 0, (width, n)
        assert (off == want[:n + 1]).all(), (width, n)
        assert out.tobytes() == data[:int(want[n])]
    rc_cpu, off_ref, _ = oracle.string_direct(ls, data, len(lens))
    assert rc_cpu == 0 and (off_ref == want).all()


def test_a_window_of_another_kind_among_full_runs(gpu, oracle):
    """Full Direct runs with a ShortRepeat or Delta run of 512 values here and there: a wavefront
    whose four windows are not all of the simple kind takes them one by one, with the same sums."""
    rng = np.random.default_rng(31)
    chunks, lens = [], []
    for k in range(64):
        if k % 5 == 3:
            chunks += [enc.short_repeat(2, 8, False)] * 64          # 512 values in 64 runs
            lens.append(np.full(512, 2, dtype=np.int64))
        elif k % 7 == 6:
            chunks.append(enc.delta(1, 0, 510, 0, False))             # 512 times the value 1
            lens.append(np.full(512, 1, dtype=np.int64))
        else:
            v = rng.integers(0, 8, size=512).astype(np.uint64)
            chunks.append(enc.direct(v, 3, False))
            lens.append(v.astype(np.int64))
    ls, lens = b"".join(chunks), np.concatenate(lens)
    data = bytes(rng.integers(32, 127, size=int(lens.sum()), dtype=np.uint8))
    want = np.concatenate([[0], np.cumsum(lens)])
    rc_cpu, off_ref, _ = oracle.string_direct(ls, data, len(lens))
    assert rc_cpu == 0 and (off_ref == want).all()
    for n in (len(lens), len(lens) - 300, 512 * 9 + 17):
        rc_gpu, off, out = gpu.string_direct(ls, data[:int(want[n])], n)
        assert rc_gpu == 0 and (off == want[:n + 1]).all(), n

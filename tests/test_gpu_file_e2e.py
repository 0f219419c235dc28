"""The string configuration as a file, end to end: an uncompressed ORC file of about a gigabyte,
decoded stripe by stripe through decode_file_shard on three contexts with the columns left on the
GPU, against pyarrow's reader (the stripe loop of tests/it/main.rs:8-24 and deserialize_str_array,
tests/it/deserialize.rs:203-222)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_string_file_of_a_gigabyte_against_pyarrow(tmp_path):
    po = pytest.importorskip("pyarrow.orc")
    from bench_util import orc_write
    from orc_format_amd import Context, read, shard
    path = str(tmp_path / "strings.orc")
    n_stripes, per_stripe = 16, 2_000_000
    templates, size = orc_write.write_string_file(path, n_stripes, per_stripe, seed=46, distinct=3)
    assert size > 1_000_000_000
    md = read.read_metadata(path)
    assert len(md.stripes) == n_stripes
    contexts = [Context(0) for _ in range(3)]
    entered = {}

    def on_stripe(context, stripe, cols):
        col = cols[0]
        assert col.num_rows == per_stripe and col.num_values == per_stripe
        offsets = context.copy_to_host(col.offsets_ptr, col.num_slots + 1, np.int64)
        data = context.copy_to_host(col.bytes_ptr, col.num_bytes, np.uint8)
        valid = context.copy_to_host(col.validity_ptr, col.num_rows, np.uint8)
        col.free()
        entered.setdefault(id(context), []).append(stripe)
        return offsets, data, valid

    res = shard.decode_file_shard(contexts[0], md, 0, 1, columns=[1], row_aligned=False,
                                  extra_contexts=contexts[1:], device=True, on_stripe=on_stripe)
    assert [r[0] for r in res] == list(range(n_stripes))
    assert [r[1] for r in res] == [i * per_stripe for i in range(n_stripes)]
    assert sorted(len(v) for v in entered.values()) == [5, 5, 6]   # every context took its share
    f = po.ORCFile(path)
    for stripe, _, (offsets, data, valid) in res:
        arr = f.read_stripe(stripe).column(0)
        arr = arr.combine_chunks() if hasattr(arr, "combine_chunks") else arr
        assert arr.null_count == 0 and valid.all()
        bufs = arr.buffers()
        pa_off = np.frombuffer(bufs[1], dtype=np.int32 if str(arr.type) == "string" else np.int64,
                               count=per_stripe + 1, offset=arr.offset * (4 if str(arr.type) == "string" else 8))
        assert (offsets - offsets[0] == (pa_off - pa_off[0]).astype(np.int64)).all()
        pa_data = np.frombuffer(bufs[2], dtype=np.uint8, count=int(pa_off[-1] - pa_off[0]), offset=int(pa_off[0]))
        assert data.size == pa_data.size and (data == pa_data).all()
        # and against what was written
        lengths, want = templates[stripe % len(templates)]
        assert (np.diff(offsets) == lengths.astype(np.int64)).all() and (data == want).all()
    for c in contexts:
        c.close()

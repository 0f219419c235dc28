"""The synthetic string-file writer (bench_util/orc_write.py) writes files that pyarrow's ORC
reader and this package's host reader (read_metadata / read_stripe_footer / read_stripe_column:
src/read/mod.rs:46-159) both understand, and whose LENGTH streams the oracle decodes."""
import numpy as np
import pytest

from bench_util import orc_write
from orc_format_amd import read
from orc_format_amd.proto import ColumnEncodingKind, StreamKind


def expected_strings(lengths, data):
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    return [data[off[k]:off[k + 1]].tobytes().decode() for k in range(lengths.size)]


@pytest.mark.parametrize("n_stripes,per_stripe", [(1, 7), (3, 512), (5, 3000)])
def test_file_reads_back(tmp_path, oracle, n_stripes, per_stripe):
    po = pytest.importorskip("pyarrow.orc")
    path = str(tmp_path / "s.orc")
    templates, size = orc_write.write_string_file(path, n_stripes, per_stripe, seed=3)
    import os
    assert os.path.getsize(path) == size
    f = po.ORCFile(path)
    assert f.nstripes == n_stripes and f.nrows == n_stripes * per_stripe
    md = read.read_metadata(path)
    assert md.number_of_rows == n_stripes * per_stripe and md.types == [12, 7]
    for i in range(n_stripes):
        lengths, data = templates[i % len(templates)]
        want = expected_strings(lengths, data)
        assert f.read_stripe(i).column(0).to_pylist() == want
        footer = read.read_stripe_footer(md, i)
        column = read.read_stripe_column(md, i, footer, 1)
        assert column.encoding() == ColumnEncodingKind.DirectV2
        assert column.number_of_rows() == per_stripe
        assert bytes(column.get_stream(StreamKind.Data)) == data.tobytes()
        rc, got, consumed = oracle.rlev2(column.get_stream(StreamKind.Length), per_stripe, False)
        assert rc == 0 and (got == lengths.astype(np.int64)).all()

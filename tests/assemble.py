"""Column assembly for tests: the logic of the reference's harness (tests/it/deserialize.rs)
over a pluggable decode backend (the CPU oracle, or the HIP library through its C ABI).

A backend offers rlev2 / boolean / floats / string_direct / string_dictionary with the
signatures of tests/oracle_ctypes.Oracle.
"""
import numpy as np

import orc_pyparse as op

BOOLEAN, BYTE, SHORT, INT, LONG, FLOAT, DOUBLE, STRING = range(8)
VARCHAR, CHAR = 16, 17
DIRECT, DICTIONARY, DIRECT_V2, DICTIONARY_V2 = range(4)


def validity(backend, f, stripe, column):
    """deserialize_validity (deserialize.rs:13-25); a missing PRESENT stream means all valid."""
    rows = f.stripes[stripe].number_of_rows
    present = f.stream(stripe, column, op.PRESENT)
    if present is None:
        return np.ones(rows, dtype=bool)
    rc, bits, _ = backend.boolean(present, rows)
    assert rc == 0, rc
    return bits


def column(backend, f, stripe, col):
    """Returns (validity, dense non-null values) like deserialize_*_array (deserialize.rs:27-222);
    strings come back as a list of str."""
    kind = f.types[col][0]
    valid = validity(backend, f, stripe, col)
    n = int(valid.sum())
    _, encodings = f.stripe_footer(stripe)
    data = f.stream(stripe, col, op.DATA)
    if kind in (SHORT, INT, LONG):
        rc, vals = backend.rlev2(data, n, True)[:2]
        assert rc == 0, rc
        return valid, vals
    if kind == BOOLEAN:
        rc, vals, _ = backend.boolean(data, n)
        assert rc == 0, rc
        return valid, vals
    if kind in (FLOAT, DOUBLE):
        rc, vals = backend.floats(data, n, np.float32 if kind == FLOAT else np.float64)
        assert rc == 0, rc
        return valid, vals
    if kind in (STRING, VARCHAR, CHAR):
        enc, dict_size = encodings[col]
        length = f.stream(stripe, col, op.LENGTH)
        if enc == DIRECT_V2:
            rc, offsets, out = backend.string_direct(length, data if data is not None else b"", n)
            assert rc == 0, rc
        elif enc == DICTIONARY_V2:
            dict_data = f.stream(stripe, col, op.DICTIONARY_DATA) or b""
            rc, dict_offsets, dict_bytes = backend.string_direct(length, dict_data, dict_size)
            assert rc == 0, rc
            rc, idx = backend.rlev2(data, n, False)[:2]
            assert rc == 0, rc
            rc, offsets, out = backend.string_dictionary(dict_offsets, dict_bytes, idx)
            assert rc == 0, rc
        else:
            raise NotImplementedError(enc)
        b = bytes(out)
        return valid, [b[offsets[i]:offsets[i + 1]].decode("utf-8") for i in range(n)]
    raise NotImplementedError(kind)


def scatter(valid, dense, null=None):
    """Row-aligned python list with `null` at invalid rows (for comparison with pyarrow)."""
    out = []
    it = iter(dense.tolist() if isinstance(dense, np.ndarray) else dense)
    for v in valid:
        out.append(next(it) if v else null)
    return out

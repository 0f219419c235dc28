"""Column assembly on the GPU decoder, with the names of the reference's harness
(tests/it/deserialize.rs): validity from PRESENT, then the dense non-null values of the column.

Differences from the harness, both where it departs from the ORC format: a column without a
PRESENT stream is all valid (deserialize.rs:14 returns InvalidKind), and the index count of a
dictionary column is the non-null count (deserialize.rs:160 uses the row count).
The ROW_INDEX entry points of each stream are handed to the decoder when the stripe has them.
"""
import numpy as np

from .context import default_context
from .proto import ColumnEncodingKind, StreamKind


def deserialize_validity(column, ctx=None):
    """deserialize_validity (deserialize.rs:13-25) -> bool array of number_of_rows()."""
    ctx = ctx or default_context()
    rows = column.number_of_rows()
    if not column.has_stream(StreamKind.Present):
        return np.ones(rows, dtype=bool)
    bits, _ = ctx.boolean(column.get_stream(StreamKind.Present), rows,
                          column.entry_points(StreamKind.Present))
    return bits


def deserialize_int_array(column, ctx=None):
    """deserialize_int_array (deserialize.rs:48-82) -> (validity, int64 values)."""
    ctx = ctx or default_context()
    validity = deserialize_validity(column, ctx)
    n = int(validity.sum())
    values = ctx.rlev2(column.get_stream(StreamKind.Data), n, True,
                       column.entry_points(StreamKind.Data))
    return validity, values


def deserialize_bool_array(column, ctx=None):
    """deserialize_bool_array (deserialize.rs:84-104)."""
    ctx = ctx or default_context()
    validity = deserialize_validity(column, ctx)
    n = int(validity.sum())
    values, _ = ctx.boolean(column.get_stream(StreamKind.Data), n,
                            column.entry_points(StreamKind.Data))
    return validity, values


def deserialize_f32_array(column, ctx=None):
    """deserialize_f32_array (deserialize.rs:27-46)."""
    return _deserialize_float(column, np.float32, ctx)


def deserialize_f64_array(column, ctx=None):
    return _deserialize_float(column, np.float64, ctx)


def _deserialize_float(column, dtype, ctx):
    ctx = ctx or default_context()
    validity = deserialize_validity(column, ctx)
    n = int(validity.sum())
    return validity, ctx.floats(column.get_stream(StreamKind.Data), n, dtype)


def deserialize_str_array(column, ctx=None, check_utf8=True):
    """deserialize_str_array (deserialize.rs:203-222) -> (validity, offsets[n + 1], bytes)."""
    ctx = ctx or default_context()
    validity = deserialize_validity(column, ctx)
    n = int(validity.sum())
    encoding = column.encoding()
    if encoding == ColumnEncodingKind.DirectV2:
        data = column.get_stream(StreamKind.Data) if column.has_stream(StreamKind.Data) else b""
        offsets, out = ctx.string_direct(column.get_stream(StreamKind.Length), data, n, check_utf8,
                                         column.entry_points(StreamKind.Length))
    elif encoding == ColumnEncodingKind.DictionaryV2:
        dict_data = (column.get_stream(StreamKind.DictionaryData)
                     if column.has_stream(StreamKind.DictionaryData) else b"")
        dict_offsets, dict_bytes = ctx.string_direct(column.get_stream(StreamKind.Length),
                                                     dict_data, column.dictionary_size(), check_utf8)
        indices = ctx.rlev2(column.get_stream(StreamKind.Data), n, False,
                            column.entry_points(StreamKind.Data))
        offsets, out = ctx.string_dictionary(dict_offsets, dict_bytes, indices)
    else:
        from . import _native as N
        from .error import OrcError
        raise OrcError(N.UNSUPPORTED, "column encoding %s" % encoding.name)
    return validity, offsets, out


def strings(offsets, data):
    """The values of a string column as a list of str."""
    b = bytes(data)
    return [b[offsets[i]:offsets[i + 1]].decode("utf-8") for i in range(len(offsets) - 1)]

"""The enums of the ORC protobuf schema used by the read path (reference: src/proto.rs)."""
import enum


class StreamKind(enum.IntEnum):  # proto::stream::Kind, src/proto.rs:127-136
    Present = 0
    Data = 1
    Length = 2
    DictionaryData = 3
    DictionaryCount = 4
    Secondary = 5
    RowIndex = 6
    BloomFilter = 7


class ColumnEncodingKind(enum.IntEnum):  # proto::column_encoding::Kind, src/proto.rs:149-154
    Direct = 0
    Dictionary = 1
    DirectV2 = 2
    DictionaryV2 = 3


class CompressionKind(enum.IntEnum):  # src/proto.rs:281-286
    NONE = 0
    Zlib = 1
    Snappy = 2
    Lzo = 3


class TypeKind(enum.IntEnum):  # proto::type::Kind, src/proto.rs:184-203
    Boolean = 0
    Byte = 1
    Short = 2
    Int = 3
    Long = 4
    Float = 5
    Double = 6
    String = 7
    Binary = 8
    Timestamp = 9
    List = 10
    Map = 11
    Struct = 12
    Union = 13
    Decimal = 14
    Date = 15
    Varchar = 16
    Char = 17

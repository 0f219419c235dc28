"""MI355X-native ORC stripe decoder.

Python face of liborcd.so (include/orcd.h).  The module tree follows the `orc-format` crate:
`read` (file tail, stripe footers, `Column`), `read.decode` (the stream decoders), `proto`
(the enums of the ORC protobuf schema that the read path uses), `error`.
"""
from . import error, proto  # noqa: F401
from .context import Context, default_context  # noqa: F401

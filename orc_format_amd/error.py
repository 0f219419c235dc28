"""Error type of the decoder: the variants of `enum Error` (reference: src/error.rs:6-21), plus
the conditions the reference answers with a panic."""
from . import _native as N


class OrcError(Exception):
    """Raised for any non-zero status of the C ABI; `.code` is the ORCD_* status."""

    NAMES = {
        N.OUT_OF_SPEC: "OutOfSpec", N.INVALID_UTF8: "InvalidUtf8",
        N.INVALID_COLUMN: "InvalidColumn", N.INVALID_KIND: "InvalidKind",
        N.DECODE_FLOAT: "DecodeFloat", N.DECOMPRESSION: "Decompression",
        N.INVALID_PROTO: "InvalidProto", N.UNSUPPORTED: "Unsupported", N.HIP_ERROR: "HipError",
        N.INVALID_ARGUMENT: "InvalidArgument",
        N.NO_MEMORY: "NoMemory",
    }

    def __init__(self, code, detail=""):
        self.code = code
        self.variant = self.NAMES.get(code, "Unknown(%d)" % code)
        msg = N.lib().orcd_strerror(code).decode()
        super().__init__("%s: %s%s" % (self.variant, msg, (" (" + detail + ")") if detail else ""))


def check(code, ctx=None):
    if code != N.OK:
        detail = ""
        if code == N.HIP_ERROR and ctx is not None:
            detail = N.lib().orcd_ctx_last_hip_error(ctx).decode()
        raise OrcError(code, detail)

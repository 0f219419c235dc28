"""Adapter giving the HIP decoder (through its C ABI) the call signatures of the CPU oracle, so
that the same column-assembly harness and the same assertions run against both."""
import numpy as np

from orc_format_amd import default_context
from orc_format_amd.error import OrcError


class GpuBackend:
    def __init__(self, ctx=None, use_entry_points=None):
        self.ctx = ctx or default_context()

    def rlev2(self, stream, n, signed, entry_points=None):
        try:
            return 0, self.ctx.rlev2(stream, n, signed, entry_points), None
        except OrcError as e:
            return e.code, np.empty(0, dtype=np.int64), None

    def runs(self, stream, n, entry_points=None):
        try:
            return 0, self.ctx.rlev2_runs(stream, n, False, entry_points)
        except OrcError as e:
            return e.code, []

    def byte_rle(self, stream, n, entry_points=None):
        try:
            return 0, self.ctx.byte_rle(stream, n, entry_points)
        except OrcError as e:
            return e.code, np.empty(0, dtype=np.uint8)

    def boolean(self, stream, n, entry_points=None):
        try:
            out, ones = self.ctx.boolean(stream, n, entry_points)
            return 0, out, ones
        except OrcError as e:
            return e.code, np.empty(0, dtype=bool), 0

    def floats(self, stream, n, dtype):
        try:
            return 0, self.ctx.floats(stream, n, dtype)
        except OrcError as e:
            return e.code, np.empty(0, dtype=dtype)

    def string_direct(self, length_stream, data, n, check_utf8=True):
        try:
            offsets, out = self.ctx.string_direct(length_stream, data, n, check_utf8)
            return 0, offsets, out
        except OrcError as e:
            return e.code, np.empty(0, dtype=np.int64), np.empty(0, dtype=np.uint8)

    def string_dictionary(self, dict_offsets, dict_bytes, indices):
        try:
            offsets, out = self.ctx.string_dictionary(dict_offsets, dict_bytes, indices)
            return 0, offsets, out
        except OrcError as e:
            return e.code, np.empty(0, dtype=np.int64), np.empty(0, dtype=np.uint8)

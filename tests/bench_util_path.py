"""Makes the repository root importable from the tests and re-exports the benchmark's generators."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from bench_util import synth  # noqa: E402,F401

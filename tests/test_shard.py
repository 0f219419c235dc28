"""The multi-GPU path on CPUs: world_size-2 gloo run of the stripe sharding used by bench.py and
by multi-stripe files.  Each rank reads and decodes only its stripes (the CPU oracle stands in for
the GPU decoder here); together the ranks must reproduce the whole column, with no exchange of
column data between them."""
import os
import sys

import numpy as np
import pytest

from orc_format_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_assign_stripes_is_a_balanced_partition():
    lengths = [5, 9, 1, 7, 7, 3, 8, 2, 6, 4]
    for world in (1, 2, 3, 8, 16):
        parts = shard.assign_stripes(lengths, world)
        assert len(parts) == world
        assert sorted(i for p in parts for i in p) == list(range(len(lengths)))
        loads = [sum(lengths[i] for i in p) for p in parts]
        assert max(loads) - min(l for l in loads if l or world <= len(lengths)) <= max(lengths)
    assert shard.assign_stripes([], 4) == [[], [], [], []]
    assert shard.row_offsets([3, 4, 5]) == [0, 3, 7]
    # equal stripes (the synthetic benchmark) spread evenly
    parts = shard.assign_stripes([10] * 16, 8)
    assert all(len(p) == 2 for p in parts)


def _worker(rank, world, port, path, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import oracle_ctypes
    from orc_format_amd import read
    from orc_format_amd.proto import StreamKind
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oracle = oracle_ctypes.load()
        md = read.read_metadata(path)
        mine = shard.stripes_of_rank(md, rank, world)
        rows = 0
        decoded = {}
        for s in mine:
            footer = read.read_stripe_footer(md, s)
            column = read.read_stripe_column(md, s, footer, 1)  # "w17": int64, no nulls
            n = column.number_of_rows()
            rc, vals, _ = oracle.rlev2(bytes(column.get_stream(StreamKind.Data)), n, True)
            assert rc == 0
            decoded[s] = vals
            rows += n
        total_rows = shard.sum_over_ranks(rows, dist)
        slowest = shard.max_over_ranks(float(rank + 1), dist)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)  # control plane only: which stripes, not the data
        results[rank] = (mine, decoded, total_rows, slowest, gathered)
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo(golden_dir, tmp_path):
    mp = pytest.importorskip("torch.multiprocessing")
    pa_orc = pytest.importorskip("pyarrow.orc")
    path = os.path.join(golden_dir, "mixed_speed.orc")
    world = 2
    manager = mp.Manager()
    results = manager.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, path, results), nprocs=world, join=True)
    assert sorted(results.keys()) == [0, 1]
    table = pa_orc.ORCFile(path).read()
    expect = np.array(table.column("w17").to_pylist(), dtype=np.int64)
    all_stripes = sorted(s for r in range(world) for s in results[r][0])
    n_stripes = len(all_stripes)
    assert all_stripes == list(range(n_stripes)) and n_stripes > 2
    assert not set(results[0][0]) & set(results[1][0])
    for r in range(world):
        assert results[r][2] == len(expect)          # rows summed over ranks
        assert results[r][3] == float(world)         # max over ranks
        assert results[r][4] == [results[0][0], results[1][0]]
    merged = {}
    for r in range(world):
        merged.update(results[r][1])
    got = np.concatenate([merged[s] for s in range(n_stripes)])
    assert (got == expect).all()


class _ExclusiveContext:
    """Stands in for a decode context: records the stripes it was given and fails if a second
    thread enters it while a call is in progress (a real context has no lock)."""

    def __init__(self, delays):
        import threading
        self.lock = threading.Lock()
        self.delays = delays
        self.seen = []
        self.ahead = []
        self.threads = set()

    def stripe_prefetch(self, metadata, stripe, footer, columns):
        import threading
        assert self.lock.acquire(blocking=False), "context entered by two threads at once"
        self.threads.add(threading.get_ident())
        self.ahead.append(stripe)
        self.lock.release()

    def decode_stripe(self, metadata, stripe, footer, columns, check_utf8=True, row_aligned=False, device=False):
        import threading
        import time
        assert self.lock.acquire(blocking=False), "context entered by two threads at once"
        try:
            self.threads.add(threading.get_ident())
            time.sleep(self.delays[stripe % len(self.delays)])
            self.seen.append(stripe)
            return [{"stripe": stripe, "columns": list(columns)}]
        finally:
            self.lock.release()


def test_decode_file_shard_keeps_a_context_on_one_thread(golden_dir):
    """Stripes of uneven duration on two and three contexts: every context is entered by one thread
    only, never re-entered, and the results come back in the order of the rank's stripes."""
    from orc_format_amd import read
    md = read.read_metadata(os.path.join(golden_dir, "mixed_speed.orc"))
    n = len(md.stripes)
    assert n >= 4
    delays = [0.05, 0.001, 0.02, 0.001, 0.03]
    for n_ctx in (2, 3):
        ctxs = [_ExclusiveContext(delays) for _ in range(n_ctx)]
        for _ in range(3):   # several passes: more jobs than one file's stripes
            for c in ctxs:
                c.threads.clear()
            out = shard.decode_file_shard(ctxs[0], md, 0, 1, columns=[1], extra_contexts=ctxs[1:])
            assert all(len(c.threads) == 1 for c in ctxs)   # one thread per context
            assert [o[0] for o in out] == list(range(n))
            assert [o[1] for o in out] == shard.row_offsets([s.number_of_rows for s in md.stripes])
            assert all(o[2][0]["stripe"] == o[0] for o in out)
        for k, c in enumerate(ctxs):
            assert sorted(set(c.seen)) == list(range(k, n, n_ctx))
            # every stripe was asked for ahead of its decode, by the thread that decodes it
            assert c.ahead == c.seen
    # without read-ahead no hint is given
    quiet = _ExclusiveContext(delays)
    shard.decode_file_shard(quiet, md, 0, 1, columns=[1], prefetch=False)
    assert quiet.ahead == [] and quiet.seen == list(range(n))

    class Failing(_ExclusiveContext):
        def decode_stripe(self, *a, **kw):
            raise RuntimeError("boom")
    with pytest.raises(RuntimeError):
        shard.decode_file_shard(_ExclusiveContext(delays), md, 0, 1, columns=[1],
                                extra_contexts=[Failing(delays)])


def _helper_budget_probe(path):
    """Child process body: reads the columns of a zlib file from several threads at once and reports
    the helper budget, the peak of helpers at work at once, and whether the bytes were the same as a
    single-threaded read's."""
    return r'''
import ctypes as C, os, sys, threading
sys.path.insert(0, %r)
from orc_format_amd import _native as N, read
lib = N.lib()
path = %r

def streams():
    md = read.read_metadata(path)
    out = []
    for si in range(len(md.stripes)):
        ft = read.read_stripe_footer(md, si)
        for ci in (1, 2):
            col = read.read_stripe_column(md, si, ft, ci)
            out.append(bytes(col.get_stream(1)))
    return out

got = [None] * 6
ths = [threading.Thread(target=lambda k=k: got.__setitem__(k, streams())) for k in range(6)]
for t in ths: t.start()
for t in ths: t.join()
b, u, p = C.c_int(), C.c_int(), C.c_int()
lib.orcd_helper_threads(C.byref(b), C.byref(u), C.byref(p))
print(b.value, u.value, p.value, int(all(g == got[0] for g in got)))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path)


@pytest.mark.parametrize("env,want_budget", [
    ({"ORCD_HELPER_THREADS": "3"}, 3),
    ({"OMP_NUM_THREADS": "16", "LOCAL_WORLD_SIZE": "8"}, 2),
    ({"OMP_NUM_THREADS": "16", "WORLD_SIZE": "2"}, 8),
    ({"ORCD_HELPER_THREADS": "1"}, 1),
])
def test_helper_threads_stay_within_the_process_budget(tmp_path, env, want_budget):
    """Six threads decompress a zlib file's chunks at once (every chunk list would take up to 16
    threads of its own): the helpers at work never exceed the budget -- the one number an 8-rank
    node sets per process -- and none is left at work afterwards."""
    import subprocess
    import sys
    pa = pytest.importorskip("pyarrow")
    orc = pytest.importorskip("pyarrow.orc")
    rng = np.random.default_rng(3)
    path = os.path.join(tmp_path, "chunks.orc")
    orc.write_table(pa.table({"a": pa.array(rng.integers(0, 2**40, size=400_000)),
                              "b": pa.array(rng.standard_normal(400_000))}),
                    path, compression="zlib", compression_block_size=65536)
    full = dict(os.environ, ORCD_INFLATE_PARALLEL_MIN="1")
    for k in ("ORCD_HELPER_THREADS", "OMP_NUM_THREADS", "LOCAL_WORLD_SIZE", "WORLD_SIZE"):
        full.pop(k, None)
    full.update(env)
    out = subprocess.run([sys.executable, "-c", _helper_budget_probe(path)], env=full, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    budget, in_use, peak, same = map(int, out.stdout.split())
    assert budget == want_budget
    assert in_use == 0 and 1 <= peak <= budget
    assert same == 1

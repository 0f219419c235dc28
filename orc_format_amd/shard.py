"""Stripe sharding across the GPUs of one node.

Stripes are self-contained (own footer, streams and row count -- reference: src/read/mod.rs:95-159,
src/proto.rs:206-217), so every rank decodes a disjoint subset of them and nothing is exchanged on
the data path.  torch.distributed (RCCL on GPUs, gloo on CPUs) is used only to agree on timing.
"""
import heapq


def assign_stripes(data_lengths, world_size):
    """Longest-processing-time-first assignment of stripes to ranks by data length.
    Returns a list (one per rank) of ascending stripe indices; every stripe appears once."""
    if world_size <= 0:
        raise ValueError("world_size must be positive")
    loads = [(0, r) for r in range(world_size)]
    heapq.heapify(loads)
    out = [[] for _ in range(world_size)]
    for idx in sorted(range(len(data_lengths)), key=lambda i: (-data_lengths[i], i)):
        load, r = heapq.heappop(loads)
        out[r].append(idx)
        heapq.heappush(loads, (load + data_lengths[idx], r))
    return [sorted(x) for x in out]


def row_offsets(rows_per_stripe):
    """First row of every stripe: exclusive prefix sum of the stripes' row counts."""
    out, acc = [], 0
    for n in rows_per_stripe:
        out.append(acc)
        acc += n
    return out


def stripes_of_rank(metadata, rank, world_size):
    """The stripes of an open file (read.FileMetadata) that `rank` decodes."""
    lengths = [s.data_length for s in metadata.stripes]
    return assign_stripes(lengths, world_size)[rank]


def max_over_ranks(value, dist=None, device=None):
    """Maximum of a float over all ranks (the job's elapsed time); a no-op without a process
    group."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, dist=None, device=None):
    """The ranks' values of a float, in rank order (every rank gets the list)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    import torch
    n, r = dist.get_world_size(), dist.get_rank()
    t = torch.zeros(n, dtype=torch.float64, device=device or "cpu")
    t[r] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def sum_over_ranks(value, dist=None, device=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def decode_file_shard(ctx, metadata, rank, world_size, columns=None, row_aligned=True,
                      check_utf8=True, extra_contexts=(), device=False, on_stripe=None, prefetch=True):
    """The stripes of an open file that fall to `rank`, decoded on that rank's context (one
    orcd_decode_stripe call per stripe).  Returns a list of (stripe index, first row of the stripe
    in the file, [column dicts as Context.decode_stripe returns them]); the ranks' lists together
    hold every stripe once, and the first rows say where each piece belongs -- no exchange is
    needed to assemble the table (reference: the stripe loop of tests/it/main.rs:8-24, which this
    spreads over ranks).

    extra_contexts: more contexts on the same GPU.  A context belongs to one thread at a time
    (it has no lock: its buffers, pinned words and stamps are per call), so every context gets one
    host thread of its own, which decodes that context's share of the stripes -- every
    len(contexts)-th of the rank's -- one after the other; one stripe's read and copy then overlap
    another's decode on a different context.

    prefetch: every context has its next stripe read and copied to the device in the background
    (orcd_stripe_prefetch) while it decodes the current one, so that even a single context overlaps
    the feed with the decode.

    device: the columns stay on the GPU (context.DeviceColumn objects instead of dicts).
    on_stripe(context, stripe, columns): called by the decoding thread after each stripe, e.g. to
    copy the offsets of a string column to the host, or to free the columns when only the rate is
    wanted; what it returns takes the columns' place in the result."""
    from . import read
    if columns is None:
        columns = list(range(1, len(metadata.types)))
    first_rows = row_offsets([s.number_of_rows for s in metadata.stripes])
    mine = stripes_of_rank(metadata, rank, world_size)
    contexts = [ctx] + list(extra_contexts)

    footers = {}

    def footer_of(stripe):
        if stripe not in footers:
            footers[stripe] = read.read_stripe_footer(metadata, stripe)
        return footers[stripe]

    def ahead(context, stripe):
        if prefetch:
            context.stripe_prefetch(metadata, stripe, footer_of(stripe), columns)

    def one(context, stripe, following=None):
        footer = footer_of(stripe)
        if following is not None:
            ahead(context, following)   # read beside this stripe's decode
        cols = context.decode_stripe(metadata, stripe, footer, columns, check_utf8=check_utf8,
                                     row_aligned=row_aligned, device=device)
        if on_stripe is not None:
            cols = on_stripe(context, stripe, cols)
        return (stripe, first_rows[stripe], cols)

    def share(k):   # what contexts[k] decodes, in order
        return list(range(k, len(mine), len(contexts)))
    results = [None] * len(mine)
    errors = []

    def drain(k):   # the only thread that ever enters contexts[k]
        try:
            idx = share(k)
            if idx:
                ahead(contexts[k], mine[idx[0]])
            for n, i in enumerate(idx):
                if errors:
                    return
                following = mine[idx[n + 1]] if n + 1 < len(idx) else None
                results[i] = one(contexts[k], mine[i], following)
        except BaseException as e:  # handed to the caller's thread below
            errors.append(e)
    if len(contexts) == 1:
        drain(0)
        if errors:
            raise errors[0]
        return results
    import threading
    threads = [threading.Thread(target=drain, args=(k,)) for k in range(len(contexts))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results

"""Headline benchmark: device-resident decode of a synthetic 1B-row int64 ORC column.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the RLE v2 decode path (header scan, segment prefix sums, run-table
emission, value expansion) over one stripe set of `--rows` values that is already resident in
HBM, through the C ABI of liborcd.so.  Stripes are independent, so with N GPUs every rank decodes
its own stripe set of the same size (weak scaling, no data-path collective); the only
torch.distributed calls are the barrier and the max-over-ranks of the elapsed time.  Rank 0
prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak (spec), GB/s
REDUCE_DEVICE = "cuda"     # where the ranks' elapsed times are reduced (main() sets it)

WORKLOADS = {
    # name: (generator, kwargs, signed, description)
    "direct_w17": ("direct_w17", {}, True,
                   "RLE v2 Direct, width 17, runs of 512, signed int64 (zigzag), seed 42"),
    "delta_w10": ("delta_monotonic", {"width": 10}, True,
                  "RLE v2 Delta, 10-bit deltas, runs of 512, monotonic int64, seed 43"),
    "delta_w16": ("delta_monotonic", {"width": 16}, True,
                  "RLE v2 Delta, 16-bit deltas, runs of 512, monotonic int64, seed 43"),
    "delta_w24": ("delta_monotonic", {"width": 24}, True,
                  "RLE v2 Delta, 24-bit deltas, runs of 512, monotonic int64, seed 43"),
    "delta_w32": ("delta_monotonic", {"width": 32}, True,
                  "RLE v2 Delta, 32-bit deltas, runs of 512, monotonic int64, seed 43"),
    "mixed_short_runs": ("mixed_short_runs", {}, False,
                         "RLE v2 mix of ShortRepeat (3..10 copies) and Direct width 4 runs of 1..64 "
                         "values, unsigned, seed 44: every 512-value window holds about 25 runs"),
    "mixed_short_runs_sparse": ("mixed_short_runs", {"runs_per_entry": 512}, False,
                                "the same mix with an entry point every 512 runs (about every 10 000 "
                                "values, the ROW_INDEX stride of a real file)"),
    # direct-encoded string column: LENGTH (unsigned RLE v2 Direct, width 6) + DATA
    "strings_w6": ("string_lengths_w6", {}, False,
                   "string column, lengths U[8,56] as RLE v2 Direct width 6 runs of 512 (seed 46), "
                   "ASCII DATA; offsets + copied bytes + UTF-8 validation"),
    # PRESENT / boolean DATA: byte-RLE (literal groups and runs) -> one byte per value + count
    "boolean_mix": ("byte_rle_mix", {}, False,
                    "boolean stream, byte-RLE mix of 128-byte literal groups (70%) and 130-byte runs "
                    "of 0xff (30%), seed 45; one byte per value + number of set bits"),
}
DEFAULT_ROWS = {"strings_w6": 250_000_000, "boolean_mix": 4_000_000_000, "mixed_short_runs": 500_000_000,
                "mixed_short_runs_sparse": 500_000_000}


EXTRA_WORKLOADS = ["delta_w10", "delta_w16", "mixed_short_runs", "boolean_mix", "strings_w6", "strings_file",
                   "real_file_direct", "real_file_delta", "mixed_table"]
FILE_WORKLOADS = ["strings_file", "zlib_file", "real_file_direct", "real_file_delta", "mixed_table"]
# strings_file: the string configuration as an ORC file, end to end (read + H2D + decode + D2H of
# the offsets); stripes of 2 000 000 strings, about 64 MB of DATA each
FILE_STRIPES, FILE_STRINGS_PER_STRIPE = 32, 2_000_000
FILE_CONTEXTS = int(os.environ.get("ORCD_BENCH_FILE_CONTEXTS", "4"))
FILE_PREFETCH = os.environ.get("ORCD_BENCH_FILE_PREFETCH", "1") != "0"   # stripes read ahead in the background


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: starts N fresh rank processes (this
    process has not touched the GPU and never does), one per GPU, that meet over RCCL on
    127.0.0.1, and exits with the worst of their exit codes.  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [p.wait() for p in procs]
    sys.exit(max(abs(c) for c in codes))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=None,
                    help="values per GPU per step (default 1e9; 2.5e8 strings for strings_w6)")
    ap.add_argument("--workload", default="direct_w17", choices=sorted(WORKLOADS) + FILE_WORKLOADS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the `workloads` object (the other configurations, shorter runs)")
    ap.add_argument("--end-to-end", action="store_true",
                    help="also measure host -> device -> decode -> host (reported separately)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="target duration of the CPU baseline sample")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)   # before anything here touches a GPU

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    # ORCD_BENCH_REHEARSAL=1: the ranks share GPU 0 and meet over gloo, to exercise the multi-rank
    # code on a one-GPU box; its numbers mean nothing and the result line says so
    rehearsal = os.environ.get("ORCD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    global REDUCE_DEVICE
    REDUCE_DEVICE = "cpu" if rehearsal else "cuda"

    explicit_rows = args.rows
    result = run_workload(args, args.workload, rank, world, local_rank, dist, torch, np)
    # The other configurations of BASELINE.json beside the headline, as shorter runs of the same
    # code: value, roofline of the dominant kernel and CPU baseline each (one GPU, default sizes).
    if (args.workload == "direct_w17" and world == 1 and explicit_rows is None
            and not args.headline_only):
        result["workloads"] = {}
        for wl in EXTRA_WORKLOADS:
            sub = argparse.Namespace(**vars(args))
            sub.rows, sub.steps, sub.warmup = None, min(args.steps, 10), min(args.warmup, 2)
            sub.cpu_seconds, sub.end_to_end = min(args.cpu_seconds, 4.0), False
            torch.cuda.empty_cache()
            try:
                r = run_workload(sub, wl, rank, world, local_rank, dist, torch, np)
            except ImportError as e:   # (the file workloads need pyarrow's writer on the host)
                result["workloads"][wl] = {"skipped": "%s: %s" % (type(e).__name__, e)}
                continue
            keep = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "roofline",
                                      "phases_ms", "decoded_frac_of_hbm_peak", "cpu_baseline",
                                      "zero_copy_variant", "decoded_GBps", "path", "per_stripe_ms", "skipped") if k in r}
            if wl not in ("strings_file", "mixed_table"):
                keep.pop("metric", None)  # (the headline's metric)
            keep["workload"] = r["config"]["workload"]
            result["workloads"][wl] = keep
        if world == 1:
            # the compressed path, when the host can write such a file (pyarrow): beside the rest,
            # never in their way
            try:
                sub = argparse.Namespace(**vars(args))
                sub.rows, sub.steps, sub.warmup = None, 3, 1
                r = run_workload(sub, "zlib_file", rank, world, local_rank, dist, torch, np)
                result["workloads"]["zlib_file"] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps",
                                                                      "dtype", "decoded_GBps")}
                result["workloads"]["zlib_file"]["workload"] = r["config"]["workload"]
            except Exception as e:  # noqa: BLE001
                result["workloads"]["zlib_file"] = {"skipped": "%s: %s" % (type(e).__name__, e)}
    if args.workload == "direct_w17" and world > 1 and explicit_rows is None and not args.headline_only:
        # the feed path on every rank at once: each rank reads and decodes a file of its own, so the
        # first run on a whole node shows at once what the ranks' readers, copy threads and PCIe
        # links do to each other (per_rank_file_GBps; the helper threads of each rank come out of
        # its share of the host's CPUs, orcd_helper_threads)
        sub = argparse.Namespace(**vars(args))
        sub.rows, sub.steps, sub.warmup = None, min(args.steps, 3), 1
        r = run_workload(sub, "strings_file", rank, world, local_rank, dist, torch, np)
        import ctypes as C
        from orc_format_amd import _native
        budget, peak = C.c_int(), C.c_int()
        _native.lib().orcd_helper_threads(C.byref(budget), None, C.byref(peak))
        result["strings_file"] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "decoded_GBps",
                                                    "per_rank_file_GBps")}
        result["strings_file"]["helper_threads_rank0"] = {"budget": budget.value, "peak": peak.value}
        result["strings_file"]["workload"] = r["config"]["workload"]
    if rank == 0:
        if rehearsal:
            result["rehearsal"] = "ranks shared one GPU over gloo: not a measurement"
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_workload(args, workload, rank, world, local_rank, dist, torch, np):
    if workload == "zlib_file":
        args = argparse.Namespace(**vars(args))
        args.workload = workload
        return bench_zlib_file(args, rank, world, local_rank, dist, torch, np)
    if workload == "strings_file":
        args = argparse.Namespace(**vars(args))
        args.workload = workload
        return bench_strings_file(args, rank, world, local_rank, dist, torch, np)
    if workload in ("real_file_direct", "real_file_delta"):
        args = argparse.Namespace(**vars(args))
        args.workload = workload
        return bench_real_file(args, rank, world, local_rank, dist, torch, np)
    if workload == "mixed_table":
        args = argparse.Namespace(**vars(args))
        args.workload = workload
        return bench_mixed_table(args, rank, world, local_rank, dist, torch, np)
    gen_name, gen_kw, signed, desc = WORKLOADS[workload]
    args = argparse.Namespace(**vars(args))
    args.workload = workload
    if args.rows is None:
        args.rows = DEFAULT_ROWS.get(workload, 1_000_000_000)
    if workload == "strings_w6":
        return bench_strings(args, rank, world, local_rank, dist, desc, torch, np)
    if workload == "boolean_mix":
        return bench_boolean(args, rank, world, local_rank, dist, desc, torch, np)
    return bench_rlev2(args, rank, world, local_rank, dist, torch, np)


def bench_rlev2(args, rank, world, local_rank, dist, torch, np):
    from bench_util import synth
    from orc_format_amd import Context

    gen_name, gen_kw, signed, desc = WORKLOADS[args.workload]
    # every rank decodes a different stripe set (seed offset by rank)
    seed = {"direct_w17": 42, "delta_monotonic": 43, "mixed_short_runs": 44}[gen_name] + 1000 * rank
    if gen_name == "mixed_short_runs":
        n_runs = max(1, args.rows // 20)     # a run holds 19.5 values on average
        stream, entries, expect, entry_values = synth.mixed_short_runs(n_runs, seed=seed, device="cuda", **gen_kw)
        rows = int(expect.numel())
        cuts = (entry_values.cpu().numpy(), entries.cpu().numpy())
        full_size = args.rows == DEFAULT_ROWS["mixed_short_runs"] and not gen_kw
    else:
        n_runs = max(1, args.rows // 512)
        rows = n_runs * 512
        full_size = rows == 1_000_000_000 - 1_000_000_000 % 512
        stream, entries, expect = getattr(synth, gen_name)(n_runs, seed=seed, device="cuda", **gen_kw)
        cuts = None
    out = torch.empty(rows, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    stream_bytes = stream.numel()

    ctx = Context(local_rank)

    def step():
        ctx.rlev2_device(stream.data_ptr(), stream_bytes, rows, signed, out.data_ptr(),
                         entries.data_ptr(), entries.numel())

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    out.zero_()   # what the check below compares was written by the timed steps, not by the warm-up
    barrier()
    phase_ms = {1: 0.0, 2: 0.0, 3: 0.0}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for p in phase_ms:  # HIP events recorded on the context's stream around each phase
            phase_ms[p] += max(0.0, ctx.phase_ms(p))  # negative: the phase did not run
    barrier()
    elapsed = time.perf_counter() - t0
    # which kernel expanded the values: the single-launch one for regular columns, or the general one
    # (the general kernel has a variant for columns with more than two runs per window)
    value_kernel = ctx.last_value_kernel()
    from orc_format_amd import shard
    elapsed = shard.max_over_ranks(elapsed, dist, device=REDUCE_DEVICE)

    # parity of what was just timed: the decoded column equals the generator's values
    if not torch.equal(out, expect):
        bad = int((out != expect).sum().item())
        raise SystemExit("bench.py: decoded column differs from the generator in %d values" % bad)

    decoded_bytes = rows * 8
    value = decoded_bytes * args.steps * world / elapsed / 1e9
    k3_ms = phase_ms[3] / args.steps
    algo_bytes = decoded_bytes + stream_bytes
    achieved = algo_bytes / (k3_ms * 1e-3) / 1e9 if k3_ms > 0 else 0.0

    result = {
        "metric": "device-resident decoded GB/s (RLEv2 int64 + strings) at 1/2/4/8 GPUs",
        "value": round(value, 2),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {"workload": "%s: %d rows per GPU per step; %s" % (args.workload, rows, desc),
                   "rows_per_gpu": rows, "stream_bytes_per_gpu": stream_bytes,
                   "entry_points_per_gpu": int(entries.numel()), "parallelism": "stripe-sharded x%d" % world},
        "roofline": {"bound": "hbm", "kernel": value_kernel,
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                     "algorithmic_bytes_per_launch": algo_bytes,
                     "avg_launch_ms": round(k3_ms, 4)},
        "phases_ms": {"header_count_scan_readback": round(phase_ms[1] / args.steps, 4),
                      "run_table_emit": round(phase_ms[2] / args.steps, 4),
                      "value_expansion": round(k3_ms, 4)},
        "decoded_frac_of_hbm_peak": round(value / world / HBM_PEAK_GBPS, 4),
        "in_plus_out_GBps": round((decoded_bytes + stream_bytes) * args.steps * world / elapsed / 1e9, 2),
    }

    # HBM bytes per launch of the value kernel come from PMC counters, which cannot be read from
    # inside this process: they are taken from the newest committed rocprofv3 summary of this same
    # command and workload (separate FETCH_SIZE / WRITE_SIZE passes, see profiles/README.md).
    if full_size:
        attach_traffic(result, args.workload, value_kernel)

    if args.end_to_end:
        result["end_to_end"] = end_to_end(args, ctx, stream, entries, expect, rows, signed, torch)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, stream, rows, stream_bytes, signed, expect, np, cuts)
    return result


def attach_traffic(result, workload, kernel):
    """What the newest committed rocprofv3 summary of this workload (profiles/) says about the
    dominant kernel, beside what this run measured itself:
      traffic           HBM bytes per launch from the PMC passes (counters cannot be read in here);
      frac_profiles     algorithmic bytes / the summary's average duration under --kernel-trace / peak
                        -- another run, usually on another box, and 2-3 % long under the tracer;
      profiles_unprofiled_same_lease   the same bench command without the profiler, run in the
                        lease the summary was taken in, so that box and tracer can be told apart.
    `frac`, `achieved` and `avg_launch_ms` are always this run's own HIP events."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_%s_traffic.json" % workload)))
    if not found:
        return
    with open(found[-1]) as f:
        prof = json.load(f)
    roof = result["roofline"]
    roof["timing_source"] = "HIP events on the context's stream, this run"
    src = os.path.relpath(found[-1], ROOT)

    def bare(name):  # exact kernel, template arguments included ("void orcd::string_bytes_kernel<true, true>")
        return name.replace("void ", "").replace("orcd::", "").strip()
    for name, k in prof.get("kernels", {}).items():
        if bare(name) == kernel and "hbm_bytes_per_launch" in k:
            roof["traffic"] = k["hbm_bytes_per_launch"]
            roof["traffic_source"] = src + " (PMC passes; bytes only, no timing)"
    for name, ns in prof.get("avg_duration_ns", {}).items():
        if bare(name) == kernel and ns > 0:
            roof["avg_launch_ms_profiles"] = round(ns / 1e6, 4)
            roof["frac_profiles"] = round(roof["algorithmic_bytes_per_launch"] / (ns * 1e-9) / 1e9 / HBM_PEAK_GBPS, 4)
            roof["frac_profiles_source"] = src.replace("_traffic.json", "_kernel_stats.csv") + \
                " (rocprofv3 --kernel-trace --stats of the same command, the builder's box)"
    if "unprofiled_same_lease" in prof:
        roof["profiles_unprofiled_same_lease"] = prof["unprofiled_same_lease"]


def bench_strings(args, rank, world, local_rank, dist, desc, torch, np):
    """One step = one direct-encoded string column: LENGTH stream -> offsets[n + 1], DATA ->
    copied bytes, every value checked to be UTF-8 (orcd_string_direct_decode)."""
    from bench_util import synth
    from orc_format_amd import Context, shard

    n_runs = max(1, args.rows // 512)
    n = n_runs * 512
    lstream, entries, lengths = synth.string_lengths_w6(n_runs, seed=46 + 1000 * rank, device="cuda")
    want_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    want_off[0] = 0
    torch.cumsum(lengths, 0, out=want_off[1:])
    del lengths
    total = int(want_off[-1].item())
    data = torch.empty(total, dtype=torch.uint8, device="cuda")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4600 + rank)
    for b0 in range(0, total, 1 << 28):   # printable ASCII, generated in pieces
        b1 = min(total, b0 + (1 << 28))
        data[b0:b1] = torch.randint(32, 127, (b1 - b0,), dtype=torch.uint8, device="cuda", generator=gen)
    offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    out = torch.empty(total, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx = Context(local_rank)

    def step():
        ctx.string_direct_device(lstream.data_ptr(), lstream.numel(), data.data_ptr(), total, n,
                                 offsets.data_ptr(), out.data_ptr(), entries.data_ptr(),
                                 entries.numel(), check_utf8=True)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    offsets.zero_()   # the checks below see what the timed steps wrote, not the warm-up
    out.zero_()
    barrier()
    phase_ms = {1: 0.0, 2: 0.0, 3: 0.0, 4: 0.0}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for p in phase_ms:
            phase_ms[p] += ctx.phase_ms(p)
    barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, dist, device=REDUCE_DEVICE)
    if not torch.equal(offsets, want_off):
        raise SystemExit("bench.py: offsets differ from the prefix sums of the generated lengths")
    if not torch.equal(out, data):
        raise SystemExit("bench.py: copied string bytes differ from DATA")

    decoded_bytes = total + 8 * (n + 1)
    value = decoded_bytes * args.steps * world / elapsed / 1e9
    k_ms = phase_ms[4] / args.steps
    # string bytes read once and written once; the offsets are consulted only for tiles that hold
    # non-ASCII bytes, of which this DATA has none
    algo_bytes = 2 * total
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    result = {
        "metric": "device-resident decoded GB/s (RLEv2 int64 + strings) at 1/2/4/8 GPUs",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "strings_w6: %d strings per GPU per step; %s" % (n, desc),
                   "strings_per_gpu": n, "data_bytes_per_gpu": total,
                   "length_stream_bytes_per_gpu": int(lstream.numel()),
                   "parallelism": "stripe-sharded x%d" % world},
        "roofline": {"bound": "hbm", "kernel": "string_bytes_kernel<true, true>", "achieved": round(achieved, 1),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                     "traffic": None, "algorithmic_bytes_per_launch": algo_bytes,
                     "avg_launch_ms": round(k_ms, 4)},
        "phases_ms": {"header_count_scan_readback": round(phase_ms[1] / args.steps, 4),
                      "run_table_emit": round(phase_ms[2] / args.steps, 4),
                      "length_sums_scan_offsets": round(phase_ms[3] / args.steps, 4),
                      "string_bytes_copy_validate": round(k_ms, 4)},
        "decoded_frac_of_hbm_peak": round(value / world / HBM_PEAK_GBPS, 4),
    }
    if n == DEFAULT_ROWS["strings_w6"] - DEFAULT_ROWS["strings_w6"] % 512:
        attach_traffic(result, args.workload, "string_bytes_kernel<true, true>")
    # Variant reported separately: DATA left in place (the column is offsets into the stripe's own
    # buffer), i.e. offsets + validation only.
    offsets.zero_()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.string_direct_device(lstream.data_ptr(), lstream.numel(), data.data_ptr(), total, n,
                                 offsets.data_ptr(), 0, entries.data_ptr(), entries.numel(),
                                 check_utf8=True)
    barrier()
    zc = shard.max_over_ranks(time.perf_counter() - t0, dist, device=REDUCE_DEVICE)
    if not torch.equal(offsets, want_off):
        raise SystemExit("bench.py: zero-copy offsets differ")
    result["zero_copy_variant"] = {
        "decoded_GBps": round(decoded_bytes * args.steps * world / zc / 1e9, 2),
        "ms_per_step": round(zc / args.steps * 1e3, 4),
        "note": "bytes_out = NULL: offsets written, DATA validated where it lies, no copy"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_ctypes
        oracle = oracle_ctypes.load()

        def run(n_vals):
            n_vals = (n_vals // 512) * 512
            ls = lstream[:n_vals // 512 * 386].cpu().numpy()
            nbytes = int(want_off[n_vals].item())
            host = data[:nbytes].cpu().numpy()
            t0 = time.perf_counter()
            rc, off, got = oracle.string_direct(ls, host, n_vals)
            dt = time.perf_counter() - t0
            assert rc == 0 and (off == want_off[:n_vals + 1].cpu().numpy()).all() and got.tobytes() == host.tobytes()
            return n_vals, nbytes, dt
        n_cal, _, dt_cal = run(min(n, 1 << 20))
        n_s, nbytes, dt = run(int(min(n, max(n_cal, n_cal * args.cpu_seconds / max(dt_cal, 1e-6)))))
        result["cpu_baseline"] = {
            "value": round((nbytes + 8 * (n_s + 1)) / dt / 1e9, 4), "unit": "GB/s", "cores": 1,
            "kind": "port", "sample": "first %d strings (%.1f MB of DATA) of the same column, single "
                                      "thread, %.1f s" % (n_s, nbytes / 1e6, dt)}
    return result


def bench_strings_file(args, rank, world, local_rank, dist, torch, np):
    """One step = one pass over this rank's ORC file of direct-encoded strings, from the file's
    bytes (page cache) to the decoded column: stripe reads, host-to-device copies, LENGTH -> offsets,
    DATA -> copied and UTF-8-checked bytes on the GPU, offsets copied back to the host.  The stripes
    alternate between FILE_CONTEXTS contexts with a host thread each, so that one stripe's read
    and copy overlap another's decode.  This line measures the feed (PCIe, page cache), not a
    kernel: it carries no roofline."""
    import shutil
    import tempfile
    from bench_util import orc_write
    from orc_format_amd import Context, read, shard

    n_stripes = FILE_STRIPES if args.rows is None else max(1, args.rows // FILE_STRINGS_PER_STRIPE)
    tmp = tempfile.mkdtemp(prefix="orcd_bench_")
    try:
        path = os.path.join(tmp, "strings_%d.orc" % rank)
        templates, size = orc_write.write_string_file(path, n_stripes, FILE_STRINGS_PER_STRIPE,
                                                      seed=46 + 1000 * rank)
        md = read.read_metadata(path)
        contexts = [Context(local_rank) for _ in range(FILE_CONTEXTS)]
        want = [np.concatenate([[0], np.cumsum(t[0])]).astype(np.int64) for t in templates]
        decoded = [0]

        def on_stripe(context, stripe, cols):
            col = cols[0]
            offsets = context.copy_to_host(col.offsets_ptr, col.num_slots + 1, np.int64)
            decoded[0] += col.num_bytes + offsets.nbytes
            keep = (stripe, offsets, col.num_bytes)
            col.free()
            return keep

        def one_pass():
            return shard.decode_file_shard(contexts[0], md, 0, 1, columns=[1], row_aligned=False,
                                           check_utf8=True, extra_contexts=contexts[1:], device=True,
                                           on_stripe=on_stripe, prefetch=FILE_PREFETCH)

        def barrier():
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()

        for _ in range(min(args.warmup, 1)):
            one_pass()
        steps = max(1, min(args.steps, 3))
        decoded[0] = 0
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = one_pass()
        torch.cuda.synchronize()
        own = time.perf_counter() - t0   # this rank's own passes, before it waits for the others
        barrier()
        elapsed = shard.max_over_ranks(time.perf_counter() - t0, dist, device=REDUCE_DEVICE)
        per_rank = [round(size * steps / t / 1e9, 2) for t in shard.gather_over_ranks(own, dist, device=REDUCE_DEVICE)]
        for _, _, (stripe, offsets, num_bytes) in res:
            w = want[stripe % len(want)]
            if num_bytes != int(w[-1]) or not np.array_equal(offsets, w):
                raise SystemExit("bench.py: offsets of stripe %d differ from the lengths written" % stripe)
        for c in contexts:
            c.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {
        "metric": "end-to-end GB/s of ORC file (read + H2D + decode + D2H of offsets)",
        "value": round(size * steps * world / elapsed / 1e9, 2), "unit": "GB/s", "n_gpus": world,
        "steps": steps, "warmup": min(args.warmup, 1), "ms_per_step": round(elapsed / steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "decoded_GBps": round(decoded[0] * world / elapsed / 1e9, 2),
        "per_rank_file_GBps": per_rank,
        "config": {"workload": "strings_file: uncompressed ORC file of %d stripes x %d strings, lengths "
                               "U[8,56] as RLE v2 Direct width 6, ASCII DATA (seed 46), %d bytes per GPU; "
                               "%d contexts per GPU%s" % (n_stripes, FILE_STRINGS_PER_STRIPE, size, FILE_CONTEXTS,
                                                          ", each reading its next stripe ahead" if FILE_PREFETCH else ""),
                   "file_bytes_per_gpu": size, "stripes": n_stripes,
                   "parallelism": "stripe-sharded x%d" % world},
    }


def bench_real_file(args, rank, world, local_rank, dist, torch, np):
    """One step = the DATA streams of an int64 column of a file written by the Apache ORC C++ writer
    (pyarrow), every stripe of it, decoded where they lie in HBM with the entry points the file's own
    ROW_INDEX gives (orcd_rlev2_decode_groups: a position per row group of 10 000 values, which is
    19 or 20 runs of 512).  real_file_direct: values below 2^16, written with
    compression_strategy='compression' -> 17-bit Direct runs; real_file_delta: sorted timestamps ->
    Delta runs.  What is timed is the decode proper, as for the synthetic columns; `path` says which
    way the calls went (1 strided, 2 planned, 0 general)."""
    import shutil
    import tempfile
    import pyarrow as pa
    import pyarrow.orc as orc
    from orc_format_amd import Context, read, shard

    rows = args.rows or 100_000_000
    rng = np.random.default_rng((7 if args.workload == "real_file_direct" else 8) + 1000 * rank)
    if args.workload == "real_file_direct":
        values = rng.integers(0, 2**16, size=rows)
        what = "values U[0, 2^16), compression_strategy='compression': Direct runs of 17 bits"
    else:
        values = 1_600_000_000_000 + np.cumsum(rng.integers(1, 1000, size=rows))
        what = "sorted timestamps, steps U[1, 1000): Delta runs"
    tmp = tempfile.mkdtemp(prefix="orcd_bench_")
    try:
        path = os.path.join(tmp, "%s_%d.orc" % (args.workload, rank))
        orc.write_table(pa.table({"v": pa.array(values)}), path, compression="uncompressed",
                        compression_strategy="compression", stripe_size=1 << 30)
        md = read.read_metadata(path)
        stripes = []
        first = 0
        stream_bytes = 0
        for k, st in enumerate(md.stripes):
            footer = read.read_stripe_footer(md, k)
            col = read.read_stripe_column(md, k, footer, 1)
            data = col.get_stream(1)
            ent = col.entry_points(1)
            d_stream = torch.from_numpy(np.array(data)).cuda()
            d_ent = torch.from_numpy(ent.astype(np.int64)).cuda()
            stripes.append((d_stream, d_ent, int(st.number_of_rows), first, col.values_per_group(1)))
            first += int(st.number_of_rows)
            stream_bytes += int(d_stream.numel())
        assert first == rows
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    expect = torch.from_numpy(values).cuda()
    out = torch.empty(rows, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx = Context(local_rank)
    paths = set()

    def step():
        for d_stream, d_ent, n, at, vpg in stripes:
            ctx.rlev2_device(d_stream.data_ptr(), d_stream.numel(), n, True, out.data_ptr() + 8 * at,
                             d_ent.data_ptr(), d_ent.numel(), values_per_group=vpg)
            paths.add(ctx.last_path())

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    out.zero_()
    paths.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, dist, device=REDUCE_DEVICE)
    if not torch.equal(out, expect):
        raise SystemExit("bench.py: %s: decoded column differs from what was written" % args.workload)
    value = rows * 8 * args.steps * world / elapsed / 1e9
    return {
        "metric": "device-resident decoded GB/s (RLEv2 int64 + strings) at 1/2/4/8 GPUs",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "decoded_frac_of_hbm_peak": round(value / world / HBM_PEAK_GBPS, 4),
        "path": sorted(paths),
        "config": {"workload": "%s: %d rows per GPU per step in %d stripe(s) of a pyarrow-written ORC file; %s; "
                               "entry points and values per row group (%s) from the file's ROW_INDEX"
                               % (args.workload, rows, len(stripes), what, sorted({s[4] for s in stripes})),
                   "rows_per_gpu": rows, "stream_bytes_per_gpu": stream_bytes, "stripes": len(stripes),
                   "parallelism": "stripe-sharded x%d" % world},
    }


def bench_mixed_table(args, rank, world, local_rank, dist, torch, np):
    """One step = one pass over this rank's mixed-schema ORC file (the Apache ORC C++ writer through
    pyarrow): low-cardinality int keys in runs of geometric length (ShortRepeat / short Direct runs),
    a nullable int64 column with 10 % nulls (PRESENT + DATA), a float64 and a boolean column.  From
    the file's bytes (page cache) to row-aligned columns on the GPU: stripe reads, host-to-device
    copies, PRESENT -> validity, DATA -> values, null scatter (orcd_decode_stripe with
    ORCD_OUT_DEVICE | ORCD_ROW_ALIGNED), FILE_CONTEXTS contexts with a host thread each and the next
    stripe read ahead.  Like strings_file it measures a whole path -- the feed bounds it -- and
    carries no roofline of its own; `per_stripe_ms` is one context alone, nothing read ahead: the
    host's time until the stripe's bytes are on their way to the device, and from then to the
    decoded columns."""
    import shutil
    import tempfile
    import pyarrow as pa
    import pyarrow.orc as orc
    from orc_format_amd import Context, read, shard

    rows = args.rows or 100_000_000
    rng = np.random.default_rng(44 + 1000 * rank)
    # keys: values 0..15, each repeated a geometric number of times (mean 4)
    n_runs = rows // 3
    reps = rng.geometric(0.25, size=n_runs)
    keys = np.repeat(rng.integers(0, 16, size=n_runs).astype(np.int32), reps)[:rows]
    if keys.size < rows:
        keys = np.concatenate([keys, np.zeros(rows - keys.size, dtype=np.int32)])
    nulls = rng.random(rows) < 0.1
    table = pa.table({
        "key": pa.array(keys),
        "val": pa.array(rng.integers(-2**40, 2**40, size=rows), mask=nulls),
        "x": pa.array(rng.standard_normal(rows)),
        "flag": pa.array(rng.random(rows) < 0.5),
    })
    del keys, reps
    tmp = tempfile.mkdtemp(prefix="orcd_bench_")
    try:
        path = os.path.join(tmp, "mixed_%d.orc" % rank)
        orc.write_table(table, path, compression="uncompressed", stripe_size=64 << 20)
        size = os.path.getsize(path)
        md = read.read_metadata(path)
        n_stripes = len(md.stripes)
        contexts = [Context(local_rank) for _ in range(FILE_CONTEXTS)]
        # parity of the first stripe with what was written, row by row
        ft0 = read.read_stripe_footer(md, 0)
        first = contexts[0].decode_stripe(md, 0, ft0, [1, 2, 3, 4], row_aligned=True)
        n0 = int(md.stripes[0].number_of_rows)
        for k, name in enumerate(["key", "val", "x", "flag"]):
            want = table.column(name).slice(0, n0)
            valid = np.array(want.is_valid())
            if not np.array_equal(first[k]["validity"].astype(bool), valid):
                raise SystemExit("bench.py: validity of column %s differs from what was written" % name)
            w = (want.fill_null(0) if name == "val" else want).to_numpy(zero_copy_only=False)
            got = first[k]["values"]
            same = got.view(np.uint64)[valid].tolist() == w.view(np.uint64)[valid].tolist() if name == "x" else \
                np.array_equal(got[valid].astype(np.int64), w[valid].astype(np.int64))
            if not same:
                raise SystemExit("bench.py: column %s of the first stripe differs from what was written" % name)
        del first
        # one context alone, nothing read ahead: where a stripe's time goes
        feed, dec = [], []
        for k in range(min(n_stripes, 8)):
            cols = contexts[0].decode_stripe(md, k, read.read_stripe_footer(md, k), [1, 2, 3, 4], row_aligned=True,
                                             device=True)
            f_ms, d_ms = contexts[0].last_stripe_ms()
            for c in cols:
                c.free()
            if k:   # (the first call grows the context's buffers)
                feed.append(f_ms)
                dec.append(d_ms)
        decoded = [0]

        def on_stripe(context, stripe, cols):
            for c in cols:
                decoded[0] += c.num_slots * (c.value_size + 1)   # values and validity, a slot per row
                c.free()
            return None

        def one_pass():
            return shard.decode_file_shard(contexts[0], md, 0, 1, columns=[1, 2, 3, 4], row_aligned=True,
                                           extra_contexts=contexts[1:], device=True, on_stripe=on_stripe,
                                           prefetch=FILE_PREFETCH)

        def barrier():
            for c in contexts:
                c.synchronize()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()

        for _ in range(min(args.warmup, 1)):
            one_pass()
        steps = max(1, min(args.steps, 3))
        decoded[0] = 0
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_pass()
        barrier()
        elapsed = shard.max_over_ranks(time.perf_counter() - t0, dist, device=REDUCE_DEVICE)
        result = {
            "metric": "end-to-end decoded GB/s of a mixed-schema ORC file (read + H2D + decode + null scatter)",
            "value": round(decoded[0] * world / elapsed / 1e9, 2), "unit": "GB/s", "n_gpus": world,
            "steps": steps, "warmup": min(args.warmup, 1), "ms_per_step": round(elapsed / steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64",
            "data": "synthetic",
            "decoded_GBps": round(decoded[0] * world / elapsed / 1e9, 2),
            "file_GBps": round(size * steps * world / elapsed / 1e9, 2),
            "per_stripe_ms": {"feed": round(float(np.mean(feed)), 3) if feed else None,
                              "decode": round(float(np.mean(dec)), 3) if dec else None,
                              "stripes_sampled": len(dec),
                              "note": "one context, nothing read ahead, host clock: until the stripe's bytes are "
                                      "read and their copies queued / from then until every column is decoded"},
            "config": {"workload": "mixed_table: pyarrow-written ORC file, uncompressed, %d rows x {int32 keys 0..15 in "
                                   "runs of geometric length, int64 with 10%% nulls, float64, boolean} (seed 44), %d bytes "
                                   "and %d stripes per GPU; row-aligned columns left on the device; %d contexts per GPU%s"
                                   % (rows, size, n_stripes, FILE_CONTEXTS,
                                      ", each reading its next stripe ahead" if FILE_PREFETCH else ""),
                       "file_bytes_per_gpu": size, "stripes": n_stripes, "rows_per_gpu": rows,
                       "parallelism": "stripe-sharded x%d" % world},
        }
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = mixed_table_cpu_baseline(args, md, read, np)
        for c in contexts:
            c.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return result


def mixed_table_cpu_baseline(args, md, read, np):
    """The CPU oracle over the same streams, stripe after stripe for about --cpu-seconds: PRESENT ->
    validity, DATA -> dense values per column (the pair deserialize_*_array returns,
    tests/it/deserialize.rs:27-104; no scatter to rows)."""
    import oracle_ctypes
    oracle = oracle_ctypes.load()
    t_all, decoded, rows_done, k = 0.0, 0, 0, 0
    kinds = {1: "int", 2: "int", 3: "f64", 4: "bool"}
    while k < len(md.stripes) and t_all < args.cpu_seconds:
        footer = read.read_stripe_footer(md, k)
        rows = int(md.stripes[k].number_of_rows)
        for ci, kind in kinds.items():
            col = read.read_stripe_column(md, k, footer, ci)
            present = np.array(col.get_stream(0)) if col.has_stream(0) else None
            data = np.array(col.get_stream(1))
            t0 = time.perf_counter()
            n = rows
            if present is not None:
                rc, bits, n = oracle.boolean(present, rows)
                assert rc == 0
            if kind == "int":
                rc, vals, _ = oracle.rlev2(data, n, True)
            elif kind == "f64":
                rc, vals = oracle.floats(data, n, np.float64)
            else:
                rc, vals, _ = oracle.boolean(data, n)
            assert rc == 0
            t_all += time.perf_counter() - t0
            decoded += n * (1 if kind == "bool" else 8) + rows
        rows_done += rows
        k += 1
    return {"value": round(decoded / t_all / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "the first %d stripes (%d rows) of the same file, every stream through the oracle, single "
                      "thread, %.1f s (streams already in memory; validity + dense values, no scatter)"
                      % (k, rows_done, t_all)}


def bench_zlib_file(args, rank, world, local_rank, dist, torch, np):
    """One step = one pass over this rank's zlib-compressed ORC file (pyarrow's writer, 64 KiB
    compression blocks; four columns: slowly varying and random 64-bit integers, doubles, short
    strings): stripe reads, the chunks inflated on the GPU (by the host's zlib threads with
    ORCD_GPU_INFLATE=0), the streams decoded, the columns left on the device.  Like strings_file it
    measures a whole path and carries no roofline; not part of the default run (writing the file
    takes the host half a minute)."""
    import shutil
    import tempfile
    import pyarrow as pa
    import pyarrow.orc as orc
    from orc_format_amd import Context, read, shard

    rows = args.rows or 20_000_000
    rng = np.random.default_rng(5 + 1000 * rank)
    table = pa.table({
        "walk": pa.array(np.cumsum(rng.integers(-50, 51, size=rows))),
        "wide": pa.array(rng.integers(0, 2**40, size=rows)),
        "real": pa.array(np.round(rng.standard_normal(rows), 3)),
        "word": pa.array(np.char.add("k", rng.integers(0, 10**6, size=rows).astype(str))),
    })
    tmp = tempfile.mkdtemp(prefix="orcd_bench_")
    try:
        path = os.path.join(tmp, "zlib_%d.orc" % rank)
        orc.write_table(table, path, compression="zlib", compression_block_size=65536, stripe_size=64 << 20,
                        dictionary_key_size_threshold=0.0)
        size = os.path.getsize(path)
        md = read.read_metadata(path)
        n_stripes = len(md.stripes)
        contexts = [Context(local_rank) for _ in range(FILE_CONTEXTS)]
        # the first stripe against what was written
        first = contexts[0].decode_stripe(md, 0, read.read_stripe_footer(md, 0), [1, 2, 3])
        n0 = first[0]["values"].size
        for k, name in enumerate(["walk", "wide", "real"]):
            if not np.array_equal(first[k]["values"], table.column(name).to_numpy()[:n0]):
                raise SystemExit("bench.py: column %s of the first stripe differs from what was written" % name)
        decoded = [0]

        def on_stripe(context, stripe, cols):
            for c in cols:
                decoded[0] += c.num_slots * 8 + c.num_bytes
                c.free()
            return None

        def one_pass():
            return shard.decode_file_shard(contexts[0], md, 0, 1, columns=[1, 2, 3, 4], row_aligned=False,
                                           extra_contexts=contexts[1:], device=True, on_stripe=on_stripe,
                                           prefetch=FILE_PREFETCH)

        def barrier():
            for c in contexts:
                c.synchronize()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()

        for _ in range(min(args.warmup, 1)):
            one_pass()
        steps = max(1, min(args.steps, 5))
        decoded[0] = 0
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_pass()
        barrier()
        elapsed = shard.max_over_ranks(time.perf_counter() - t0, dist, device=REDUCE_DEVICE)
        for c in contexts:
            c.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    where = "the host's zlib threads" if os.environ.get("ORCD_GPU_INFLATE", "")[:1] == "0" else "the GPU"
    return {
        "metric": "end-to-end GB/s of zlib ORC file (read + H2D + inflate + decode)",
        "value": round(size * steps * world / elapsed / 1e9, 2), "unit": "GB/s", "n_gpus": world,
        "steps": steps, "warmup": min(args.warmup, 1), "ms_per_step": round(elapsed / steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "decoded_GBps": round(decoded[0] * world / elapsed / 1e9, 2),
        "config": {"workload": "zlib_file: pyarrow-written ORC file, zlib, 64 KiB compression blocks, %d rows x 4 "
                               "columns (seed 5), %d bytes and %d stripes per GPU; chunks inflated by %s; %d contexts "
                               "per GPU%s" % (rows, size, n_stripes, where, FILE_CONTEXTS,
                                              ", each reading its next stripe ahead" if FILE_PREFETCH else ""),
                   "file_bytes_per_gpu": size, "stripes": n_stripes,
                   "parallelism": "stripe-sharded x%d" % world},
    }


def bench_boolean(args, rank, world, local_rank, dist, desc, torch, np):
    """One step = one boolean stream (PRESENT or boolean DATA): byte-RLE -> Vec<bool>-style bytes
    and the number of set bits (orcd_boolean_decode)."""
    from bench_util import synth
    from orc_format_amd import Context, shard

    n_blocks = max(1, args.rows // 8 // 129)
    stream, entries, packed = synth.byte_rle_mix(n_blocks, seed=45 + 1000 * rank, device="cuda")
    num_bits = packed.numel() * 8 - 3          # not a multiple of 8: the last byte is partial
    want = synth.unpack_bools(packed, num_bits)
    want_ones = int(want.sum(dtype=torch.int64).item())
    out = torch.empty(num_bits, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx = Context(local_rank)
    ones = [0]

    def step():
        ones[0] = ctx.boolean_device(stream.data_ptr(), stream.numel(), num_bits, out.data_ptr(), 0,
                                     entries.data_ptr(), entries.numel(), want_count=True)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    out.zero_()   # the check below sees what the timed steps wrote, not the warm-up
    ones[0] = -1
    barrier()
    phase_ms = {1: 0.0, 2: 0.0, 3: 0.0}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for p in phase_ms:
            phase_ms[p] += ctx.phase_ms(p)
    barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, dist, device=REDUCE_DEVICE)
    if ones[0] != want_ones or not torch.equal(out, want):
        raise SystemExit("bench.py: decoded booleans differ from the generator's")

    decoded_bytes = num_bits
    value = decoded_bytes * args.steps * world / elapsed / 1e9
    k_ms = phase_ms[3] / args.steps
    algo_bytes = num_bits + int(stream.numel())
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    result = {
        "metric": "device-resident decoded GB/s (RLEv2 int64 + strings) at 1/2/4/8 GPUs",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "boolean_mix: %d values per GPU per step; %s" % (num_bits, desc),
                   "values_per_gpu": num_bits, "stream_bytes_per_gpu": int(stream.numel()),
                   "entry_points_per_gpu": int(entries.numel()),
                   "parallelism": "stripe-sharded x%d" % world},
        "roofline": {"bound": "hbm", "kernel": "bool_expand_kernel", "achieved": round(achieved, 1),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                     "traffic": None, "algorithmic_bytes_per_launch": algo_bytes,
                     "avg_launch_ms": round(k_ms, 4)},
        "phases_ms": {"header_count_scan_readback": round(phase_ms[1] / args.steps, 4),
                      "run_table_emit": round(phase_ms[2] / args.steps, 4),
                      "expand_to_booleans": round(k_ms, 4)},
        "decoded_frac_of_hbm_peak": round(value / world / HBM_PEAK_GBPS, 4),
    }
    if args.rows == DEFAULT_ROWS["boolean_mix"]:
        attach_traffic(result, args.workload, "bool_expand_kernel")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_ctypes
        oracle = oracle_ctypes.load()

        def run(n_entries):
            nbytes = int(entries[n_entries].item()) if n_entries < entries.numel() else int(stream.numel())
            host = stream[:nbytes].cpu().numpy()
            # decoded bytes of the first n_entries * 10 blocks
            n_bits = min(num_bits, 8 * 128 * 10 * n_entries)   # a lower bound of what the prefix holds
            t0 = time.perf_counter()
            rc, vals, cnt = oracle.boolean(host, n_bits)
            dt = time.perf_counter() - t0
            assert rc == 0 and (vals == want[:n_bits].cpu().numpy().astype(bool)).all()
            return n_bits, dt
        n_cal, dt_cal = run(min(entries.numel(), 2000))
        scale = args.cpu_seconds / max(dt_cal, 1e-6)
        n_s, dt = run(int(min(entries.numel(), max(2000, 2000 * scale))))
        result["cpu_baseline"] = {
            "value": round(n_s / dt / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "first %d values of the same stream, single thread, %.1f s" % (n_s, dt)}
    return result


def end_to_end(args, ctx, stream, entries, expect, rows, signed, torch):
    """Rate of the same decode when the stream starts in (pinned) host memory and the decoded
    column ends there: H2D copy + decode + D2H copy on one HIP stream, no overlap.  Not the
    headline value -- PCIe, not HBM, bounds it."""
    n = min(rows, 256 * 1024 * 1024)
    n_runs = n // 512
    n = n_runs * 512
    sbytes = int(stream.numel() * (n / rows))
    h_stream = torch.empty(sbytes, dtype=torch.uint8).pin_memory()
    h_stream.copy_(stream[:sbytes])
    h_out = torch.empty(n, dtype=torch.int64).pin_memory()
    d_stream = torch.empty(sbytes, dtype=torch.uint8, device="cuda")
    d_out = torch.empty(n, dtype=torch.int64, device="cuda")
    n_entries = (n_runs + 19) // 20
    side = torch.cuda.Stream()  # copies and kernels share this (non-default) stream
    ctx.set_stream(side.cuda_stream)
    times = []
    with torch.cuda.stream(side):
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d_stream.copy_(h_stream, non_blocking=True)
            ctx.rlev2_device(d_stream.data_ptr(), sbytes, n, signed, d_out.data_ptr(),
                             entries.data_ptr(), n_entries)
            h_out.copy_(d_out, non_blocking=True)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    ctx.set_stream(None)
    assert torch.equal(h_out, expect[:n].cpu())
    best = min(times)
    # the same with ordinary (pageable) buffers on both sides, through the one host-pointer call a
    # binding makes: the library moves them in chunks through pinned staging on several threads
    import numpy as np
    p_stream = h_stream.numpy().copy()
    p_entries = entries[:n_entries].cpu().numpy().astype(np.uint64)
    p_times = []
    for _ in range(3):
        t0 = time.perf_counter()
        p_out = ctx.rlev2(p_stream, n, signed, p_entries)
        p_times.append(time.perf_counter() - t0)
    assert (p_out == h_out.numpy()).all()
    return {"decoded_GBps": round(n * 8 / best / 1e9, 2), "rows": n, "seconds": round(best, 4),
            "host_to_device_bytes": sbytes, "device_to_host_bytes": n * 8,
            "note": "pinned host buffers, H2D + decode + D2H serialized on one stream",
            "pageable_buffers": {"decoded_GBps": round(n * 8 / min(p_times) / 1e9, 2),
                                 "seconds": round(min(p_times), 4),
                                 "note": "orcd_rlev2_decode on pageable host memory, output array "
                                         "allocated (and first touched) inside the timed call"}}


def cpu_baseline(args, stream, rows, stream_bytes, signed, expect, np, cuts=None):
    """The CPU oracle (a C restatement of the reference's value-at-a-time decoder, not the Rust
    crate itself) timed on the host cores over a bounded prefix of the same stream.
    `cuts` = (values before, bytes before) at run boundaries where the stream may be cut; None for
    the columns whose runs all hold 512 values in the same number of bytes."""
    import oracle_ctypes
    oracle = oracle_ctypes.load()
    if cuts is None:
        run_bytes = stream_bytes // (rows // 512)

        def cut(n_vals):   # largest boundary with at most n_vals values before it
            k = max(1, min(n_vals, rows) // 512)
            return k * 512, k * run_bytes
    else:
        cut_vals, cut_bytes = cuts

        def cut(n_vals):
            i = max(1, int(np.searchsorted(cut_vals, n_vals, side="right")) - 1)
            if i >= len(cut_vals):
                return rows, stream_bytes
            return int(cut_vals[i]), int(cut_bytes[i])

    def run(n_target):   # calibrate, then size the sample for about --cpu-seconds of one thread
        n_vals, nbytes = cut(n_target)
        host = stream[:nbytes].cpu().numpy()
        t0 = time.perf_counter()
        rc, vals, _ = oracle.rlev2(host, n_vals, signed)
        dt = time.perf_counter() - t0
        assert rc == 0
        assert (vals == expect[:n_vals].cpu().numpy()).all(), "oracle disagrees with the generator"
        return n_vals, nbytes, dt
    n_cal, _, dt_cal = run(min(rows, 4 << 20))
    n, nbytes, dt = run(int(min(rows, max(n_cal, n_cal * args.cpu_seconds / max(dt_cal, 1e-6)))))
    out = {"value": round(n * 8 / dt / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
           "sample": "first %d values (%.1f MB of stream) of the same column, single thread, "
                     "%.1f s" % (n, nbytes / 1e6, dt)}
    # The same oracle, stripe-parallel: the sample is cut into one stripe per thread at run
    # boundaries and the threads decode side by side (ctypes releases the GIL during the C call).
    # Reported beside the single-thread figure, which is what the reference -- a single-threaded
    # iterator -- does.
    import threading
    threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1), 64))
    bounds = [(0, 0)] + [cut((i + 1) * n // threads) for i in range(threads)]
    bounds = [b for i, b in enumerate(bounds) if i == 0 or b[0] > bounds[i - 1][0]]
    host = stream[:bounds[-1][1]].cpu().numpy()
    parts = [(host[b0[1]:b1[1]], b1[0] - b0[0]) for b0, b1 in zip(bounds[:-1], bounds[1:])]
    want_last = [int(expect[b[0] - 1].item()) for b in bounds[1:]]
    ok = [False] * len(parts)

    def work(i):
        rc, vals, _ = oracle.rlev2(parts[i][0], parts[i][1], signed)
        ok[i] = rc == 0 and int(vals[-1]) == want_last[i]
    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(parts))]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    dtp = time.perf_counter() - t0
    assert all(ok), "parallel oracle run disagrees with the generator"
    out["stripe_parallel"] = {"value": round(bounds[-1][0] * 8 / dtp / 1e9, 4), "unit": "GB/s",
                              "cores": len(parts),
                              "sample": "%d stripes of about %d values, one thread each, %.1f s"
                                        % (len(parts), bounds[-1][0] // len(parts), dtp)}
    return out


if __name__ == "__main__":
    main()

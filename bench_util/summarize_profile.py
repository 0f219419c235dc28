"""Condenses rocprofv3 output directories into the summaries kept under profiles/.

    python bench_util/summarize_profile.py <trace_dir> <fetch_dir> <write_dir> <out_prefix> <workload> [<unprofiled.json>]

trace_dir: `rocprofv3 --kernel-trace --stats --output-format csv` of bench.py;
fetch_dir / write_dir: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of the same command.
unprofiled.json (optional): the JSON line of the same bench command run without the profiler in
the same lease; its event timings are kept as "unprofiled_same_lease".
Writes <out_prefix>_kernel_stats.csv (the decoder's kernels only) and <out_prefix>_traffic.json.

Counter units and the gfx950 correction: rocprofv3 reports FETCH_SIZE and WRITE_SIZE in units of
1024 bytes.  FETCH_SIZE tallies the 128-byte requests of wide coalesced streaming reads (16 bytes
per lane) at 64 bytes, so for a kernel whose reads are of that kind the read side is doubled; that
is calibrated only for such reads.  WIDE_READS below names those kernels (their stream bytes come
in through 16-byte-per-lane loads of consecutive chunks); for every other kernel -- the header
scans with their per-lane walks, the gathers, the small table passes -- the raw figure is kept as
the estimate and the doubled one is given beside it as the upper bound
(`hbm_read_bytes_per_launch_upper`): the truth lies between the two.
"""
import collections
import csv
import glob
import json
import os
import sys


# kernels whose dominant reads are 16 bytes per lane, consecutive lanes consecutive chunks
WIDE_READS = ("rlev2_values_kernel", "rlev2_strided_values_kernel", "rlev2_selfplan_values_kernel",
              "rlev2_planned_values_kernel", "string_bytes_kernel",
              "byte_rle_exits_kernel", "byte_rle_tile_kernel")


def _one(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        raise SystemExit("no file matches " + pattern)
    return files[0]


def main():
    trace_dir, fetch_dir, write_dir, prefix, workload = sys.argv[1:6]
    stats = _one(os.path.join(trace_dir, "**", "*_kernel_stats.csv"))
    rows = [r for r in csv.DictReader(open(stats)) if "orcd::" in r["Name"]]
    with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    traffic = {}
    for kind, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        cc = _one(os.path.join(d, "**", "*_counter_collection.csv"))
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(cc)):
            if "orcd::" in r["Kernel_Name"] and r["Counter_Name"] == kind:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            traffic.setdefault(k, {})[kind + "_kb_avg"] = sum(v) / len(v)
            traffic[k]["dispatches_" + kind] = len(v)
    for k, t in traffic.items():
        raw = t.get("FETCH_SIZE_kb_avg", 0.0) * 1024
        wide = any(w in k for w in WIDE_READS)
        fetch = raw * 2 if wide else raw   # gfx950: counts half of wide reads
        write = t.get("WRITE_SIZE_kb_avg", 0.0) * 1024
        t["read_correction"] = "x2 (wide coalesced reads)" if wide else "none (uncalibrated pattern; upper bound = x2)"
        if not wide:
            t["hbm_read_bytes_per_launch_upper"] = raw * 2
        t["hbm_read_bytes_per_launch"] = fetch
        t["hbm_write_bytes_per_launch"] = write
        t["hbm_bytes_per_launch"] = fetch + write
    durations = {r["Name"].split("(")[0]: float(r["AverageNs"]) for r in rows}
    out = {"workload": workload, "kernels": traffic, "avg_duration_ns": durations}
    if len(sys.argv) > 6 and os.path.exists(sys.argv[6]):
        with open(sys.argv[6]) as f:
            line = [ln for ln in f.read().splitlines() if ln.startswith("{")][-1]
        b = json.loads(line)
        out["unprofiled_same_lease"] = {
            "kernel": b["roofline"]["kernel"], "avg_launch_ms": b["roofline"]["avg_launch_ms"],
            "frac": b["roofline"]["frac"], "ms_per_step": b["ms_per_step"], "steps": b["steps"],
            "note": "bench.py without rocprofv3, HIP events, same box and lease as the trace"}
    with open(prefix + "_traffic.json", "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by tools/prof.sh into one per-kernel table (mean per dispatch)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
mode = sys.argv[3] if len(sys.argv) > 3 else "kernels"

# queue columns k_trace streams per ray besides the traversal blob (kernels_trace.h refill): valid flag 4 B + ray 24 B, + 12 B
# of target for shadow rays (about half of the rays); coalesced, 128-B requests that miss L1 and L2 alike
STREAM_READ_REQ_PER_RAY = (4 + 24 + 6) / 128.0


def ubench_calibration():
    """The micro-benchmark's own cache counters per lane-step, next to the rates it was asked to reproduce."""
    want = open(os.path.join(out, "ubench_args.txt")).read().split()
    line = open(os.path.join(out, "ubench_plain.txt")).read().strip()
    m1, m2, lanes, wps = float(want[0]), float(want[1]), int(want[2]), int(want[3])
    per_launch = 256.0 * wps * 4 * lanes * 2000.0
    a, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(out, "ubench", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_gather_tiers" in row["Kernel_Name"]:
                a[row["Counter_Name"]] += float(row["Counter_Value"])
                n[row["Counter_Name"]] += 1
    print("== gather-ceiling micro-benchmark (tools/ubench_gather.hip, k_gather_tiers) under the cache counters")
    print(line)
    print(f"asked for: {m1:.4f} TCP->TCC read requests and {m2:.4f} TCC->fabric read requests per lane-step (the trace kernel's, "
          f"{tag}_pmc_traffic.json k_trace_all), {lanes} lanes, {wps} waves/SIMD; {per_launch:.0f} lane-steps per launch")
    for c in sorted(a):
        print(f"   {c:36s} {a[c] / max(n[c], 1):16.1f} per launch (n={n[c]})  = {a[c] / max(n[c], 1) / per_launch:8.4f} per lane-step")
    g = lambda c: a[c] / max(n[c], 1) if c in a else None
    if g("TCP_TCC_READ_REQ_sum") is not None:
        print(f"   -> measured TCP->TCC read requests per lane-step {g('TCP_TCC_READ_REQ_sum') / per_launch:.4f}  (asked {m1:.4f})")
    if g("TCC_EA0_RDREQ_sum") is not None:
        print(f"   -> measured TCC->fabric read requests per lane-step {g('TCC_EA0_RDREQ_sum') / per_launch:.4f}  (asked {m2:.4f})")
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
        print(f"   -> TCP miss share per access {g('TCP_TCC_READ_REQ_sum') / g('TCP_TOTAL_CACHE_ACCESSES_sum') * 100:.1f} %")


if mode == "ubench":
    ubench_calibration()
    sys.exit(0)


def short(name):
    """k_trace<COUNT, GEN> -> k_trace / k_trace<primary> (+ <count>); k_shade<PRIMARY> -> k_shade / k_shade<primary>."""
    base = name.split("(")[0].replace("void ", "").strip()
    args = [a.strip() for a in base[base.index("<") + 1: base.rindex(">")].split(",")] if "<" in base and ">" in base else []
    if "k_trace" in base:
        return "k_trace" + ("<primary>" if len(args) > 1 and args[1] == "true" else "") + ("<count>" if args and args[0] == "true" else "")
    if "k_shade_alt" in base:
        return "k_shade_alt"
    if "k_shade" in base:
        return "k_shade" + ("<primary>" if args and args[0] == "true" else "")
    for k in ("k_raygen", "k_resolve", "k_pack", "k_unpack"):
        if k in base:
            return k
    return base[:40]


print("== kernel-trace stats (rocprofv3 --kernel-trace --stats)")
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print(f"{short(row['Name']):24s} calls {row['Calls']:>6s}  avg {float(row['AverageNs'])/1e3:9.1f} us  total {float(row['TotalDurationNs'])/1e6:9.2f} ms  {row['Percentage']}%")

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
print("\n== PMC, mean per dispatch")
for k in sorted(acc):
    print(f"[{k}]")
    m = {c: acc[k][c] / max(cnt[k][c], 1) for c in acc[k]}
    for c in sorted(m):
        print(f"   {c:36s} {m[c]:16.1f}   (n={cnt[k][c]})")
    g = m.get
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
        print(f"   -> VALU lane utilisation   {g('SQ_THREAD_CYCLES_VALU') / (g('SQ_ACTIVE_INST_VALU') * 64) * 100:6.1f} %  (THREAD_CYCLES_VALU / (ACTIVE_INST_VALU*64))")
    if g("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM"):
            if g(c) is not None:
                print(f"   -> {c:24s} / SQ_WAVE_CYCLES = {g(c) / g('SQ_WAVE_CYCLES') * 100:6.1f} %")
    if g("SQ_BUSY_CYCLES") and g("SQ_WAVE_CYCLES"):
        print(f"   -> mean waves in flight per SQ-busy cycle = {g('SQ_WAVE_CYCLES') / g('SQ_BUSY_CYCLES'):6.2f}")
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
        print(f"   -> L2 hit rate {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')) * 100:6.1f} %")
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
        print(f"   -> L1 (TCP) miss share {g('TCP_TCC_READ_REQ_sum') / g('TCP_TOTAL_CACHE_ACCESSES_sum') * 100:6.1f} %  (TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES)")
    if g("FETCH_SIZE") is not None or g("WRITE_SIZE") is not None:
        f_, w_ = g("FETCH_SIZE") or 0.0, g("WRITE_SIZE") or 0.0
        print(f"   -> HBM-side traffic per dispatch: 2*FETCH_SIZE {2 * f_ * 1024 / 1e6:9.3f} MB + WRITE_SIZE {w_ * 1024 / 1e6:9.3f} MB  (gfx950: FETCH_SIZE counts 64 B per 128-B request)")

# ---- machine-readable HBM-side traffic per launch for bench.py's roofline.traffic (same workload as ptbench's default)
plain = {}
try:
    plain = json.loads(open(os.path.join(out, "plain.json")).read().strip().splitlines()[-1])
except Exception:
    pass
kern = {}
for k in acc:
    f_ = acc[k].get("FETCH_SIZE", 0.0) / max(cnt[k].get("FETCH_SIZE", 1), 1)
    w_ = acc[k].get("WRITE_SIZE", 0.0) / max(cnt[k].get("WRITE_SIZE", 1), 1)
    if "FETCH_SIZE" in acc[k] or "WRITE_SIZE" in acc[k]:
        kern[k] = {"fetch_size_kb": f_, "write_size_kb": w_, "hbm_bytes_per_launch": (2.0 * f_ + w_) * 1024.0,
                   "launches_profiled": cnt[k].get("FETCH_SIZE", 0)}
for k in kern:
    tot, req = acc[k].get("TCP_TOTAL_CACHE_ACCESSES_sum"), acc[k].get("TCP_TCC_READ_REQ_sum")
    if tot:
        kern[k]["l1_miss_share"] = (req or 0.0) / tot
    for c, name in (("TCP_TCC_READ_REQ_sum", "tcp_tcc_read_req"), ("TCC_EA0_RDREQ_sum", "tcc_ea_rdreq"), ("TCP_TOTAL_CACHE_ACCESSES_sum", "tcp_total_cache_accesses"),
                    ("SQ_INSTS_VALU", "sq_insts_valu"), ("SQ_INSTS_SALU", "sq_insts_salu"), ("SQ_THREAD_CYCLES_VALU", "sq_thread_cycles_valu"),
                    ("SQ_ACTIVE_INST_VALU", "sq_active_inst_valu")):
        if c in acc[k]:
            kern[k][name] = acc[k][c] / max(cnt[k][c], 1)
    hit, miss = acc[k].get("TCC_HIT_sum"), acc[k].get("TCC_MISS_sum")
    if hit is not None and miss is not None and hit + miss > 0:
        kern[k]["l2_hit_rate"] = hit / (hit + miss)
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Name"])
        if k in kern:
            kern[k]["avg_launch_ms_kernel_trace"] = float(row["AverageNs"]) / 1e6
# "k_trace_all": every trace launch of a batch (primary + bounce stages), launch-weighted -- the population bench.py's
# HIP-event average and rays-per-launch are taken over
tr = [k for k in kern if k.startswith("k_trace") and "count" not in k]
if tr:
    n = sum(kern[k]["launches_profiled"] for k in tr) or 1
    allk = {"launches_profiled": n}
    for f in ("fetch_size_kb", "write_size_kb", "hbm_bytes_per_launch", "l1_miss_share", "l2_hit_rate", "avg_launch_ms_kernel_trace",
              "tcp_tcc_read_req", "tcc_ea_rdreq", "tcp_total_cache_accesses", "sq_insts_valu", "sq_insts_salu", "sq_thread_cycles_valu",
              "sq_active_inst_valu"):
        if all(f in kern[k] for k in tr):
            allk[f] = sum(kern[k][f] * kern[k]["launches_profiled"] for k in tr) / n
    if plain.get("trace_launches"):
        allk["rays_per_launch"] = plain["rays"] / plain["trace_launches"]
    # per lane-step (one lane fetching one 64-B record): cache-miss requests of the record fetches alone, i.e. after
    # taking out the streamed queue columns -- what bench.py hands to the gather-ceiling micro-benchmark
    try:
        counted = json.loads(open(os.path.join(out, "counted.json")).read().strip().splitlines()[-1])
    except Exception:
        counted = {}
    if counted.get("wave_iters") and counted.get("trace_launches") and "tcp_tcc_read_req" in allk and "tcc_ea_rdreq" in allk:
        steps = (counted["node_steps"] + counted["tri_tests"]) / counted["trace_launches"]
        top = counted.get("top_steps", 0) / counted["trace_launches"]  # served from the LDS copy of the tree top: never on the vector-memory path
        gsteps = max(steps - top, 1.0)
        rays = counted["rays"] / counted["trace_launches"]
        primary = 1920.0 * 1080.0 * counted["frames"] / counted["trace_launches"]  # fused primary stage: rays generated, nothing streamed in
        stream = STREAM_READ_REQ_PER_RAY * max(rays - primary, 0.0)
        allk["lane_steps_per_launch"] = steps
        allk["global_lane_steps_per_launch"] = gsteps
        allk["lane_occupancy"] = (counted["node_steps"] + counted["tri_tests"]) / (64.0 * counted["wave_iters"])
        allk["record_miss_l1_per_lane_step"] = max(allk["tcp_tcc_read_req"] - stream, 0.0) / gsteps  # per GLOBAL lane-step
        allk["record_miss_l2_per_lane_step"] = max(allk["tcc_ea_rdreq"] - stream, 0.0) / gsteps
        allk["streamed_read_requests_assumed"] = stream
        with open(os.path.join(out, "ubench_args.txt"), "w") as fh:
            fh.write(f"{allk['record_miss_l1_per_lane_step']:.4f} {allk['record_miss_l2_per_lane_step']:.4f} "
                     f"{max(1, round(allk['lane_occupancy'] * 64 * gsteps / steps))} 8\n")
    kern["k_trace_all"] = allk
if False and "k_trace" in kern and plain.get("trace_launches"):
    kern["k_trace"]["rays_per_launch"] = plain["rays"] / plain["trace_launches"]
if kern:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.setdefault("PT355_NO_TORCH", "1")
    import pathtracer_amd  # only for the stamp: the device sources these counters were taken on
    json.dump({"workload": "bunny-1920x1080-b5", "kernel_sha256": pathtracer_amd.kernel_source_sha256(), "kernel_sources": list(pathtracer_amd.KERNEL_SOURCES),
               "source": f"profiles/{tag}_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes "
               "over pathtracer_amd/ptbench; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)",
               "kernels": kern}, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)

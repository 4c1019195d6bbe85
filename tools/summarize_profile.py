#!/usr/bin/env python3
"""Condenses rocprofv3 output (gpurun_out/prof_<tag>/) into the small files committed under profiles/.

    python tools/summarize_profile.py gpurun_out/prof_r01 profiles/r01

Inputs (any subset): <dir>/trace (--kernel-trace --stats), <dir>/pmc_fetch (--pmc FETCH_SIZE),
<dir>/pmc_write (--pmc WRITE_SIZE) and the bench logs next to them.
HBM traffic follows MI355X_MICROARCH.md §HBM: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on
gfx950 FETCH_SIZE counts exactly half of the bytes of a coalesced streaming read (calibrated here on
curvature_valid_kernel, whose read volume is known: 24 B/point), WRITE_SIZE is exact.
"""
import collections
import csv
import glob

import json
import os
import re
import sys


def one_pass(pattern):
    """the ONE output file of a rocprofv3 pass: gpurun merges a call's files into what is already under gpurun_out/, so a
    directory that was profiled twice holds both runs' files (named by process id) — summarising a mix is worse than failing"""
    files = glob.glob(pattern)
    if len(files) > 1:
        sys.exit("summarize_profile: %d files match %s — remove the directory before profiling into it again" % (len(files), pattern))
    return files


def source_hash():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from loam_amd import build
    return build.source_hash()


def short(name):
    m = re.search(r"(\w+_kernel(?:<[\w, ]+>)?)", name)
    return m.group(1) if m else name[:48]


def bench_json(path):
    if not os.path.exists(path):
        return None
    for line in open(path):
        if line.startswith("{"):
            return json.loads(line)
    return None


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    md = ["# rocprofv3 summary (" + os.path.basename(src) + ")", ""]
    trace_tables = {}
    for sub, log, title in (("trace", "bench_under_rocprof.log", "`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-streamed`"),
                            ("trace_seq", "bench_under_rocprof_seq.log", "the same with `LOAMX_NO_AUX_STREAM=1` (association chains in sequence on one stream: "
                             "per-kernel durations without the inflation that concurrent kernels report)")):
        stats = one_pass(os.path.join(src, sub, "*", "*_kernel_stats.csv"))
        if not stats:
            continue
        rows = list(csv.DictReader(open(stats[0])))
        with open(dst + ("_kernel_stats.csv" if sub == "trace" else "_kernel_stats_seq.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ns", "average_ns", "percentage"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
        trace_tables[sub] = {short(r["Name"]): {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                                                "share": round(float(r["Percentage"]) / 100.0, 4)} for r in rows}
        md += ["## Kernel time: " + title, "", "| kernel | calls | avg µs | total ms | % |", "|---|---|---|---|---|"]
        for r in rows:
            if float(r["Percentage"]) < 0.01:
                continue
            md.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | "
                      f"{float(r['TotalDurationNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |")
        bj = bench_json(os.path.join(src, log))
        if bj:
            md += ["", f"bench.py under the profiler: {bj['value']} {bj['unit']}, {bj['ms_per_step']} ms/step; "
                       "HIP-event averages measured inside bench.py for the same run:", "",
                   "| kernel (event scope) | launches | avg ms | algorithmic GB/s | frac of 8 TB/s |", "|---|---|---|---|---|"]
            for k, v in bj["kernels"].items():
                md.append(f"| {k} | {v['launches']} | {v['avg_ms']} | {v['achieved_GBs']} | {v['hbm_frac']} |")
        md.append("")
    pmc = {}
    for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        files = one_pass(os.path.join(src, sub, "*", "*_counter_collection.csv"))
        if not files:
            continue
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] != ctr:
                continue
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
        for k, (n, v) in agg.items():
            pmc.setdefault(k, {})[ctr] = {"dispatches": n, "kib_total_reported": v}
    if pmc:
        bj = bench_json(os.path.join(src, "pmc_fetch.log")) or {}
        kern = bj.get("kernels", {})
        md += ["", "## HBM traffic (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate passes)", "",
               "bytes = FETCH_SIZE[KiB] x 1024 x 2 (gfx950 half-count correction) + WRITE_SIZE[KiB] x 1024; per dispatch.", "",
               "| kernel | dispatches | fetch MB (corrected) | write MB | traffic MB | algorithmic MB | traffic / algorithmic |",
               "|---|---|---|---|---|---|---|"]
        out = {}
        alias = {"curvature_valid_kernel": "curvature_valid_kernel", "sweep_kernel": "sweep_kernel", "moment_kernel": "moment_kernel",
                 "select_kernel<4>": "select_kernel", "select_mis_kernel<2, 4>": "select_kernel"}
        alias.update({k: "select_kernel" for k in pmc if k.startswith("select_mis_kernel") or k.startswith("select_rows_kernel")})
        for k in sorted(pmc):
            f = pmc[k].get("FETCH_SIZE", {"dispatches": 0, "kib_total_reported": 0.0})
            w = pmc[k].get("WRITE_SIZE", {"dispatches": 0, "kib_total_reported": 0.0})
            n = max(f["dispatches"], w["dispatches"], 1)
            fetch = f["kib_total_reported"] * 1024 * 2 / n
            write = w["kib_total_reported"] * 1024 / n
            algo = None
            if k in alias and alias[k] in kern:
                algo = kern[alias[k]]["algorithmic_bytes_per_launch"]
            if k.startswith("associate_kernel") and "associate_kernel" in kern:
                algo = None  # edge + plane launches share one event scope
            ratio = (fetch + write) / algo if algo else None
            out[k] = {"dispatches": n, "fetch_bytes_per_dispatch": fetch, "write_bytes_per_dispatch": write,
                      "traffic_bytes_per_dispatch": fetch + write, "algorithmic_bytes_per_dispatch": algo,
                      "traffic_over_algorithmic": ratio}
            if fetch + write < 1e5:
                continue
            md.append(f"| {k} | {n} | {fetch/1e6:.2f} | {write/1e6:.2f} | {(fetch+write)/1e6:.2f} | "
                      f"{'' if algo is None else f'{algo/1e6:.2f}'} | {'' if ratio is None else f'{ratio:.3f}'} |")
        cfg = bj.get("config", {})
        scopes = {k: v["launches"] for k, v in kern.items()}  # HIP-event scopes of that run (associate = kNN + fit, edge + plane)
        pmc_json = {"source": os.path.basename(src), "source_sha256": (open(os.path.join(src, "source_sha256.txt")).read().strip() if os.path.exists(os.path.join(src, "source_sha256.txt")) else source_hash()), "bench_config": cfg, "event_scopes": scopes, "kernels": out,
                    "kernel_trace": trace_tables.get("trace", {}), "kernel_trace_sequential": trace_tables.get("trace_seq", {})}
        json.dump(pmc_json, open(dst + "_pmc.json", "w"), indent=1)
    # any further counter passes (pmc_x*): per-kernel average per dispatch
    extra = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for d in sorted(glob.glob(os.path.join(src, "pmc_x*"))):
        if not os.path.isdir(d):
            continue
        files = one_pass(os.path.join(d, "*", "*_counter_collection.csv"))
        if not files:  # (an aborted rocprofv3 pass leaves the directory without a counter file)
            continue
        for r in csv.DictReader(open(files[0])):
            a = extra[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    if extra and os.path.exists(dst + "_pmc.json"):  # averages per dispatch, for bench.py's compute roofline
        pj = json.load(open(dst + "_pmc.json"))
        pj["counters"] = {k: {c: v[1] / v[0] for c, v in extra[k].items()} for k in extra}
        json.dump(pj, open(dst + "_pmc.json", "w"), indent=1)
    if extra:
        ctrs = sorted({c for k in extra for c in extra[k]})
        md += ["", "## Other counters (average per dispatch, one `--pmc` pass per set)", "",
               "| kernel | " + " | ".join(ctrs) + " |", "|---|" + "---|" * len(ctrs)]
        for k in sorted(extra):
            if not re.search(r"knn|fit|select|compact|grid_build|curvature|sweep", k):
                continue
            md.append(f"| {k} | " + " | ".join(f"{extra[k][c][1] / extra[k][c][0]:.4g}" if c in extra[k] else "" for c in ctrs) + " |")
    open(dst + "_summary.md", "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()

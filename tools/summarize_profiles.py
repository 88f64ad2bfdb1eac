#!/usr/bin/env python3
"""Condense one round's rocprofv3 output (gpurun_out/rNN/) into profiles/ (tracked).

  python tools/summarize_profiles.py gpurun_out/r01 r01

Writes profiles/<tag>_kernel_stats.csv (our kernels' rows of `rocprofv3 --kernel-trace --stats`),
profiles/<tag>_summary.md and profiles/pmc_traffic.json (what bench.py reports as roofline.traffic).
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE come from
SEPARATE --pmc passes, are in KiB, and on gfx950 FETCH_SIZE counts a 16-B-per-lane streaming read at
exactly half its bytes, so reads = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"gcp_scan_main<0": "cumprod_fwd", "gcp_scan_main<1": "cumsum_fwd", "gcp_scan_main<2": "cumprod_bwd",
         "gcp_scan_main<3": "cumsum_rev", "gcp_fallback<0": "fallback_fwd", "gcp_fallback<2": "fallback_bwd"}


def short(kernel):
    for k, v in NAMES.items():
        if k in kernel:
            return v
    return None


def pmc(dirname, counter):
    acc = collections.defaultdict(list)
    files = sorted(glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:  # gpurun MERGES into gpurun_out/: only the newest run counts
        for r in csv.DictReader(open(f)):
            s = short(r["Kernel_Name"])
            if s and r["Counter_Name"] == counter:
                acc[s].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    stats = max(glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(stats)))
    # the library's kernels: the scans (gcp_*) and, from bench.py's function_level / caller_level legs, k_* (binning,
    # blend, sort, projection, loss)
    ours = [r for r in rows if "gcp_" in r["Name"] or "::k_" in r["Name"]]
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(ours)
    bench = json.loads(open(os.path.join(src, "bench_stats.json")).read().strip().splitlines()[-1])
    plain = None
    if os.path.exists(os.path.join(src, "bench_plain.json")):
        plain = json.loads(open(os.path.join(src, "bench_plain.json")).read().strip().splitlines()[-1])
    m = bench["config"]["pairs_per_gpu"]
    fetch, write = pmc(os.path.join(src, "pmc_fetch"), "FETCH_SIZE"), pmc(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    traffic = {}
    lines = [f"# {tag}: rocprofv3 summary of `python3 bench.py` (cfg3, M = {m} pairs)", "",
             "| kernel | calls | avg us (rocprof) | algorithmic B/launch | algorithmic GB/s | FETCH_SIZE KiB (raw) | "
             "WRITE_SIZE KiB | HBM bytes/launch (2*FETCH+WRITE) | traffic / algorithmic | HBM GB/s |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    per = {"cumprod_fwd": 12, "cumprod_bwd": 20, "cumsum_fwd": 12, "cumsum_rev": 12}
    for r in ours:
        s = short(r["Name"])
        avg_us = float(r["AverageNs"]) / 1e3
        if s in per:
            alg = per[s] * m
            hb = None
            if s in fetch and s in write:
                hb = (2.0 * fetch[s] + write[s]) * 1024.0
                traffic[s] = {"hbm_bytes_per_launch": hb, "fetch_size_kib_raw": fetch[s], "write_size_kib": write[s],
                              "algorithmic_bytes_per_launch": alg, "avg_us_rocprof": avg_us}
            lines.append(f"| {s} | {r['Calls']} | {avg_us:.1f} | {alg} | {alg/avg_us/1e3:.0f} | {fetch.get(s, float('nan')):.0f} | "
                         f"{write.get(s, float('nan')):.0f} | {hb if hb else 'n/a':.4g} | {hb/alg if hb else float('nan'):.3f} | "
                         f"{hb/avg_us/1e3 if hb else float('nan'):.0f} |")
        else:
            lines.append(f"| {s} | {r['Calls']} | {avg_us:.2f} | - | - | {fetch.get(s, 0):.1f} | {write.get(s, 0):.1f} | - | - | - |")
    lines += ["", "bench.py under the profiler (HIP events on the launch stream, op = main kernel + fallback launch):", "",
              "```", json.dumps({k: bench[k] for k in ("value", "ms_per_step")}),
              json.dumps(bench["roofline"]), "```"]
    if plain:
        lines += ["", "bench.py without the profiler, same box:", "", "```",
                  json.dumps({k: plain[k] for k in ("value", "ms_per_step")}), json.dumps(plain["roofline"]),
                  json.dumps(plain.get("cpu_baseline")), "```"]
    open(os.path.join(out, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    path = os.path.join(out, "pmc_traffic.json")
    allt = json.load(open(path)) if os.path.exists(path) else {}
    allt["cfg3"] = traffic
    allt["_source"] = f"{tag}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py (separate runs); reads doubled per the gfx950 correction"
    json.dump(allt, open(path, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()

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


def pmc(dirname, counter, blocks=None):
    """kernel -> mean of `counter` over the launches of `blocks` workgroups (None: all sizes)."""
    acc = collections.defaultdict(list)
    files = sorted(glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:  # gpurun MERGES into gpurun_out/: only the newest run counts
        for r in csv.DictReader(open(f)):
            s = short(r["Kernel_Name"])
            if s and r["Counter_Name"] == counter:
                if blocks is not None and s in ("cumprod_fwd", "cumprod_bwd", "cumsum_fwd", "cumsum_rev") and \
                        abs(int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])) - blocks) > 3:
                    continue  # the same kernel on another workload of the same command (cfg2 block, wrappers, ...)
                acc[s].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    stats = max(glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    raw = list(csv.DictReader(open(stats)))
    with open(os.path.join(out, f"{tag}_kernel_stats_raw.csv"), "w", newline="") as f:  # rocprofv3's own --stats rows, ours only
        w = csv.DictWriter(f, fieldnames=raw[0].keys())
        w.writeheader()
        w.writerows([r for r in raw if "gcp_" in r["Name"] or "::k_" in r["Name"]])
    # bench.py launches the same kernels on several workloads (the timed cfg3 list, the unclipped variant, the cfg5 / cfg3
    # sharded frames): --stats averages over all of them, so the rows below are re-aggregated from the kernel trace PER
    # GRID SIZE (= tiles of the launch)
    trace = max(glob.glob(os.path.join(src, "stats", "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    groups = collections.OrderedDict()
    for r in csv.DictReader(open(trace)):
        if "gcp_" not in r["Kernel_Name"] and "::k_" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
        groups.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows = []
    for (name, blocks), d in groups.items():
        rows.append({"Name": name, "Blocks": blocks, "Calls": len(d), "TotalDurationNs": sum(d), "AverageNs": sum(d) / len(d),
                     "MinNs": min(d), "MaxNs": max(d)})
    rows.sort(key=lambda r: -r["TotalDurationNs"])
    ours = rows
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
    bench = json.loads(open(os.path.join(src, "bench_stats.json")).read().strip().splitlines()[-1])
    plain = None
    if os.path.exists(os.path.join(src, "bench_plain.json")):
        plain = json.loads(open(os.path.join(src, "bench_plain.json")).read().strip().splitlines()[-1])
    m = bench["config"]["pairs_per_gpu"]
    hb_blocks = (m + 4095) // 4096
    fetch, write = pmc(os.path.join(src, "pmc_fetch"), "FETCH_SIZE", hb_blocks), pmc(os.path.join(src, "pmc_write"), "WRITE_SIZE", hb_blocks)
    traffic = {}
    lines = [f"# {tag}: rocprofv3 summary of `python3 bench.py` (cfg3, M = {m} pairs = {(m + 4095) // 4096} tiles)", "",
             "Rows are aggregated from the kernel trace per (kernel, grid size): the headline's timed region is the "
             f"{(m + 4095) // 4096}-tile launches of cumprod_fwd / cumprod_bwd; the other sizes of the same kernels belong to the "
             "`unclipped`, `cfg2`, `wrapper_level` and `sharded_frames` blocks of the same command (rocprofv3's own --stats rows, which average over all of "
             f"them, are in {tag}_kernel_stats_raw.csv).  PMC traffic is from the separate --pmc passes, averaged over the launches "
             "of those passes (GCP_BENCH_NO_SHARDED=1: the timed workload and its unclipped twin, same size within 0.01 %).", "",
             "| kernel | calls | avg us (rocprof) | algorithmic B/launch | algorithmic GB/s | FETCH_SIZE KiB (raw) | "
             "WRITE_SIZE KiB | HBM bytes/launch (2*FETCH+WRITE) | traffic / algorithmic | HBM GB/s |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    per = {"cumprod_fwd": 12, "cumprod_bwd": 20, "cumsum_fwd": 12, "cumsum_rev": 12}
    tile = 4096
    headline_blocks = (m + tile - 1) // tile
    for r in ours:
        s = short(r["Name"])
        if s is None:
            continue  # kernels of the wrapper / Function / caller blocks of the same command: <tag>_wrappers.md, <tag>_function_kernels.md
        avg_us = float(r["AverageNs"]) / 1e3
        if s in per and r["Blocks"] != headline_blocks:
            lines.append(f"| {s} ({r['Blocks']} tiles: another workload of the same run) | {r['Calls']} | {avg_us:.1f} | - | "
                         f"{per[s] * r['Blocks'] * tile / avg_us / 1e3:.0f} (upper bound: last tile partial) | - | - | - | - | - |")
            continue
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
            name = s or r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
            lines.append(f"| {name} | {r['Calls']} | {avg_us:.2f} | - | - | {fetch.get(s, 0):.1f} | {write.get(s, 0):.1f} | - | - | - |")
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
    long_groups(src, tag, out)
    function_kernels(src, tag, out)
    wrappers(src, tag, out)


def per_launch(dirname, counter, pick):
    """kernel -> list of per-launch values of `counter` in launch order, newest run of `dirname`."""
    acc = collections.defaultdict(list)
    files = sorted(glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            k = pick(r["Kernel_Name"])
            if k and r["Counter_Name"] == counter:
                acc[k].append(float(r["Counter_Value"]))
    return acc


def long_groups(src, tag, out):
    """Traffic of the scans on inputs whose groups are far longer than a tile (tools/pathological_bench.py "whole array"
    "5000": two cases x two modes, forward scans then backward scans per case; 14 launches of each per case)."""
    if not os.path.isdir(os.path.join(src, "long_fetch")):
        return
    n = 166_000_000
    fetch = per_launch(os.path.join(src, "long_fetch"), "FETCH_SIZE", short)
    write = per_launch(os.path.join(src, "long_write"), "WRITE_SIZE", short)
    lines = [f"# {tag}: scans on groups far longer than one tile (n = {n}, tools/pathological_bench.py)", "",
             "Timing (same script, no profiler):", "", "```",
             *[ln for ln in open(os.path.join(src, "long_groups.txt")).read().splitlines() if "amdgpu.ids" not in ln], "```", "",
             "HBM traffic per launch from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (reads = 2 x FETCH_SIZE "
             "KiB x 1024 on gfx950, writes = WRITE_SIZE KiB x 1024); launches in order: case 'one group = whole array' then "
             "'groups of 5000-13000', first with the descriptor tree (main kernel does everything, the follow-up kernel is a "
             "no-op), then with the walk switched off (two-pass: the follow-up kernel re-reads and re-writes the leading "
             "elements of every unresolved tile).", "",
             "| kernel | mode | case | launches | HBM bytes / launch | algorithmic bytes | traffic / algorithmic |", "|---|---|---|---|---|---|---|"]
    per = {"cumprod_fwd": 12, "cumprod_bwd": 20}
    fb = {"cumprod_fwd": "fallback_fwd", "cumprod_bwd": "fallback_bwd"}
    for k, bpe in per.items():
        f, w = fetch.get(k, []), write.get(k, [])
        ff, wf = fetch.get(fb[k], []), write.get(fb[k], [])
        if not f or len(f) != len(w) or len(f) % 4:
            continue
        q = len(f) // 4
        for i, (mode, case) in enumerate((("tree", "one group"), ("tree", "groups 5-13k"), ("two-pass", "one group"), ("two-pass", "groups 5-13k"))):
            hb = sum((2 * a + b) * 1024 for a, b in zip(f[i * q:(i + 1) * q], w[i * q:(i + 1) * q])) / q
            hf = sum((2 * a + b) * 1024 for a, b in zip(ff[i * q:(i + 1) * q], wf[i * q:(i + 1) * q])) / q if len(ff) == len(f) else float("nan")
            lines.append(f"| {k} + follow-up | {mode} | {case} | {q} | {hb:.4g} + {hf:.3g} | {bpe * n:.4g} | {(hb + hf) / (bpe * n):.3f} |")
    open(os.path.join(out, f"{tag}_long_groups.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def function_kernels(src, tag, out):
    """Per-kernel durations and SQ counters of the Function's kernels (tools/raster_bench.py, cfg3 scene)."""
    if not os.path.isdir(os.path.join(src, "fn_pmc")):
        return
    pick = lambda k: next((n for n in ("k_blend_bwd", "k_blend_fwd<true>", "k_blend_fwd<false>", "k_grad_reduce") if n in k), None)  # noqa: E731
    counters = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
                "SQ_INSTS_LDS", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE"]
    vals = {c: {k: sum(v) / len(v) for k, v in per_launch(os.path.join(src, "fn_pmc"), c, pick).items()} for c in counters}
    stats = max(glob.glob(os.path.join(src, "fn_stats", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    dur = {}
    for r in csv.DictReader(open(stats)):
        k = pick(r["Name"])
        if k:
            dur[k] = float(r["AverageNs"]) / 1e3
    lines = [f"# {tag}: the Function's kernels on the cfg3 scene (1920x1080, 10^6 Gaussians, 1.65e8 pairs, 3.0e6 tile entries)", "",
             "`rocprofv3 --kernel-trace --stats` and a separate `--pmc` pass of `python3 tools/raster_bench.py --no-cameras`; counters are "
             "per launch, summed over the chip; the *_CYCLES / WAIT / ACTIVE ones count quad-cycles per wave.", "",
             "| kernel | avg us | " + " | ".join(counters) + " | VALU issue time at 1024 SIMDs x 2.4 GHz / 4 (us) |", "|---|---|" + "---|" * (len(counters) + 1)]
    pj = {}
    for k in dur:
        row = [f"{vals[c].get(k, float('nan')):.4g}" for c in counters]
        iv = vals["SQ_INSTS_VALU"].get(k)
        lines.append(f"| {k} | {dur[k]:.1f} | " + " | ".join(row) + f" | {iv / (1024 * 2.4e9 / 4) * 1e6 if iv else float('nan'):.0f} |")
        pj[k] = {c: vals[c].get(k) for c in counters}
        pj[k]["avg_us_rocprof"] = dur[k]
    keep = [ln for ln in open(os.path.join(src, "fn_bench.txt")).read().splitlines()
            if ln.strip() and "rocprofv3" not in ln and not ln.startswith(("W2", "E2", "I2")) and "amdgpu.ids" not in ln]
    lanes = {k: (vals["SQ_THREAD_CYCLES_VALU"].get(k) or 0) / max(vals["SQ_INSTS_VALU"].get(k) or 1, 1) for k in dur}
    lines += ["", "SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = lanes ENABLED per VALU instruction (of 64): "
              + ", ".join(f"{k} {v:.1f}" for k, v in lanes.items()) + ".  The backward's body is branch-free (selects), so "
              "nearly all lanes are enabled whether or not their pixel lies in the entry's box; the forward masks with exec.  "
              "Neither says how many lanes did USEFUL work: that fraction (pixels inside the visited entry's box) is computed from "
              "the tile lists by bench.py (function_level.roofline.active_lane_frac = 0.375 on this scene).", "", "```", *keep, "```"]
    open(os.path.join(out, f"{tag}_function_kernels.md"), "w").write("\n".join(lines) + "\n")
    path = os.path.join(out, "pmc_function.json")
    json.dump({"cfg3": pj, "_source": f"{tag}: rocprofv3 --pmc (SQ counters) pass of tools/raster_bench.py"}, open(path, "w"), indent=1)
    print("\n".join(lines))


WRAPPER_KERNELS = [  # (name fragment, label, algorithmic bytes per pair, route)
    ("k_rect_rows_local", "rect list -> rows: flags, per-tile row records, coordinate range (reads the rects once)", 8, "auto"),
    ("k_rect_rows_gather", "rows: records to their final places (kept packed: 8 B per row)", 0.6, "auto"),
    ("k_rows_rectangles<false, true>", "rows -> rectangles: count (16 rows per thread, straight-line loads)", 0.3, "auto"),
    ("k_rows_rectangles<true, true>", "rows -> rectangles: write", 0.3, "auto"),
    ("k_rectangle_boxes", "rectangles -> boxes, offsets, tiles per box (the binning's counting pass)", 0.2, "auto"),
    ("k_sort_hist2<true>", "sort pass 0: histogram (reads the rects)", 8, "rects"),
    ("k_sort_scatter2<true, true>", "sort pass 0: scatter (rects -> key + index)", 16, "rects"),
    ("k_sort_hist2<false>", "sort passes 1, 2: histogram (per pass)", 4, "rects"),
    ("k_sort_scatter2<false, false>", "sort passes 1, 2: scatter (per pass)", 16, "rects"),
    ("gcp_scan_main<0, true, false, true", "indexed scan, cumprod (gather + scan + un-sort)", 16, "rects"),
    ("gcp_scan_main<3, true, false, true", "indexed scan, suffix sum (grad_cumsum)", 16, "rects"),
    ("k_compact<true, false>", "compaction: count pass (sort route only)", 4, "rects"),
    ("k_compact<true, true>", "compaction: write pass (mask + kept values; sort route only)", 13, "rects"),
    ("k_pairs_scan_boxes<0, false, 2>", "tile-list walk, cumprod: writes FINAL values (inclusive / self) + clears the mask bytes of dropped pairs", 8, "boxes"),
    ("k_pairs_scan_boxes<2, false, 2>", "tile-list walk, suffix sum: writes FINAL values (inclusive - self) + mask", 8, "boxes"),
    ("k_compact_kept", "kept values moved together (only when the walk dropped something)", 9, "boxes"),
]


def wrappers(src, tag, out):
    """Rows a5 / a6: kernel table of `tools/wrapper_bench.py cfg3 --profile` with the FETCH / WRITE passes beside it."""
    if not os.path.isdir(os.path.join(src, "wr_stats")):
        return
    trace = max(glob.glob(os.path.join(src, "wr_stats", "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(trace)))
    m = json.loads([ln for ln in open(os.path.join(src, "wr_prof.txt")).read().splitlines() if ln.startswith("{")][-1])["pairs"]
    pick = lambda k: next((label for frag, label, _, _ in WRAPPER_KERNELS if frag in k), None)  # noqa: E731
    dur = collections.defaultdict(list)
    other = collections.defaultdict(list)
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        lab = pick(r["Kernel_Name"])
        (dur[lab] if lab else other[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]]).append(d)
    fetch = {k: sum(v) / len(v) for k, v in per_launch(os.path.join(src, "wr_fetch"), "FETCH_SIZE", pick).items()}
    write = {k: sum(v) / len(v) for k, v in per_launch(os.path.join(src, "wr_write"), "WRITE_SIZE", pick).items()}
    lines = [f"# {tag}: rows a5 / a6 — create_alpha_brend / grad_cumsum (gs_model.py:544-566, :716-722) at the cfg3 scene, M = {m} pairs", "",
             "`rocprofv3 --kernel-trace --stats` of `python3 tools/wrapper_bench.py cfg3 --profile` (create_alpha_brend(rects) and "
             "grad_cumsum(rects) by the default route — the list cut into boxes (one call, one read-back) and walked, the walk "
             "writing the final values and the mask — and by the general sort route, create_alpha_brend_boxes, grad_cumsum_boxes, 4 times each) and separate `--pmc FETCH_SIZE` / `--pmc "
             "WRITE_SIZE` passes of the same command.  HBM bytes = 2 x FETCH_SIZE KiB x 1024 + WRITE_SIZE KiB x 1024 (gfx950 "
             "correction of the guide; FETCH_SIZE halves only wide streaming reads, so for the gather / scatter kernels the figure "
             "is an upper bound).", "",
             "| kernel | launches | avg us | algorithmic B / pair | algorithmic GB/s | frac of 8 TB/s | HBM bytes / launch (PMC) | traffic / algorithmic |",
             "|---|---|---|---|---|---|---|---|"]
    for frag, label, bpp, route in WRAPPER_KERNELS:
        if label not in dur:
            continue
        avg = sum(dur[label]) / len(dur[label])
        alg = bpp * m
        hb = (2.0 * fetch[label] + write[label]) * 1024.0 if label in fetch and label in write else None
        lines.append(f"| {label} | {len(dur[label])} | {avg:.1f} | {bpp} | {alg / avg / 1e3:.0f} | {alg / avg / 1e3 / 8000:.3f} | "
                     f"{hb:.4g} | {hb / alg:.2f} |" if hb else f"| {label} | {len(dur[label])} | {avg:.1f} | {bpp} | {alg / avg / 1e3:.0f} | {alg / avg / 1e3 / 8000:.3f} | n/a | n/a |")
    lines += ["", "Small kernels of the same calls (K-sized binning, prefix sums, follow-up launches):", "", "| kernel | launches | avg us |", "|---|---|---|"]
    for k, v in sorted(other.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) / len(v) > 2.0 and ("k_" in k or "gcp_" in k):
            lines.append(f"| {k} | {len(v)} | {sum(v) / len(v):.1f} |")
    if os.path.exists(os.path.join(src, "wr_bench.txt")):
        keep = [ln for ln in open(os.path.join(src, "wr_bench.txt")).read().splitlines() if ln.startswith("{")]
        lines += ["", "Whole calls and stages, HIP events around the Python calls, no profiler (`tools/wrapper_bench.py cfg2 cfg3 --stages`):", "", "```", *keep, "```"]
    if os.path.exists(os.path.join(src, "pairing.txt")):
        keep = [ln for ln in open(os.path.join(src, "pairing.txt")).read().splitlines() if ln.startswith("{")]
        open(os.path.join(out, f"{tag}_blend_pairing.jsonl"), "w").write("\n".join(keep) + "\n")
    open(os.path.join(out, f"{tag}_wrappers.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

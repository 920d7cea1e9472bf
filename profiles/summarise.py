"""Condenses the rocprofv3 CSVs under gpurun_out/prof_<tag>/ into the committed summaries:
profiles/<tag>_kernel_stats.txt (per-kernel calls / total / average duration), profiles/<tag>_pmc.txt
(counter totals per kernel) and profiles/traffic.json (HBM bytes per launch of each dominant kernel).

HBM bytes follow the gfx950 rule that FETCH_SIZE reports half the bytes of a wide coalesced streaming read:
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch (both counters are in KiB)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(root, "gpurun_out", f"prof_{tag}")


def rows(pattern):
    # gpurun merges every run's files into the same local directory: keep the newest run of each directory
    # (rocprofv3 prefixes its files with the process id, so compare modification times, not names)
    paths = glob.glob(os.path.join(base, pattern), recursive=True)
    newest = {}
    for path in paths:
        d = os.path.dirname(path)
        if d not in newest or os.path.getmtime(path) > os.path.getmtime(newest[d]):
            newest[d] = path
    for path in newest.values():
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                yield r


def short(name):
    name = name.split("(")[0]
    for k in ("direct_kernel", "direct_finish_kernel", "walk_kernel", "radix_scatter_kernel", "radix_hist_kernel",
              "scan_reduce_kernel", "scan_sums_kernel", "scan_apply_kernel", "keys_kernel", "permute_kernel",
              "radix_tree_kernel", "mark_cells_kernel", "emit_nodes_kernel", "cell_mass_kernel", "carry_dead_kernel",
              "refresh_mirror_kernel", "kinetic_kernel", "potential_kernel", "psum_apply_kernel", "psum_reduce_kernel",
              "psum_scan_sums_kernel", "scan_rows_kernel", "scan_single_kernel", "ref_com_kernel"):
        if k in name:
            return k
    return name[:60]


# ---- kernel traces ----
durations = {}
with open(os.path.join(root, "profiles", f"{tag}_kernel_stats.txt"), "w") as f:
    for sub, cmd in (("trace_direct", "python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra"),
                     ("trace_tree", "python3 bench.py --workload tree --steps 20 --warmup 2 --no-cpu-baseline --no-extra")):
        agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
        for r in rows(f"{sub}/**/*kernel_trace.csv"):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3   # us
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
        total = sum(a[1] for a in agg.values()) or 1.0
        durations[sub] = {k: a[1] / a[0] for k, a in agg.items() if a[0]}
        f.write(f"# rocprofv3 --kernel-trace --stats -- {cmd}   (tag {tag})\n")
        f.write(f"{'kernel':28s} {'calls':>7s} {'total_us':>14s} {'avg_us':>12s} {'min_us':>12s} {'max_us':>12s} {'share':>7s}\n")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{k:28s} {a[0]:7d} {a[1]:14.1f} {a[1] / a[0]:12.2f} {a[2]:12.2f} {a[3]:12.2f} {100 * a[1] / total:6.2f}%\n")
        f.write("\n")

# ---- the distributed tree: eight ranks side by side on one GPU ----
agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
for r in rows("trace_dist/**/*kernel_trace.csv"):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    name = name.split("<")[0]
    a = agg[name]
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
if agg:
    with open(os.path.join(root, "profiles", f"{tag}_dist_kernel_stats.txt"), "w") as f:
        total = sum(a[1] for a in agg.values())
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 profiles/dist_step.py   (tag {tag})\n"
                f"# key-range-distributed tree, 16,777,216 bodies, eight ranks on one GPU, 8 steps (the first moves every body\n"
                f"# to its owner: classify_kernel and pack_kernel are dominated by it); per rank and step divide calls by 64\n")
        f.write(f"{'kernel':32s} {'calls':>7s} {'total_us':>14s} {'avg_us':>12s} {'min_us':>12s} {'max_us':>12s} {'share':>7s}\n")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{k:32s} {a[0]:7d} {a[1]:14.1f} {a[1] / a[0]:12.2f} {a[2]:12.2f} {a[3]:12.2f} {100 * a[1] / total:6.2f}%\n")
        log = os.path.join(base, "dist_step.log")
        if os.path.exists(log):
            f.write("\n# " + "# ".join(l for l in open(log) if l.startswith("phase")))

# ---- the two 16M-body workloads (benchmark configurations 4 and 5): kernels by precision ----
agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
for r in rows("trace_walk16/**/*kernel_trace.csv"):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    full = r["Kernel_Name"].replace("void ", "")
    name = full.split("(")[0].split("<")[0]
    if name.startswith("walk"):
        args = [a.strip() for a in full.split("(")[0].split("<", 1)[1].rstrip(">").split(",")]
        if name == "walk_pair_kernel":      # <A, OUT, COUNT>
            prec, counting = ("mixed" if args[0] == "double" else "f32"), args[2] == "true"
        else:                               # <R, A, NodeT, OUT, COUNT, SOFT>
            prec = "f64" if args[0] == "double" else "mixed" if args[1] == "double" else "f32"
            counting = args[4] == "true"
        name += f"<{prec}>" + (" instrumented" if counting else "")
    a = agg[name]
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
if agg:
    with open(os.path.join(root, "profiles", f"{tag}_walk_16M_kernel_stats.txt"), "w") as f:
        total = sum(a[1] for a in agg.values())
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 profiles/walk_16M.py   (tag {tag})\n"
                f"# 16,777,216 bodies, theta 0.5, six steps each of the fp32 Plummer sphere and of the two 8M-body galaxies in mixed\n"
                f"# precision (eps = 1 m); both walk two bodies per lane\n")
        f.write(f"{'kernel':32s} {'calls':>7s} {'total_us':>14s} {'avg_us':>12s} {'min_us':>12s} {'max_us':>12s} {'share':>7s}\n")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
            f.write(f"{k:32s} {a[0]:7d} {a[1]:14.1f} {a[1] / a[0]:12.2f} {a[2]:12.2f} {a[3]:12.2f} {100 * a[1] / total:6.2f}%\n")
        log = os.path.join(base, "walk_16M.log")
        if os.path.exists(log):
            f.write("\n# " + "# ".join(l for l in open(log) if "walk" in l))

# ---- counters ----
pmc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
def walk_name(full):
    """walk kernels by precision and instrumentation (the two 16M-body workloads run several of them)"""
    name = full.replace("void ", "").split("(")[0].split("<")[0]
    args = [a.strip() for a in full.split("(")[0].split("<", 1)[1].rstrip(">").split(",")]
    if name == "walk_pair_kernel":
        prec, counting = ("mixed" if args[0] == "double" else "f32"), args[2] == "true"
    else:
        prec = "f64" if args[0] == "double" else "mixed" if args[1] == "double" else "f32"
        counting = args[4] == "true"
    return name + f"<{prec}>" + (" instrumented" if counting else "")


for sub in ("pmc_fetch_direct", "pmc_write_direct", "pmc_fetch_tree", "pmc_write_tree", "pmc_sq_direct", "pmc_sq_tree",
            "pmc_lds_tree", "pmc_grbm_direct", "pmc_sq_walk16"):
    for r in rows(f"{sub}/**/*counter_collection.csv"):
        kname = r["Kernel_Name"]
        key = walk_name(kname) if sub == "pmc_sq_walk16" and "walk" in kname.split("(")[0] else short(kname)
        a = pmc[sub][(key, r["Counter_Name"])]
        a[0] += 1; a[1] += float(r["Counter_Value"])
with open(os.path.join(root, "profiles", f"{tag}_pmc.txt"), "w") as f:
    f.write(f"# rocprofv3 --pmc passes (one counter group per run), tag {tag}; value = mean per dispatch\n")
    for sub in sorted(pmc):
        f.write(f"\n[{sub}]\n")
        for (k, c), a in sorted(pmc[sub].items()):
            f.write(f"{k:36s} {c:24s} dispatches {a[0]:6d}  mean {a[1] / a[0]:18.1f}\n")


def mean(sub, kernel, counter):
    a = pmc.get(sub, {}).get((kernel, counter))
    return a[1] / a[0] if a and a[0] else None


traffic = {}
for workload, kernel, fs, ws in (("direct_fp32_plummer_1M", "direct_kernel", "pmc_fetch_direct", "pmc_write_direct"),
                                 ("bh_theta0.5_fp32_plummer_1M", "walk_kernel", "pmc_fetch_tree", "pmc_write_tree")):
    fe, wr = mean(fs, kernel, "FETCH_SIZE"), mean(ws, kernel, "WRITE_SIZE")
    if fe is not None and wr is not None:
        traffic[workload] = {"kernel": kernel, "FETCH_SIZE_KiB": fe, "WRITE_SIZE_KiB": wr,
                             "hbm_bytes_per_launch": (2 * fe + wr) * 1024, "tag": tag,
                             "rule": "(2*FETCH_SIZE + WRITE_SIZE) * 1024: gfx950 FETCH_SIZE counts half of a wide coalesced read"}
# the clock the direct-sum kernel ran at: GRBM_GUI_ACTIVE is summed over the 8 XCDs; / 8 / kernel time (kernel-trace pass)
ga = mean("pmc_grbm_direct", "direct_kernel", "GRBM_GUI_ACTIVE")
dk = durations.get("trace_direct", {}).get("direct_kernel")
if ga is not None and dk and "direct_fp32_plummer_1M" in traffic:
    traffic["direct_fp32_plummer_1M"]["clock_ghz"] = ga / 8.0 / (dk * 1e-6) / 1e9
# sort and build phases of the Barnes-Hut step: HBM bytes per step summed over their kernels (launches per step x bytes)
SORT = ("keys_kernel", "radix_hist_kernel", "scan_rows_kernel", "radix_scatter_kernel", "permute_kernel")
BUILD = ("radix_tree_kernel", "scan_reduce_kernel", "scan_sums_kernel", "scan_apply_kernel", "psum_reduce_kernel",
         "psum_scan_sums_kernel", "psum_apply_kernel", "emit_nodes_kernel")
walk_calls = pmc.get("pmc_fetch_tree", {}).get(("walk_kernel", "FETCH_SIZE"), [0])[0]
tree_calls = pmc.get("pmc_fetch_tree", {}).get(("radix_tree_kernel", "FETCH_SIZE"), [0])[0]
if tree_calls:
    for phase, names in (("sort", SORT), ("build", BUILD)):
        tot = 0.0
        for k in names:
            fa, wa = pmc["pmc_fetch_tree"].get((k, "FETCH_SIZE")), pmc["pmc_write_tree"].get((k, "WRITE_SIZE"))
            if fa and wa:
                tot += (2 * fa[1] + wa[1]) * 1024 / tree_calls
        traffic.setdefault("bh_theta0.5_fp32_plummer_1M", {})[f"{phase}_hbm_bytes_per_step"] = tot
if traffic:
    with open(os.path.join(root, "profiles", "traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
# ---- HBM rate of every kernel of the Barnes-Hut step (sort and build phases are the HBM-bound ones) ----
with open(os.path.join(root, "profiles", f"{tag}_hbm_phases.txt"), "w") as f:
    f.write(f"# Barnes-Hut step, 1M-body Plummer sphere: HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes,\n"
            f"# (2*FETCH_SIZE + WRITE_SIZE) KiB, over the average duration of the kernel-trace pass (tag {tag}).\n"
            f"# Peak 8000 GB/s.  Kernels of a few microseconds are launch-latency bound at this size.  The factor 2 on\n"
            f"# FETCH_SIZE holds for wide coalesced reads; for 16-byte gathers (permute_kernel) it overstates the bytes.\n")
    f.write(f"{'kernel':28s} {'avg_us':>9s} {'fetch_KiB':>11s} {'write_KiB':>11s} {'GB/s':>9s} {'of_peak':>8s}\n")
    for k, us in sorted(durations.get("trace_tree", {}).items(), key=lambda kv: -kv[1]):
        fe, wr = mean("pmc_fetch_tree", k, "FETCH_SIZE"), mean("pmc_write_tree", k, "WRITE_SIZE")
        if fe is None or wr is None:
            continue
        gbs = (2 * fe + wr) * 1024 / (us * 1e-6) / 1e9
        f.write(f"{k:28s} {us:9.2f} {fe:11.1f} {wr:11.1f} {gbs:9.1f} {gbs / 8000:8.3f}\n")
print(open(os.path.join(root, "profiles", f"{tag}_kernel_stats.txt")).read())
print(open(os.path.join(root, "profiles", f"{tag}_hbm_phases.txt")).read())
print(json.dumps(traffic, indent=1))

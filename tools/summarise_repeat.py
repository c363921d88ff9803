"""min / median / max over separate-process rocprofv3 runs of the per-kernel averages (tools/prof_shards_repeat.sh)."""
import csv, glob, json, os, re, statistics, sys
d = sys.argv[1]
groups = {}
for f in sorted(glob.glob(os.path.join(d, "*_kernel_stats.csv"))):
    key = re.sub(r"_rep\d+_kernel_stats\.csv$", "", os.path.basename(f))
    meta = f.replace("_kernel_stats.csv", ".json")
    plan = None
    try:
        plan = json.loads([l for l in open(meta) if l.startswith("{")][-1])["plan"]
    except Exception:
        pass
    g = groups.setdefault(key, {"plan": plan, "kernels": {}})
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Name"])
        if not any(t in name for t in ("k_gemv_colsplit", "k_update_xr", "k_prefold_ap", "k_mailbox")) or int(r["Calls"]) < 100:
            continue
        if "tagged<true>" in name:      # the self-test's instantiation
            continue
        g["kernels"].setdefault(name, []).append((float(r["AverageNs"]) / 1e3, int(r["Calls"])))
out = {}
lines = ["| case | K1 plan | kernel | launches per run | avg us: min / median / max over runs | runs |", "|---|---|---|---|---|---|"]
for key, g in groups.items():
    out[key] = {"plan": g["plan"], "kernels": {}}
    for name, vals in sorted(g["kernels"].items()):
        us = [v[0] for v in vals]
        out[key]["kernels"][name] = {"avg_us_min": min(us), "avg_us_median": statistics.median(us), "avg_us_max": max(us),
                                    "runs": len(us), "calls_per_run": vals[0][1]}
        pl = g["plan"] or {}
        lines.append("| %s | R=%s U=%s light=%s split=%s | `%s` | %d | %.2f / %.2f / %.2f | %d |" % (
            key, pl.get("R"), pl.get("U"), pl.get("light"), pl.get("split"), name.replace("cgx::", "")[:60], vals[0][1],
            min(us), statistics.median(us), max(us), len(us)))
    it = sum(v["avg_us_median"] for k, v in out[key]["kernels"].items() if "<8, 2, 4, 0" not in k and "<4, 4, 4, 0" not in k and ", 0, " not in k)
    out[key]["sum_of_medians_us_fused_kernels"] = it
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
open(os.path.join(d, "summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

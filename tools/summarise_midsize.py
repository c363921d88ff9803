"""Table of tools/prof_midsize.sh: per size, from the dispatch timestamps of one rocprofv3 kernel trace of `cgsolver N out 300`
and three separate PMC passes -- the iteration's main kernel (K1, or the persistent kernel: then per iteration = duration / 300),
K3, the period of an iteration, the gaps, the fractions of the 8 TB/s peak, and HBM traffic over algorithmic bytes.
Reads /tmp/mid_trace_<n>.csv and /tmp/mid_pmc_<n>_<counter>.csv; writes the reduced per-dispatch CSVs beside the summary."""
import csv, os, re, statistics, sys

tag, sizes = sys.argv[1], [int(v) for v in sys.argv[2:]]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r05_midsize")
PEAK = 8.0e12
MAIN = re.compile(r"k_gemv_colsplit<\d+, \d+, \d+, 1|k_gemv_ldsp<\d+, \d+, \d+, 1|k_cg_resident|k_cg_stream")
K3 = re.compile(r"k_update_xr")
ITERS = 300


def short(name):
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def pmc(n, counter_file, counter):
    """mean per main-kernel dispatch, number of dispatches; the reduced CSV is kept"""
    path = "/tmp/mid_pmc_%d_%s.csv" % (n, counter_file)
    vals, keep, header = [], [], None
    with open(path) as fh:
        rd = csv.DictReader(fh)
        header = rd.fieldnames
        for r in rd:
            if not MAIN.search(r["Kernel_Name"]):
                continue
            keep.append(r)
            if r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    with open(os.path.join(out, "n%d_%s_%s_counter_collection.csv" % (n, tag, counter_file)), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=header)
        w.writeheader()
        w.writerows(keep[:64])   # the first 64 dispatches: enough to check the mean against
    return (sum(vals) / len(vals), len(vals)) if vals else (float("nan"), 0)


print("| N | main kernel | main us / iteration | K3 us | period us | gaps us (K1->K3 + K3->K1) | main frac of 8 TB/s | whole-iteration frac | "
      "FETCH x2 + WRITE over algorithmic | TCC hit / (hit + miss) | dispatches |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for n in sizes:
    rows = []
    with open("/tmp/mid_trace_%d.csv" % n) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    main = [(s, e, k) for s, e, k in rows if MAIN.search(k)]
    k3 = [(s, e, k) for s, e, k in rows if K3.search(k)]
    alg = 8.0 * (n * n + n + n)
    persistent = len(main) < ITERS // 2
    if persistent:
        dur = sum(e - s for s, e, _ in main) / 1e3 / ITERS
        period, k3us, gaps = dur, 0.0, 0.0
    else:
        body = main[5:]                                   # the first launches: the clocks are still settling
        dur = statistics.mean(e - s for s, e, _ in body) / 1e3
        k3us = statistics.mean(e - s for s, e, _ in k3[5:]) / 1e3 if len(k3) > 5 else 0.0
        starts = [s for s, _, _ in body]
        period = statistics.median(b - a for a, b in zip(starts, starts[1:])) / 1e3
        gaps = period - dur - k3us
    f, nf = pmc(n, "FETCH_SIZE", "FETCH_SIZE")
    w, _ = pmc(n, "WRITE_SIZE", "WRITE_SIZE")
    h, _ = pmc(n, "TCC_HIT_sum", "TCC_HIT_sum")
    m, _ = pmc(n, "TCC_HIT_sum", "TCC_MISS_sum")
    per = ITERS if persistent else 1
    traffic = (f * 1024 * 2 + w * 1024) / per
    print("| %d | `%s` | %.2f | %.2f | %.2f | %.2f | %.3f | %.3f | %.4f | %.3f | %d |" % (
        n, short(main[len(main) // 2][2]), dur, k3us, period, gaps, alg / (dur * 1e-6) / PEAK, alg / (period * 1e-6) / PEAK,
        traffic / alg, h / (h + m) if h + m > 0 else float("nan"), len(main)))

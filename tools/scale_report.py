#!/usr/bin/env python3
"""Read bench.py lines of a 1 -> N GPU sweep (files with one JSON line each, or a SCALE_rNN.json-like file holding a list /
dict of them) and say where the time of an iteration went on every size: value and speed-up over the 1-GPU line, K1 per rank
(median, fraction of the 8 TB/s roofline), the update kernel per rank (on several GPUs its duration holds the wait for the
peers = the cost of the exchange), what is left (kernel boundaries, launch latency), transport and its calibration.
    python tools/scale_report.py line_n1.json line_n2.json line_n4.json line_n8.json      (dev tool)"""
import json, sys


def lines_of(path):
    txt = open(path).read().strip()
    out = []
    try:
        doc = json.loads(txt)
        stack = [doc]
        while stack:
            o = stack.pop()
            if isinstance(o, dict) and "metric" in o and "n_gpus" in o:
                out.append(o)
            elif isinstance(o, dict):
                stack.extend(o.values())
            elif isinstance(o, list):
                stack.extend(o)
    except ValueError:
        for l in txt.splitlines():
            if l.startswith("{"):
                try:
                    d = json.loads(l)
                    if "metric" in d:
                        out.append(d)
                except ValueError:
                    pass
    return out


def main(paths):
    rows = sorted((d for p in paths for d in lines_of(p)), key=lambda d: d["n_gpus"])
    if not rows:
        print("no bench lines found")
        return 1
    base = next((d for d in rows if d["n_gpus"] == 1 and d.get("value")), None)
    for d in rows:
        n, v = d["n_gpus"], d.get("value")
        if not v:
            print("N=%d: no value (%s)" % (n, (d.get("error") or {}).get("message", "?")[:120]))
            continue
        c, rf = d["config"], d["roofline"]
        ms = d["ms_per_step"]
        k1 = [q["median_ms"] for q in d.get("k1_per_rank", [])] or [rf.get("median_launch_ms") or 0.0]
        upd = [q["median_ms"] for q in (d.get("update_kernel") or {}).get("per_rank", []) if q["launches_timed"]]
        line = "N=%d: %.1f it/s, %.4f ms/step" % (n, v, ms)
        if base:
            line += ", speed-up %.2fx (efficiency %.1f %%)" % (v / base["value"], 100.0 * v / base["value"] / n)
        print(line)
        print("     transport %s  calibration %s  rccl_nranks %s  distinct_gpus %s" % (
            c.get("transport"), c.get("transport_calibration_ms_per_iteration"), c.get("rccl_nranks"), c.get("distinct_gpus")))
        print("     K1 median per rank: min %.4f / max %.4f ms (roofline frac of the slowest %.3f)" % (min(k1), max(k1), rf.get("frac") or 0.0))
        if upd:
            rest = ms - max(k1) - max(upd)
            print("     update kernel (%s): min %.4f / max %.4f ms;  step - K1 - update = %.4f ms (boundaries, launch latency, skew)" % (
                (d["update_kernel"]["kernel"].split(" ")[0]), min(upd), max(upd), rest))
        else:
            print("     step - K1 = %.4f ms (K3, boundaries)" % (ms - max(k1)))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

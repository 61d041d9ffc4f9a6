#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (gpurun_out/<round>_{stats,fetch,write,sq,tcc}) into the small
summaries committed under profiles/.  Usage: python profiles/summarize.py r01

Commands that produced the inputs (on the MI355X box, `cd /tmp && export TMPDIR=/tmp`):
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01_stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc FETCH_SIZE  ... -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline   (own pass)
  rocprofv3 --kernel-trace --pmc WRITE_SIZE  ... (own pass)
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE ...
  rocprofv3 --kernel-trace --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum ...
HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of wide (16 B/lane) coalesced reads, so both the raw and the doubled figure are kept.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, suffix):
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", d, "*", "*" + suffix))
    return max(fs, key=os.path.getmtime) if fs else None      # gpurun merges runs into the same directory: take the latest


def short(name):
    name = name.replace("void ", "").replace("ibgs::", "")
    return name.split("(")[0][:60]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    # which bench workload the passes ran (bench.py's tag: "<config>[ geo][ forward-only] opacity=<init|trained>") and which kernel
    # sources: bench.py drops these numbers as soon as either differs from what it is timing
    workload = sys.argv[2] if len(sys.argv) > 2 else "C3 opacity=init"
    sys.path.insert(0, ROOT)
    from ibgs_amd._build import csrc_sha, tu_shas
    import datetime
    # stamps: the whole source tree (csrc_sha) and every translation unit on its own (tu_shas): bench.py quotes a kernel's counters while the
    # unit that holds the kernel is unchanged
    out = {"tag": tag, "workload": workload, "csrc_sha": csrc_sha(), "tu_shas": tu_shas(), "date": datetime.date.today().isoformat()}
    st = find(tag + "_stats", "kernel_stats.csv")
    if st:
        rows = list(csv.DictReader(open(st)))
        with open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline (23 steps + 3 stage-timing steps + 10 forward-only passes)\n")
            f.write("kernel,calls,total_ms,avg_us,min_us,max_us,pct\n")
            for r in rows[:24]:
                f.write("%s,%s,%.3f,%.2f,%.2f,%.2f,%s\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                          float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
        for r in rows:
            if "render_bwd" in r["Name"]:
                out["render_bwd_avg_us_rocprof"] = float(r["AverageNs"]) / 1e3
            if "render_fwd" in r["Name"]:
                out["render_fwd_avg_us_rocprof"] = float(r["AverageNs"]) / 1e3
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ("_fetch", "_write", "_sq", "_tcc", "_mfma"):
        cc = find(tag + d, "counter_collection.csv")
        if not cc:
            continue
        for r in csv.DictReader(open(cc)):
            if "ibgs::" not in r["Kernel_Name"]:
                continue          # every kernel of the library (the step's torch kernels -- three adds of the loss -- are not the rasterizer's)
            per_kernel[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in per_kernel.items()}      # mean per launch
    out["per_launch_counters"] = counters
    # HBM traffic of ONE STEP as measured: sum over the library's kernels of (2 x FETCH_SIZE + WRITE_SIZE) x launches per step.  The PMC passes
    # run full steps and forward-only passes: a forward kernel's launches are counted per forward (= launches of the preprocess kernel),
    # a backward / loss kernel's per backward (= launches of preprocess_bwd)
    launches = {k: len(v.get("FETCH_SIZE", [])) for k, v in per_kernel.items()}
    n_fwd = max([n for k, n in launches.items() if "preprocess_kernel" in k] or [0])
    n_bwd = max([n for k, n in launches.items() if "preprocess_bwd" in k] or [0])
    if n_fwd and n_bwd:
        total, table = 0.0, {}
        for k, c in counters.items():
            if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                continue
            bwd = any(s in k for s in ("render_bwd", "geo_window", "tile_order", "preprocess_bwd", "l1_", "det_"))
            per_step = launches[k] / float(n_bwd if bwd else n_fwd)
            b = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 * per_step
            table[k] = {"launches_per_step": round(per_step, 3), "bytes_per_step": b}
            total += b
        out["step_traffic_measured"] = {"bytes_per_step": total, "by_kernel": table,
                                        "how": "sum over ibgs:: kernels of (2 x FETCH_SIZE + WRITE_SIZE) KiB x launches per step (separate --pmc passes; MI355X_MICROARCH.md gfx950 correction)"}
    for k, c in counters.items():
        if "render_bwd" in k and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            fetch, write = c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
            out["render_bwd_fetch_bytes_raw"] = fetch
            out["render_bwd_write_bytes"] = write
            # records are gathered as 16-B-per-lane loads: apply the guide's x2 FETCH_SIZE correction
            out["render_bwd_bytes_per_launch"] = 2.0 * fetch + write
    json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_counters.json"), "w"), indent=1, sort_keys=True)
    if workload == "C3 opacity=init":      # what the default bench line quotes (matched on csrc_sha + workload)
        json.dump(out, open(os.path.join(ROOT, "profiles", "counters_latest.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in out.items() if k != "per_launch_counters"}, indent=1))
    for k, c in counters.items():
        print(k, {a: ("%.4g" % b) for a, b in c.items()})


if __name__ == "__main__":
    main()

#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of the preprocess stage's kernels for the current library and an older one, alternating, on one box
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-extras --no-geo-line --no-trained-geo-line"
other=${1:-ibgs_amd/_exp/libibgs_rast_r06b.so}
for rep in 1 2; do
  for lib in "" "$other"; do
    rm -rf gpurun_out/pre_stats
    IBGS_LIB="$lib" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pre_stats -- python3 bench.py --steps 20 --warmup 3 $B > gpurun_out/pre_stats.log 2>&1
    echo "== ${lib:-current}"
    python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/pre_stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(p in r["Name"] for p in ("sh_color", "preprocess_kernel", "rendered_note", "cell_place", "cell_setup", "cell_colscan")):
            print("   %-40s calls %4s avg %7.2f us min %7.2f max %7.2f" % (r["Name"].replace("ibgs::", "").replace("void ", "").split("(")[0][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  done
done

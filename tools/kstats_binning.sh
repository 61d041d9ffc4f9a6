cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-extras --no-geo-line --no-trained-geo-line"
PAT="cell_count|cell_place|cell_colscan|cell_setup|expand_count|cell_scan|tile_ranges|expand_scatter"
for wl in init trained; do
  if [ $wl = init ]; then A=""; else A="--opacity trained --cluster 0.3 --anisotropy plane --scale-sigma 1.0 --geo"; fi
  rm -rf gpurun_out/fe_stats
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fe_stats -- python3 bench.py --steps 10 --warmup 2 $B $A > gpurun_out/fe_stats.log 2>&1
  echo "== $wl"
  python3 - "$PAT" <<'PY'
import csv, glob, sys
pats = sys.argv[1].split("|")
for f in glob.glob("gpurun_out/fe_stats/*/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if any(p in r["Name"] for p in pats)]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows:
        print("%-40s calls %5s avg %8.2f us min %8.2f max %8.2f" % (r["Name"].replace("ibgs::", "").replace("void ", "").split("(")[0][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done

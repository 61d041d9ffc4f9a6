# usage (on the MI355X box): bash tools/pmc_tile_map.sh "<layouts>"  -> gpurun_out/tile_map_pmc.txt
# L2 <-> fabric traffic of the geo blend kernels (FETCH_SIZE, WRITE_SIZE, TCC hit/miss: separate passes) per workgroup -> tile layout.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/tile_map_pmc.txt
: > $out
for m in $1; do
  export IBGS_TILE_MAP_FWD=$m IBGS_TILE_MAP_BWD=$m IBGS_TILE_MAP_FWD_GEO=$m IBGS_TILE_MAP_BWD_GEO=$m
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    d=gpurun_out/tmpmc_${m}_$(echo $c | cut -d' ' -f1)
    rm -rf $d
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 bench.py --geo --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $d.log
  done
  python3 - $m >> $out <<'PY'
import csv, glob, collections, sys
m = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/tmpmc_%s_*/*/*counter_collection.csv" % m):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "render_" in k or "geo_window" in k:
            agg[k.split("(")[0].replace("ibgs::", "").replace("void ", "")[:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    a = {c: sum(x) / len(x) for c, x in v.items()}
    print("%-6s %-34s fetch(x2) %.0f MB  write %.0f MB  TCC hit %.2e miss %.2e" % (m, k, 2 * a.get("FETCH_SIZE", 0) / 1024, a.get("WRITE_SIZE", 0) / 1024, a.get("TCC_HIT_sum", 0), a.get("TCC_MISS_sum", 0)))
PY
done
cat $out

# usage: bash tools/pmc_quick.sh "<counters>" [bench args]   -- prints per-launch averages for the render kernels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcq
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d gpurun_out/pmcq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $2 > gpurun_out/pmcq.log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcq/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "render_" in k:
            agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, "  ".join("%s=%.4g" % (c, sum(xs)/len(xs)) for c, xs in sorted(v.items())))
PY

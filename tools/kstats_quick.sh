# usage: bash tools/kstats_quick.sh [bench args]   -- rocprofv3 kernel stats (average us per kernel) of a short bench run
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ksq
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ksq -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-geo-line --no-trained-geo-line $@ > gpurun_out/ksq.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/ksq/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:32]:
        print("%-70s calls %5s avg %9.2f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY

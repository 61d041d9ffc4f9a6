# usage: bash tools/kstats_test_frame.sh   -- rocprofv3 kernel stats of the reference's test-time frame (bench.py test_frame: 13 frames)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ktf
cat > /tmp/tf.py <<'PY'
import sys, torch
sys.argv = ["bench.py"]
sys.path.insert(0, ".")
import bench
from ibgs_amd import _lib, synthetic as syn
_lib.load()
print(bench.test_frame(torch.device("cuda", 0), syn.CONFIGS["C3"]))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ktf -- python3 /tmp/tf.py > gpurun_out/ktf.log 2>&1
tail -1 gpurun_out/ktf.log | cut -c1-200
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/ktf/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:24]:
        print("%-70s calls %5s avg %9.2f us  per frame %8.1f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 13e3, float(r["Percentage"])))
PY

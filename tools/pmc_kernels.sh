# usage: bash tools/pmc_kernels.sh "<kernel substrings, |-separated>" [bench args]  -- FETCH_SIZE / WRITE_SIZE / SQ counters (separate passes) of the named kernels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
pat="$1"; shift
n=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  n=$((n+1)); rm -rf gpurun_out/pk_$n
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pk_$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-geo-line --no-trained-geo-line "$@" > gpurun_out/pk_$n.log 2>&1
done
python3 - "$pat" <<'PY'
import csv, glob, collections, sys
pats = sys.argv[1].split("|")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pk_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(p in k for p in pats):
            agg[k.split("(")[0].replace("ibgs::", "").replace("void ", "")[:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    a = {c: sum(x) / len(x) for c, x in v.items()}
    print(k, " ".join("%s=%.4g" % (c, x) for c, x in sorted(a.items())), "| fetch x2 %.0f MB write %.0f MB" % (2 * a.get("FETCH_SIZE", 0) / 1024, a.get("WRITE_SIZE", 0) / 1024))
PY

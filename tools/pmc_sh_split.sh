# PMC counters of the SH kernels with the coefficients in one array / in the model's two arrays (tools/run_train_iter.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for mode in "" "--no-split-sh"; do
  n=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
    n=$((n+1)); rm -rf gpurun_out/ps_$n
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/ps_$n -- python3 tools/run_train_iter.py 4 $mode > gpurun_out/ps_$n.log 2>&1
  done
  echo "== ${mode:-split}"
  python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/ps_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sh_color" in k or "preprocess_bwd" in k:
            agg[k.split("(")[0].replace("ibgs::", "").replace("void ", "")[:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    a = {c: sum(x) / len(x) for c, x in v.items()}
    print(k, " ".join("%s=%.4g" % (c, x) for c, x in sorted(a.items())), "| fetch x2 %.0f MB write %.0f MB" % (2 * a.get("FETCH_SIZE", 0) / 1024, a.get("WRITE_SIZE", 0) / 1024))
PY
done

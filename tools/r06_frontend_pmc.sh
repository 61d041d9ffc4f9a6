#!/bin/bash
# round 6: the front end's kernels on the trainer-shaped workload beside the init scene -- rocprofv3 kernel stats and PMC passes (each counter group alone,
# --kernel-trace only).  usage (MI355X box): bash tools/r06_frontend_pmc.sh   -> gpurun_out/r06_front/{kstats,pmc}_{init,trained}.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_front; mkdir -p $out
B="--no-cpu-baseline --no-extras --no-geo-line --no-trained-geo-line"
PAT="cell_count|cell_place|cell_colscan|cell_setup|expand_count|cell_scan|tile_ranges|expand_scatter|preprocess_kernel|sh_color|onesweep|rendered_note"
for wl in init trained; do
  if [ $wl = init ]; then A=""; else A="--opacity trained --cluster 0.3 --anisotropy plane --scale-sigma 1.0 --geo"; fi
  rm -rf gpurun_out/fe_stats
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fe_stats -- python3 bench.py --steps 10 --warmup 2 $B $A > gpurun_out/fe_stats.log 2>&1
  python3 - "$PAT" > $out/kstats_$wl.txt <<'PY'
import csv, glob, sys
pats = sys.argv[1].split("|")
for f in glob.glob("gpurun_out/fe_stats/*/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if any(p in r["Name"] for p in pats)]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows:
        print("%-64s calls %5s avg %8.2f us min %8.2f max %8.2f" % (r["Name"].replace("ibgs::", "").replace("void ", "").split("(")[0][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  n=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum"; do
    n=$((n+1)); rm -rf gpurun_out/fe_pmc_$n
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/fe_pmc_$n -- python3 bench.py --steps 3 --warmup 1 $B $A > gpurun_out/fe_pmc_$n.log 2>&1
  done
  python3 - "$PAT" > $out/pmc_$wl.txt <<'PY'
import csv, glob, collections, sys
pats = sys.argv[1].split("|")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/fe_pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(p in k for p in pats):
            agg[k.split("(")[0].replace("ibgs::", "").replace("void ", "")[:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    a = {c: sum(x) / len(x) for c, x in v.items()}
    print(k, " ".join("%s=%.4g" % (c, x) for c, x in sorted(a.items())), "| fetch x2 %.1f MB write %.1f MB" % (2 * a.get("FETCH_SIZE", 0) / 1024, a.get("WRITE_SIZE", 0) / 1024))
PY
done
for wl in init trained; do echo "== $wl"; cat $out/kstats_$wl.txt; cat $out/pmc_$wl.txt; done

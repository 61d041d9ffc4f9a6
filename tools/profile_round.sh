# usage (on the MI355X box): bash tools/profile_round.sh r02 [extra bench args, e.g. --geo]  -> gpurun_out/<tag>_{stats,fetch,write,sq,tcc,mfma}
# Then, back in the repo: python profiles/summarize.py <tag> ["<workload tag>"]
#   (writes profiles/<tag>_kernel_stats.csv, <tag>_counters.json and -- for the default workload -- counters_latest.json, stamped
#    with the kernel-source fingerprint so that bench.py can tell fresh numbers from stale ones)
# PMC passes run alone (--kernel-trace only), one counter group per pass, as MI355X_MICROARCH.md "rocprofv3 PMC slots" prescribes.
tag=${1:-r02}
shift
extra="$@"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-extras --no-geo-line --no-trained-geo-line $extra"
rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_sq gpurun_out/${tag}_tcc gpurun_out/${tag}_mfma
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --steps 20 --warmup 3 $B > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python3 bench.py --steps 5 --warmup 2 $B > /dev/null 2> gpurun_out/${tag}_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python3 bench.py --steps 5 --warmup 2 $B > /dev/null 2> gpurun_out/${tag}_write.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_sq -- python3 bench.py --steps 5 --warmup 2 $B > /dev/null 2> gpurun_out/${tag}_sq.log
rocprofv3 --kernel-trace --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/${tag}_tcc -- python3 bench.py --steps 5 --warmup 2 $B > /dev/null 2> gpurun_out/${tag}_tcc.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA --output-format csv -d gpurun_out/${tag}_mfma -- python3 bench.py --steps 5 --warmup 2 $B > /dev/null 2> gpurun_out/${tag}_mfma.log
ls gpurun_out/${tag}_*/*/ | head -30

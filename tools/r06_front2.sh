#!/bin/bash
# round 6, second front-end pass (row walks of the preprocess kernel flattened over the wave; the R note carried by the SH kernel's first workgroup; cell_setup folded
# into the place kernel): full GPU suite, then A/B of the current build against ibgs_amd/_exp/libibgs_rast_r06b.so (= the commit before) on four workloads
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_front2; mkdir -p $out; rm -f $out/*.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.txt
L=ibgs_amd/_exp/libibgs_rast_r06b.so
bash tools/ab_lib.sh $L --opacity trained --cluster 0.3 --anisotropy plane --scale-sigma 1.0 --geo > $out/bench_ab.txt 2>&1
bash tools/ab_lib.sh $L >> $out/bench_ab.txt 2>&1
bash tools/ab_lib.sh $L --opacity trained --cluster 0.5 >> $out/bench_ab.txt 2>&1
bash tools/ab_lib.sh $L --config C3_720p --opacity trained >> $out/bench_ab.txt 2>&1
tail -4 $out/pytest_gpu.txt; cat $out/bench_ab.txt

#!/bin/bash
# round 6: front end on trainer-shaped scenes, A/B of the current build against ibgs_amd/_exp/libibgs_rast_r06a.so (one GPU call)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_front; mkdir -p $out; rm -f $out/*.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trained_scene.py tests/test_gpu_fullsize_geo.py tests/test_gpu_depth_batch.py tests/test_gpu_renderer.py -x -q > $out/pytest_subset.txt 2>&1; echo "pytest rc $?" >> $out/pytest_subset.txt
bash tools/ab_lib.sh ibgs_amd/_exp/libibgs_rast_r06a.so --opacity trained --cluster 0.3 --anisotropy plane --scale-sigma 1.0 --geo > $out/bench_ab.txt 2>&1
bash tools/ab_lib.sh ibgs_amd/_exp/libibgs_rast_r06a.so >> $out/bench_ab.txt 2>&1
bash tools/ab_lib.sh ibgs_amd/_exp/libibgs_rast_r06a.so --opacity trained --cluster 0.5 >> $out/bench_ab.txt 2>&1
tail -4 $out/pytest_subset.txt; cat $out/bench_ab.txt

#!/bin/bash
# round 6: near-singular conics in the blend backward, A/B (one GPU call).  Output -> gpurun_out/r06_ab/
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_ab2; mkdir -p $out; rm -f $out/*.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_anisotropic.py tests/test_gpu_deterministic.py tests/test_gpu_hybrid.py tests/test_gpu_balanced_order.py tests/test_gpu_rccl_world1.py tests/test_gpu_geo_no_window.py tests/test_gpu_fullsize_geo.py -x -q > $out/pytest_subset.txt 2>&1; echo "pytest rc $?" >> $out/pytest_subset.txt
timeout 900 python tools/ab_ref_arith.py all >> $out/ab.txt 2>&1
AB_REF_ARITH=1 timeout 900 python tools/ab_ref_arith.py parity >> $out/ab.txt 2>&1
AB_ATOMICS=1 AB_REPEAT=4 timeout 900 python tools/ab_ref_arith.py parity >> $out/ab.txt 2>&1
bash tools/ab_lib.sh ibgs_amd/_exp/libibgs_rast_r05.so > $out/bench_ab.txt 2>&1
bash tools/ab_lib.sh ibgs_amd/_exp/libibgs_rast_r05.so --opacity trained --cluster 0.3 --anisotropy plane --scale-sigma 1.0 >> $out/bench_ab.txt 2>&1
bash tools/ab_lib.sh ibgs_amd/_exp/libibgs_rast_r05.so --opacity trained --cluster 0.3 --anisotropy plane --scale-sigma 1.0 --geo >> $out/bench_ab.txt 2>&1
bash tools/ab_lib.sh ibgs_amd/_exp/libibgs_rast_r05.so --opacity trained --anisotropy needle >> $out/bench_ab.txt 2>&1
tail -15 $out/pytest_subset.txt; grep -v amdgpu.ids $out/ab.txt | cut -c1-330; cat $out/bench_ab.txt

#!/bin/bash
# round 6 closing measurements on ONE box: the -m gpu suite, the default bench line, the three profile rounds (C3 colour, C3 geo, trained geo), the train-iteration kernel tables
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_final
python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r06_final/pytest_gpu.txt
python bench.py > gpurun_out/r06_final/bench_default.json 2> gpurun_out/r06_final/bench_default.err
bash tools/profile_round.sh r06 > gpurun_out/r06_final/profile_r06.log 2>&1
bash tools/profile_round.sh r06geo --geo > gpurun_out/r06_final/profile_r06geo.log 2>&1
bash tools/profile_round.sh r06tg --opacity trained --cluster 0.3 --anisotropy plane --scale-sigma 1.0 --geo > gpurun_out/r06_final/profile_r06tg.log 2>&1
python tools/train_iter_profile.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_final/train_iter_kernels.txt
python tools/train_iter_profile.py full 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_final/train_iter_full_kernels.txt
python tools/train_iter_profile.py sh_factored 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_final/train_iter_sh_factored_kernels.txt
python tools/train_iter_profile.py full sh_factored 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_final/train_iter_full_sh_factored_kernels.txt
# the counter CSVs are large: keep what profiles/summarize.py reads
find gpurun_out -name "*kernel_trace.csv" -path "*r06*" -delete 2>/dev/null
du -sh gpurun_out | tail -1
cat gpurun_out/r06_final/pytest_gpu.txt; python -c "
import json; d=json.load(open('gpurun_out/r06_final/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('step_frac_measured')); print(d['trained_geo']['ms_per_step'] if 'trained_geo' in d else None)"
head -12 gpurun_out/r06_final/train_iter_full_kernels.txt

#!/bin/bash
# round 6: the SH coefficients straight from the factors -- kernel tables of the trainer-shaped iteration with and without (one box), and the bench's train_iter object
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_sh_factored; mkdir -p $O
python tools/train_iter_profile.py 2>&1 | grep -v amdgpu.ids > $O/warm_in_dense.txt
python tools/train_iter_profile.py sh_factored 2>&1 | grep -v amdgpu.ids > $O/warm_in_factored.txt
python tools/train_iter_profile.py full 2>&1 | grep -v amdgpu.ids > $O/full_dense.txt
python tools/train_iter_profile.py full sh_factored 2>&1 | grep -v amdgpu.ids > $O/full_factored.txt
for f in warm_in_dense warm_in_factored full_dense full_factored; do echo "== $f"; grep -E "^wall|adam|preprocess_bwd" $O/$f.txt | cut -c1-160; done

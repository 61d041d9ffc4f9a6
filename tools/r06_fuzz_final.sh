#!/bin/bash
# Fresh seeds on the FINAL library of round 6 (after the second front-end pass: flattened row walk in preprocess, note in the SH kernel, cell_setup folded).  MI355X.
mkdir -p gpurun_out/r06_fuzz_final
O=gpurun_out/r06_fuzz_final
timeout 600 python tools/fuzz_parity.py 120 9701 > $O/parity_9701.txt 2>&1
timeout 900 python tools/fuzz_parity.py 80 9702 - trained > $O/parity_9702_trained.txt 2>&1
timeout 600 python tools/fuzz_parity.py 60 9703 - big > $O/parity_9703_big.txt 2>&1
timeout 900 python tools/fuzz_fused.py 60 9704 > $O/fused_9704.txt 2>&1
timeout 600 python tools/fuzz_depth_batch.py 40 9705 > $O/depth_batch_9705.txt 2>&1
for f in $O/*.txt; do echo "== $f"; grep -E "FAIL|failures|worst|OUTSIDE" $f | tail -4; done

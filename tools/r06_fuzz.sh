#!/bin/bash
# The round-5 sweeps on the round-6 library, same seeds (VERDICT r5 item 1's "0 cases outside the bar"), plus fresh seeds.  MI355X; writes gpurun_out/r06_fuzz/*.txt
mkdir -p gpurun_out/r06_fuzz
O=gpurun_out/r06_fuzz
python -m pytest tests/test_gpu_fuzz_pins.py -x -q -m gpu -s 2>&1 | grep -E "fuzz pin|passed|failed|Error" > $O/pins.txt
FUZZ_DETERMINISTIC=1 timeout 1500 python tools/fuzz_parity.py 120 9103 - trained > $O/parity_9103_trained_det.txt 2>&1
timeout 1500 python tools/fuzz_parity.py 120 9103 - trained > $O/parity_9103_trained_atomics.txt 2>&1
timeout 1200 python tools/fuzz_fused.py 100 9105 > $O/fused_9105.txt 2>&1
timeout 900 python tools/fuzz_parity.py 80 9601 - trained > $O/parity_9601_trained.txt 2>&1
timeout 600 python tools/fuzz_parity.py 120 9602 > $O/parity_9602.txt 2>&1
timeout 600 python tools/fuzz_parity.py 60 9603 - big > $O/parity_9603_big.txt 2>&1
timeout 900 python tools/fuzz_fused.py 80 9604 > $O/fused_9604.txt 2>&1
for f in $O/*.txt; do echo "== $f"; grep -E "FAIL|failures|worst|outside|passed|failed" $f | tail -8; done

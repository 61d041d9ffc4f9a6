for cfg in "0 8" "768 7" "768 4" "1536 6" "1536 4" "3072 5" "3072 4" "0 8" "0 4"; do set -- $cfg; IBGS_BWD_PAD_LDS=$1 IBGS_BWD_SLOT_ROUNDS=$2 IBGS_BENCH_SKIP=torch_l1,abs,hint,hop python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-geo-line --no-trained-geo-line 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('pad $1 rounds $2', 'step', round(d['ms_per_step'],4), 'bwd kernel', round(d['roofline'].get('kernel_ms'),4))"; done

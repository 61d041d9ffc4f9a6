for mt in 4096 1000000; do for args in "" "--cluster 0.5" "--opacity trained" "--opacity trained --cluster 0.5"; do
IBGS_HYBRID_MAX_TILES=$mt IBGS_BENCH_SKIP=torch_l1,abs,hint,hop python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-geo-line --no-trained-geo-line --no-extras $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); st=d['stages_ms']
print('max_tiles $mt  $args  median %.3f  fwd %.3f  bwd %.3f  order %.3f' % (d['median_ms_hipevent'], st['render_fwd'], st['render_bwd'], st['tile_order']))"
done; done

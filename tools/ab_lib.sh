# A/B of two builds of libibgs_rast.so on ONE box (box-to-box differences are 2-3 %, which is what many changes are worth).
# usage: bash tools/ab_lib.sh <other .so (same ABI), e.g. ibgs_amd/_exp/libibgs_rast_r03.so> [bench args]   -> prints stage times of both, twice (A B A B)
cd $GRAFT_REPO_ROOT
other="$1"; shift
for rep in 1 2; do
  for lib in "" "$other"; do
    IBGS_LIB="$lib" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-geo-line --no-trained-geo-line "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-40s median %.4f wall %.4f R %d |' % ('${lib:-current}', d['median_ms_hipevent'], d['ms_per_step'], d['config']['num_rendered']), ' '.join('%s %.3f' % (k[:12], v) for k, v in d['stages_ms'].items() if v > 0))
"
  done
done

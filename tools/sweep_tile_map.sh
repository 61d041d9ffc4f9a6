# usage (on the MI355X box): bash tools/sweep_tile_map.sh [extra bench args]  -> gpurun_out/tile_map_sweep.txt
# Blend-kernel times (bench.py's hipEvent stage timers, colour line and geo line) for every workgroup -> tile layout of csrc/common.h.
cd $GRAFT_REPO_ROOT
out=gpurun_out/tile_map_sweep.txt
: > $out
for m in ${MAPS:-rr g2 g4 g8 g16 g64 b2x2 b4x2 b4x4 b8x4 b8x8 g1024}; do
  IBGS_TILE_MAP_FWD=$m IBGS_TILE_MAP_BWD=$m IBGS_TILE_MAP_FWD_GEO=$m IBGS_TILE_MAP_BWD_GEO=$m python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > gpurun_out/tm_$m.json 2> gpurun_out/tm_$m.err
  python3 - $m >> $out <<'PY'
import json, sys
m = sys.argv[1]
try:
    d = json.load(open("gpurun_out/tm_%s.json" % m))
    s, g = d["stages_ms"], d["geo"]["stages_ms"]
    print("%-6s colour: fwd %.3f bwd %.3f step %.3f | geo: fwd %.3f bwd %.3f step %.3f" % (m, s["render_fwd"], s["render_bwd"], d["ms_per_step"], g["render_fwd"], g["render_bwd"], d["geo"]["ms_per_step"]))
except Exception as e:
    print(m, "failed", e)
PY
done
cat $out

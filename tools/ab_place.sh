#!/bin/bash
# A/B of the look-back placement (IBGS_PLACE_LOOKBACK=1, default) against the four-kernel placement (=0): step time at C3 with initial and
# with trained-like opacities, and with half of the Gaussians in one blob.  Run on the GPU box; prints one line per run.
for args in "" "--opacity trained" "--cluster 0.5"; do
  for v in 0 1; do
    IBGS_PLACE_LOOKBACK=$v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10 $args 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lookback=$v', '$args', 'fps', round(d['value'],1), 'ms', round(d['ms_per_step'],4), {k: round(v,3) for k,v in (d.get('stages_ms') or {}).items()})"
  done
done

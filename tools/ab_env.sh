#!/bin/bash
# A/B of one environment switch over the bench workloads.  usage: tools/ab_env.sh VAR "v0 v1 ..." [extra bench args]
var=$1; vals=$2; shift 2
for args in "" "--opacity trained" "--cluster 0.5" "--cluster 0.5 --opacity trained"; do
  for v in $vals; do
    env $var=$v timeout 300 python bench.py --no-cpu-baseline --no-extras --no-geo-line --no-trained-geo-line --steps 40 --warmup 10 $args "$@" 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d.get('stages_ms') or {}; print('$var=$v', '$args', 'ms', round(d['ms_per_step'],4), 'render_bwd', round(s.get('render_bwd',0),4), 'render_fwd', round(s.get('render_fwd',0),4))"
  done
done

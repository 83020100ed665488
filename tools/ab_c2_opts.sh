#!/bin/bash
# usage: tools/ab_c2_opts.sh "<bench args A>" "<bench args B>" ...   -- C2 with the in-tree library under different bench.py
# arguments (e.g. "--opt 9=0" against ""), alternating, three rounds: points/s, layer-kernel milliseconds, step time.
for round in 1 2 3; do
  for args in "$@"; do
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-voigt --no-extras $args 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stages_ms']
print('[$args]', round(d['value'], 1), 'full', round(s['full_layers_ms'], 2), 'red', round(s['reduced_layers_ms'], 2), 'surf', round(s['surface_ms'], 2), 'step', round(d['ms_per_step'], 2))"
  done
done

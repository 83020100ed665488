#!/bin/bash
# usage: tools/ab_c2_libs_q4.sh libA.so libB.so ...  -- C2 with each library, alternating, three rounds (like ab_c2.sh) with the reduced launch printed
for round in 1 2 3; do
  for lib in "$@"; do
    MOM_LIBRARY=$PWD/$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-voigt --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stages_ms']
print('$lib', round(d['value'], 1), 'full', round(s['full_layers_ms'], 2), 'red', round(s['reduced_layers_ms'], 2), 'step', round(d['ms_per_step'], 2))"
  done
done

#!/bin/bash
# usage: tools/ab.sh libA.so libB.so ...  -- the full bench with each library build, on one GPU box (same-box A/B)
for lib in "$@"; do
  MOM_LIBRARY=$PWD/$lib python bench.py --steps 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('$lib', round(d['value'],1), 'full', round(s['full_layers_ms'],2), 'red', round(s['reduced_layers_ms'],2))"
done

#!/bin/bash
# bench.py at several overlap depths (sampling launches in flight), one summary line each.  Run on the GPU box.
for d in 1 2 3 2 1; do
  python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --overlap $d --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('overlap $d', round(d['value'], 1), 'jets/s', round(d['ms_per_step'], 2), 'ms/step  sample_ms', round(d['sample_ms'], 2), 'train_ms', round(d['train_ms'], 2), 'frac', round(d['roofline']['frac'], 3))"
done

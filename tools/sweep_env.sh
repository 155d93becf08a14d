#!/bin/bash
# tools/sweep_env.sh VAR v1 v2 ... -- bench.py once per value of an environment tunable (dev experiments)
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout -k 10 200 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('$VAR=$v', 'Mrays/s', d['value'], 'ms', d['ms_per_step'], c['frame_ms_split_rank0'], 'nodes', c['bvh_nodes'], 'depth', c['bvh_depth'], 'box/ray', r['per_ray']['box_tests'], 'tri/ray', r['per_ray']['tri_tests'])"
done

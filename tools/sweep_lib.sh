#!/bin/bash
# tools/sweep_lib.sh name1 name2 ... -- bench.py once per A/B library in build_variants/ (dev experiments)
for n in "$@"; do
  CRT_LIB=$PWD/build_variants/libcrt_$n.so timeout -k 10 200 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('$n', 'Mrays/s', d['value'], 'ms', d['ms_per_step'], c['frame_ms_split_rank0'])"
done

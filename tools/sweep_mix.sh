#!/bin/bash
# dev experiment runner: lines of "ENV=VAL,... libname"
while read -r envs lib; do
  [ -z "$lib" ] && continue
  env $(echo $envs | tr ',' ' ') CRT_LIB=$PWD/build_variants/libcrt_$lib.so timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$envs $lib', 'Mrays/s', d['value'], 'ms', d['ms_per_step'], c['frame_ms_split_rank0'])"
done

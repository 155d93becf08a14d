#!/bin/bash
# dev experiment: k_shade's grid (CRT_SHADE_BLOCKS) against the frame's shade_resolve milliseconds
for b in 4096 1280 2560 3840 5120 6400 8192 16384; do
  CRT_SHADE_BLOCKS=$b timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 10 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print($b, b['value'], b['ms_per_step'], b['config']['frame_ms_split_rank0']['shade_resolve'])" || exit 1
done

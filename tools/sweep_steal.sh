for s in "16,4" "8,4" "24,4" "32,4" "16,2" "16,8" "16,1" "48,4"; do
  CRT_STEAL=$s timeout -k 10 200 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$s', d['value'], d['ms_per_step'])"
done

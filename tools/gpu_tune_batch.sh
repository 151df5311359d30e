#!/bin/bash
for cfg in "512 0.5 1024" "512 0.25 1024" "768 1 768" "768 0.5 768" "384 1 1280" "1024 0.5 512"; do
  set -- $cfg; ts=$1; band=$2; bt=$3
  b=$(python -c "print(0.115*($ts**0.5)*$band)")
  MNAV_TILE_SIZE=$ts MNAV_TILE_BAND=$b python bench.py --steps 2 --warmup 1 --no-cpu --no-latency --batch $bt 2>/dev/null | python -c "
import sys,json
l=[x for x in sys.stdin if x.startswith('{')][-1]; b=json.loads(l); print('tile',$ts,'bandmult',$band,'batch',$bt,'plans/s %.0f ms/step %.1f frac %.4f'%(b['value'],b['ms_per_step'],b['roofline']['frac']))"
done

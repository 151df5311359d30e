#!/bin/bash
# scratch: throughput of the persistent engine over tile size / batch size
for cfg in "512 1024" "512 1280" "512 2560" "448 1536" "448 3072" "416 1536" "384 1792"; do
  set -- $cfg; ts=$1; bt=$2
  MNAV_TILE_SIZE=$ts timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu --no-latency --batch $bt 2>/dev/null | python -c "
import sys,json
l=[x for x in sys.stdin if x.startswith('{')][-1]; b=json.loads(l); print('tile',$ts,'batch',$bt,'plans/s %.0f ms/step %.1f frac %.4f'%(b['value'],b['ms_per_step'],b['roofline']['frac']))"
done

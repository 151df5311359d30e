#!/bin/bash
# scratch: batch throughput vs tile size / band / blocks
for ts in 256 512 1024; do for band in 0.25 0.5 1 2; do
  b=$(python -c "print(0.115*($ts**0.5)*$band)")
  MNAV_TILE_SIZE=$ts MNAV_TILE_BAND=$b python bench.py --steps 2 --warmup 1 --no-cpu --no-latency 2>/dev/null | python -c "
import sys,json
l=[x for x in sys.stdin if x.startswith('{')][-1]; b=json.loads(l); print('tile',$ts,'bandmult',$band,'plans/s %.0f ms/step %.1f launches %.0f avg_us %.1f'%(b['value'],b['ms_per_step'],b['roofline']['launches_per_step'],b['roofline']['avg_launch_us']))"
done; done

#!/bin/bash
# scratch: single-plan latency (k_tile_round) over tile size / band multiplier
for cfg in "512 4" "1024 4" "1024 2" "2048 4" "2048 2" "2048 1"; do
  set -- $cfg; ts=$1; bm=$2
  MNAV_ROUNDS_BAND_MULT=$bm MNAV_TILE_SIZE=$ts timeout 120 python bench.py --steps 1 --warmup 1 --no-cpu --batch 128 2>/dev/null | python -c "
import sys,json
l=[x for x in sys.stdin if x.startswith('{')][-1]; b=json.loads(l); print('tile',$ts,'bandmult',$bm,'single ms %.2f p95 %.2f'%(b['ms_per_makeplan_single'],b['ms_per_makeplan_single_p95']))"
done

#!/bin/bash
# scratch: throughput of the persistent engine over tile size / batch size
for cfg in "512 1280" "640 1024" "768 768" "768 1536" "1024 512" "1024 1024"; do
  set -- $cfg; ts=$1; bt=$2
  MNAV_VERBOSE=1 MNAV_TILE_SIZE=$ts timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu --no-latency --batch $bt 2>gpurun_out/tune_err.log | python -c "
import sys,json
l=[x for x in sys.stdin if x.startswith('{')][-1]; b=json.loads(l); print('tile',$ts,'batch',$bt,'plans/s %.0f ms/step %.1f kernel_us %.0f'%(b['value'],b['ms_per_step'],b['roofline']['avg_launch_us']))"
  grep "mnav" gpurun_out/tune_err.log | head -1
done

#!/bin/bash
# scratch: throughput of the persistent engine over batch size (tail effect)
for bt in 1280 2560 3840 5120; do
  timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu --no-latency --batch $bt 2>/dev/null | python -c "
import sys,json
l=[x for x in sys.stdin if x.startswith('{')][-1]; b=json.loads(l); print('batch',$bt,'plans/s %.0f ms/step %.1f kernel_us %.0f'%(b['value'],b['ms_per_step'],b['roofline']['avg_launch_us']))"
done

"""The C4 leg of bench.py (10M-vertex terrain: single plans + one warm-up and one timed batch on the tile-batch engine) on its own,
for the profiler: python tools/gpu_c4_batch.py [batch=4096] [grid=3163]"""
import json
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3163
a = types.SimpleNamespace(c4_grid=n, c4_batch=b, offset=0.3, no_cpu=True)
r = bench.run_leg(lambda: bench.leg_c4(0, a))
print(json.dumps(r))

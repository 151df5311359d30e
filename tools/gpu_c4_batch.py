import sys, json, types
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
import bench
for b in (1536,):
    a=types.SimpleNamespace(c4_grid=3163,c4_batch=b,offset=0.3,no_cpu=True)
    r=bench.run_leg(lambda: bench.leg_c4(0,a))
    print(json.dumps({k:v for k,v in r.items() if k!='workload'})[:900])

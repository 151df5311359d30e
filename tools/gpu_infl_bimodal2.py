"""The C3 leg of bench.py on its own, twice (is the 85 ms inflation wave of the driver's line a property of the leg, or of what ran before it?)"""
import json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
a = types.SimpleNamespace(grid=1000, offset=0.3, no_cpu=True)
for k in range(2):
    r = bench.run_leg(lambda: bench.leg_c3(0, a))
    print(json.dumps({"run": k, "spec": r.get("cost_stack_as_specified", {}).get("inflation_wave"), "used": r.get("cost_stack_used", {}).get("inflation_wave"),
                      "err": r.get("error")}), flush=True)

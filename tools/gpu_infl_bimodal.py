"""The inflation wave of bench.py's C3 cost stack, repeated: does its time fall into two modes (6 ms / 85 ms)?
One JSON line: per repeat the counters of stack(0.3) and stack(0.6).  Run under `rocprofv3 --kernel-trace` to see which launches differ.

    python tools/gpu_infl_bimodal.py [repeats=4] [fresh_context_per_repeat=0] [warm=0] [idle_seconds=0]
warm=1 runs a C2-sized Dijkstra batch in the process first (what the driver's full bench.py does before its C3 leg)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mesh_navigation_amd import capi, meshgen  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    fresh = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    warm = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    idle_s = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0          # host-side pause before every wave (bench.py analyses the first stack on the CPU for seconds)
    N = 1000
    if warm:
        m2 = meshgen.terrain(N, 0.1, 2)
        c2 = capi.MnavContext(0)
        c2.upload_mesh(m2.xyz, m2.faces, m2.edges, None)
        c2.upload_costs(np.zeros(m2.V, np.float32), meshgen.edge_lengths(m2))
        g = np.random.default_rng(5).choice(m2.V, size=2048, replace=False).astype(np.uint32)
        c2.plan_dijkstra_batch(g, np.full(2048, m2.vertex_at(0.9, 0.9), np.uint32), want_fields=False, path_cap=16384, want_stats=False)
        c2.close()
    mesh = meshgen.terrain(N, 0.1, 3)
    vnrm, _ = bench.vertex_normals(mesh)
    out = []
    ctx = None
    for r in range(reps):
        if ctx is None or fresh:
            if ctx is not None:
                ctx.close()
            ctx = capi.MnavContext(0)
            ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, vnrm)
        row = {}
        for thr in (0.3, 0.6):
            ctx.layer_steepness(0, thr)
            if idle_s > 0:
                import time
                time.sleep(idle_s)
            row[str(thr)] = ctx.layer_inflation(1, 0)
        out.append(row)
        print("repeat", r, {k: (round(v["ms_wave"], 2), v["steps"], v["evals"]) for k, v in row.items()}, file=sys.stderr)
    print(json.dumps(dict(repeats=reps, fresh=fresh, warm=warm, runs=out)))


if __name__ == "__main__":
    main()

"""One configuration of tools/gpu_infl_fuzz.py with the library's trace on stderr (MNAV_TRACE=1): python tools/gpu_infl_one.py i [i ...]"""
import os, sys
os.environ.setdefault("MNAV_TRACE", "1")
sys.path.insert(0, ".")
import numpy as np
from mesh_navigation_amd import capi
from oracle import oracle as O
from tests.test_gpu_layers import _sparse_lethal_case
for a in sys.argv[1:]:
    i = int(a)
    case, lethal, inv, radius = _sparse_lethal_case(i)
    m = case.mesh
    cfg = O.InflationCfg.defaults(); cfg.inflation_radius = radius
    cost, dist, vec = case.om.inflation(lethal, case.edge_dist, cfg, invalid=inv)
    ctx = capi.MnavContext(0)
    ctx.upload_mesh(m.xyz, m.faces, m.edges, case.vn)
    ctx.layer_upload(0, np.zeros(m.V, np.float32), lethal)
    print(f"== configuration {i}: V {m.V} radius {radius} invalid {inv is not None}", file=sys.stderr, flush=True)
    try:
        st = ctx.layer_inflation(1, 0, inflation_radius=radius, invalid=inv)
        c, _, d = ctx.layer_download(1, distances=True)
        print(i, "ok", st, "bits equal", bool(np.array_equal(np.asarray(d).view(np.uint32), dist.view(np.uint32))), flush=True)
    except RuntimeError as e:
        print(i, "FAILED", str(e)[:120], flush=True)
    ctx.close()

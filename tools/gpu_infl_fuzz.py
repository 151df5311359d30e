"""Deterministic fuzz of the inflation wave on the GPU box (round 5): configuration i -> mesh, lethal set (steep faces / scattered edges /
ISOLATED vertices), radius, invalid vertices; device layer against the oracle bit for bit.  Found the tied pop times of DESIGN.md 3.2
(isolated lethal vertices: configurations 16, 40, 48, 62, 67, 84, 129 ... ran into the step cap; 62, 67, 84 still do).
    python tools/gpu_infl_fuzz.py [first index] [seconds]  ->  one JSON line"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
torch.cuda.init()
from mesh_navigation_amd import capi, meshgen
from oracle import oracle as O
from tests.common import Case
def cfg_of(i):
    rng = np.random.default_rng(5000 + i)
    N = int(rng.choice([64, 128, 200])); seed = int(rng.integers(1000)); amp = float(rng.choice([0.3, 0.8]))
    kind = int(rng.integers(3)); a = float(rng.choice([0.3, 0.5])); b = int(rng.choice([100, 400])); c = int(rng.choice([30, 300]))
    use_inv = bool(rng.random() < 0.5); radius = float(rng.choice([0.25, 0.4, 0.9, 1.3]))
    return rng, N, seed, amp, kind, a, b, c, use_inv, radius
def build(i):
    rng, N, seed, amp, kind, a, b, c, use_inv, radius = cfg_of(i)
    case = Case(meshgen.terrain(N, 0.1, seed, amplitude=amp)); m = case.mesh
    lethal = np.zeros(m.V, np.uint8)
    if kind == 0: _, lethal = case.om.steepness(case.vn, a)
    elif kind == 1:
        lethal[m.edges[rng.choice(m.E, max(1, m.E // b), replace=False)].ravel()] = 1
        lethal[rng.choice(m.V, max(1, m.V // 80), replace=False)] = 1
    else: lethal[rng.choice(m.V, max(1, m.V // c), replace=False)] = 1
    inv = None
    if use_inv:
        inv = np.zeros(m.V, np.uint8); inv[rng.choice(m.V, m.V // 40, replace=False)] = 1
    return case, lethal, inv, radius, dict(i=i, N=N, seed=seed, amp=amp, kind=kind, radius=radius, inv=use_inv)
if __name__ == "__main__":
    t0 = time.perf_counter(); i0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0; budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
    i = i0; bad = []
    while time.perf_counter() - t0 < budget:
        case, lethal, inv, radius, desc = build(i); m = case.mesh
        cfg = O.InflationCfg.defaults(); cfg.inflation_radius = radius
        cost, dist, vec = case.om.inflation(lethal, case.edge_dist, cfg, invalid=inv)
        ctx = capi.MnavContext(0)
        ctx.upload_mesh(m.xyz, m.faces, m.edges, case.vn)
        ctx.layer_upload(0, np.zeros(m.V, np.float32), lethal)
        try:
            st = ctx.layer_inflation(1, 0, inflation_radius=radius, invalid=inv)
            c, _, d = ctx.layer_download(1, distances=True)
            if not (np.array_equal(np.asarray(d).view(np.uint32), dist.view(np.uint32)) and np.array_equal(np.asarray(c).view(np.uint32), cost.view(np.uint32))):
                bad.append(dict(desc, what="bits differ"))
        except RuntimeError as e:
            bad.append(dict(desc, what=str(e)[:80]))
        ctx.close(); i += 1
    print(json.dumps(dict(runs=i - i0, bad=bad)))

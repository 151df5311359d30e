"""CPU: the PROTOCOL of the asynchronous tile engine (mesh_navigation_amd/csrc/mnav_async.h: ticket queue of woken tiles / state
word per tile / the count of filed-and-not-retired tickets that ends a plan) on oracle/async_model.cpp -- the kernel's shared-memory
operations restated one by one, run by several virtual workgroups that are interleaved pseudo-randomly (seeded, reproducible) at
every single operation, with long random stalls -- against the sequential oracle (dijkstra_mesh_planner.cpp:287-348).

What the engine promises (like every engine: the finalize pass and the lazy path walk rely on it): when a plan is declared finished,
every vertex the reference pops -- dist <= dist[target] + max(offset, 0) -- holds the reference's float32 potential bit for bit.
The model also checks, while it runs, that a plan is finished exactly once and only when none of its tiles is pending, queued or in
solve, that a tile never has two solvers and never two live tickets."""
from __future__ import annotations

import numpy as np
import pytest

from mesh_navigation_amd import meshgen
from oracle import oracle as O
from tests.common import Case, terrain_case


def run(case: Case, seeds, targets, offset=0.3, cost_limit=1.0, **kw):
    m = case.mesh
    r = O.async_tile_model(m.xyz, m.faces, m.edges, case.weights, case.costs, seeds, targets, offset=offset, cost_limit=cost_limit,
                           invalid=case.invalid, **kw)
    info = {k: v for k, v in r.items() if k != "dist"}
    assert r["code"] == 0 and r["abort"] == 0, info
    assert r["violations"] == 0, info
    assert r["finishes"] == len(seeds), info
    for k, (s, t) in enumerate(zip(seeds, targets)):
        full = case.om.dijkstra(case.weights, case.costs, int(s), int(t), goal_dist_offset=np.inf, cost_limit=cost_limit, invalid=case.invalid).dist
        dt = full[t]
        bound = np.float32(np.float64(dt) + max(offset, 0.0)) if np.isfinite(dt) else np.float32(np.inf)
        popped = full <= bound
        got = r["dist"][k]
        assert np.array_equal(got[popped].view(np.uint32), full[popped].view(np.uint32)), (k, info)
        assert (got >= full).all()                                   # every value is the length of a real path
    return r


@pytest.mark.parametrize("workgroups", [1, 3, 8])
def test_terrain_offsets_bands_and_workgroup_counts(workgroups):
    case = terrain_case(40, 5)
    m = case.mesh
    rng = np.random.default_rng(workgroups)
    for k, (offset, band, tile) in enumerate(((0.3, 0.0, 64), (0.0, 0.05, 32), (np.inf, "tile", 64), (-0.5, 0.0, 32))):   # band 0: local fixed point per solve (the default)
        n = 1 + k % 3
        st = rng.choice(m.V, 2 * n, replace=False)                    # (a plan whose seed is its target never reaches an engine: dijkstra :252-255)
        seeds, targets = st[:n], st[n:]
        r = run(case, seeds, targets, offset=offset, band=band, tile=tile, workgroups=workgroups, sched_seed=10 * workgroups + k)
        assert r["max_concurrent_solves"] <= workgroups
    # a batch larger than the workgroups: they move on to the plans that are left
    st = rng.choice(m.V, 12, replace=False)
    seeds, targets = st[:6], st[6:]
    run(case, seeds, targets, tile=64, workgroups=workgroups, sched_seed=99)


def test_cost_limit_invalid_and_unreachable_targets():
    mesh = meshgen.terrain(36, 0.1, 11)
    rng = np.random.default_rng(5)
    costs = rng.uniform(0.0, 1.4, mesh.V).astype(np.float32)
    inv = (rng.uniform(size=mesh.V) < 0.05).astype(np.uint8)
    case = Case(mesh, costs, edge_cost_factor=1.0, invalid=inv)
    ok = np.flatnonzero((inv == 0) & (costs <= 0.8))
    st = rng.choice(ok, 6, replace=False)
    run(case, st[:3], st[3:], cost_limit=0.8, tile=32, workgroups=4, sched_seed=3)
    # a wall of over-limit vertices: the target is never reached, the plan still ends (its component is swept, then nothing is pending)
    N = 36
    costs2 = np.zeros(mesh.V, np.float32)
    costs2[np.arange(N) * N + N // 2] = 5.0
    case2 = Case(mesh, costs2, edge_cost_factor=0.0)
    s, t = mesh.vertex_at(0.2, 0.5), mesh.vertex_at(0.8, 0.5)
    r = run(case2, [s], [t], tile=32, workgroups=5, sched_seed=4)
    assert not np.isfinite(r["dist"][0][t])
    # holes, several components, a face-less target
    p = meshgen.punched(40, 0.1, 4, drop=0.15)
    casep = Case(p)
    deg = np.bincount(p.edges.ravel(), minlength=p.V)
    okp = np.flatnonzero(deg > 0)
    st = rng.choice(okp, 6, replace=False)
    run(casep, st[:3], st[3:], tile=32, workgroups=4, sched_seed=5)


def test_many_interleavings_on_a_mesh_of_seven_tiles():
    """Few tiles: the count of live tickets is 1 or 2 most of the time, which is where a premature `finished` or a lost wake-up
    would show; 40 schedules, half of them with a band far narrower than a tile (every tile is solved many times and wakes ITSELF:
    the solver's look at its own wake-up value after clearing the state word)."""
    case = terrain_case(14, 5)
    m = case.mesh
    rng = np.random.default_rng(0)
    tickets = 0
    for seed in range(40):
        s, t = (int(x) for x in rng.choice(m.V, 2, replace=False))
        r = run(case, [s], [t], offset=[np.inf, 0.0, 0.3, -1.0][seed % 4], tile=32, band=0.05 if seed % 2 else 0.0, workgroups=2 + seed % 5,
                sched_seed=seed + 1)
        tickets += r["tickets"]
        assert r["tickets"] >= r["activations"] + r["drops"]
        assert (r["epochs"] > 0) == bool(seed % 2)                    # band advances only with a band
    assert tickets > 40 * 7


def test_a_full_ring_gives_up_instead_of_hanging():
    case = terrain_case(14, 5)
    m = case.mesh
    r = O.async_tile_model(m.xyz, m.faces, m.edges, case.weights, case.costs, [97], [5], offset=np.inf, tile=32, band=0.05, workgroups=3,
                           sched_seed=2, ring_cap=6)
    assert r["code"] == 1 and r["abort"] == 5                         # the product re-runs such a call on the tile rounds


def test_the_model_sees_protocol_errors():
    """Deliberately broken variants of the protocol are caught by the model's checks (some schedule of a small sweep shows each of
    them; the intact protocol passes all of those schedules): that is what makes a green run mean something."""
    case = terrain_case(14, 5)
    m = case.mesh
    kw = dict(offset=np.inf, tile=32, band=0.5, workgroups=4, budget=3_000_000)   # (a band a few tiles wide: several tiles in flight at once)
    seeds = range(1, 25)
    def bad(mutate):
        return [O.async_tile_model(m.xyz, m.faces, m.edges, case.weights, case.costs, [97], [5], sched_seed=s, mutate=mutate, **kw) for s in seeds]
    # 3: wakers file a ticket whatever the state word says -- two live tickets / two solvers of one tile
    assert any(r["violations"] > 0 for r in bad(3))
    # 2: the solver does not look at its wake-up value again after clearing the state word -- a wake-up that arrived during the solve
    # is lost: "finished" with a tile still pending (or the potential is wrong)
    assert any(r["violations"] > 0 for r in bad(2))
    good = bad(0)
    assert all(r["violations"] == 0 and r["finishes"] == 1 and r["abort"] == 0 for r in good)

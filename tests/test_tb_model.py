"""CPU: the tile-batch SSSP engine's streams and schedule (mesh_navigation_amd/csrc/mnav_tb_build.h, interpreted by
oracle/tb_model.cpp exactly as k_tb_solve_q reads them) against the sequential oracle (dijkstra_mesh_planner.cpp:287-348).

What the engine promises (and the lazy path walk / the finalize pass rely on): every vertex the reference POPS --
dist <= goal_dist = dist[target] + goal_dist_offset -- holds the reference's float32 potential bit for bit."""
from __future__ import annotations

import numpy as np
import pytest

from mesh_navigation_amd import meshgen
from oracle import oracle as O
from tests.common import Case, terrain_case


def check(case: Case, seeds, targets, offset=0.3, cost_limit=1.0, tile=128, jacobi=1, band=None):
    r = O.tile_batch_model(case.mesh.xyz, case.mesh.faces, case.mesh.edges, case.weights, case.costs, seeds, targets, offset=offset,
                           cost_limit=cost_limit, tile=tile, band=band, jacobi=jacobi, invalid=case.invalid)
    assert r["code"] == 0
    for k, (s, t) in enumerate(zip(seeds, targets)):
        ref = case.om.dijkstra(case.weights, case.costs, int(s), int(t), goal_dist_offset=np.inf, cost_limit=cost_limit, invalid=case.invalid)
        full = ref.dist                                              # full-field potential of the reference loop
        dt = full[t]
        goal_dist = np.float32(np.float64(dt) + offset) if np.isfinite(dt) else np.float32(np.inf)
        popped = full <= goal_dist
        got = r["dist"][k]
        assert np.array_equal(got[popped].view(np.uint32), full[popped].view(np.uint32)), (k, int(popped.sum()))
        assert got[t].view(np.uint32) == full[t].view(np.uint32)
        # nothing below the reference's potential anywhere (every value is the length of a real path)
        assert (got >= full).all()
    return r


@pytest.mark.parametrize("tile", [64, 128])
@pytest.mark.parametrize("jacobi", [0, 1])
def test_c1_terrain_batch(tile, jacobi):
    case = terrain_case(224, 1)
    rng = np.random.default_rng(3)
    seeds = rng.choice(case.mesh.V, 6, replace=False)
    targets = np.full(6, case.mesh.vertex_at(0.9, 0.9))
    r = check(case, seeds, targets, tile=tile, jacobi=jacobi)
    assert r["max_sweeps"] <= 12                                    # the four diagonal orders converge in a handful of sweeps


def test_offsets_and_bands():
    case = terrain_case(96, 7)
    seeds = [case.mesh.vertex_at(0.2, 0.2), case.mesh.vertex_at(0.5, 0.1)]
    targets = [case.mesh.vertex_at(0.8, 0.7)] * 2
    for offset in (0.0, 0.3, np.inf):
        for band in (0.05, None, 50.0):
            check(case, seeds, targets, offset=offset, band=band, tile=64)


def test_costs_cost_limit_invalid_and_unreachable():
    mesh = meshgen.terrain(80, 0.1, 11)
    rng = np.random.default_rng(5)
    costs = rng.uniform(0.0, 1.4, mesh.V).astype(np.float32)
    inv = (rng.uniform(size=mesh.V) < 0.05).astype(np.uint8)
    case = Case(mesh, costs, edge_cost_factor=1.0, invalid=inv)
    ok = np.flatnonzero((inv == 0) & (costs <= 0.8))
    seeds = rng.choice(ok, 5, replace=False)
    targets = rng.choice(ok, 5, replace=False)
    check(case, seeds, targets, cost_limit=0.8, tile=64)
    # a wall of over-limit vertices: the target is unreachable, the potential stays +inf there
    costs2 = np.zeros(mesh.V, np.float32)
    N = 80
    costs2[np.arange(N) * N + N // 2] = 5.0
    case2 = Case(mesh, costs2, edge_cost_factor=0.0)
    s, t = mesh.vertex_at(0.2, 0.5), mesh.vertex_at(0.8, 0.5)
    r = check(case2, [s], [t], cost_limit=1.0, tile=64)
    assert not np.isfinite(r["dist"][0][t])


def test_punched_and_fan_meshes():
    m = meshgen.punched(72, 0.1, 4, drop=0.15)
    case = Case(m)
    deg = np.bincount(m.edges.ravel(), minlength=m.V)
    ok = np.flatnonzero(deg > 0)
    rng = np.random.default_rng(2)
    check(case, rng.choice(ok, 4, replace=False), rng.choice(ok, 4, replace=False), tile=64)
    f = meshgen.fan_field(spokes=40, rings=6, seed=1)               # a valence-40 hub: continuation blocks, many ghosts
    casef = Case(f)
    for tile in (64, 128):                                          # the hub's rows need continuation blocks (7 sources per block)
        r = check(casef, [1, f.V - 1], [f.V - 2, 0], tile=tile)
        assert r["max_sweeps"] <= 40


def test_a_quarter_that_reruns_its_last_chunk_changes_nothing():
    """k_tb_solve_q runs the four quarters of a wave to the LONGEST of their sweep streams: a quarter past the end of its own stream
    re-runs its last chunk.  A relaxation applied again changes nothing: same bits, never more sweeps.  (The sweep chunks are
    stored transposed for the DPP reads of the kernel -- mnav_tb_build.h tb_sweep_index --; the model reads them through the same
    index function, so every test of this file also checks that layout.)"""
    case = terrain_case(128, 1)
    m = case.mesh
    rng = np.random.default_rng(3)
    seeds = rng.choice(m.V, 6, replace=False)
    targets = np.full(6, m.vertex_at(0.9, 0.9))
    kw = dict(tile=120, jacobi=1)
    plain = O.tile_batch_model(m.xyz, m.faces, m.edges, case.weights, case.costs, seeds, targets, **kw)
    rerun = O.tile_batch_model(m.xyz, m.faces, m.edges, case.weights, case.costs, seeds, targets, rerun_last_chunk=2, **kw)
    assert plain["code"] == 0 and rerun["code"] == 0 and rerun["sweeps"] <= plain["sweeps"]
    assert np.array_equal(rerun["dist"].view(np.uint32), plain["dist"].view(np.uint32))
    f = meshgen.fan_field(spokes=40, rings=6, seed=1)               # a valence-40 hub: continuation blocks of one row
    casef = Case(f)
    for tile in (64, 128):
        a = O.tile_batch_model(f.xyz, f.faces, f.edges, casef.weights, casef.costs, [1, f.V - 1], [f.V - 2, 0], tile=tile)
        c = O.tile_batch_model(f.xyz, f.faces, f.edges, casef.weights, casef.costs, [1, f.V - 1], [f.V - 2, 0], tile=tile, rerun_last_chunk=3)
        assert c["code"] == 0 and c["sweeps"] <= a["sweeps"]
        assert np.array_equal(a["dist"].view(np.uint32), c["dist"].view(np.uint32))



def test_v_layout_streams_of_the_register_resident_kernel():
    """k_tbv_solve (mnav_tbv.h) keeps a tile's distances in VGPRs and reads its sweeps from the V layout of the streams (blocks of
    up to six sources, rows as ready-made register-index words, no forwarding rule): the model interprets that layout (jacobi bit
    1) on the same schedule -- same potential bit for bit, never more sweeps than the Q layout with its reserved forwarding slot;
    on the terrain, with costs / a cost limit / invalid vertices, on the punched mesh and on the valence-40 fan (continuation
    blocks of one target)."""
    case = terrain_case(224, 1)
    rng = np.random.default_rng(3)
    seeds = rng.choice(case.mesh.V, 6, replace=False)
    targets = np.full(6, case.mesh.vertex_at(0.9, 0.9))
    for jac in (2, 3):
        rv = check(case, seeds, targets, tile=120, jacobi=jac)
        rq = check(case, seeds, targets, tile=120, jacobi=jac & 1)
        assert np.array_equal(rv["dist"].view(np.uint32), rq["dist"].view(np.uint32))
        assert rv["max_sweeps"] <= 12 and rv["sweeps"] <= rq["sweeps"] * 1.05
    mesh = meshgen.terrain(80, 0.1, 11)
    rng = np.random.default_rng(5)
    costs = rng.uniform(0.0, 1.4, mesh.V).astype(np.float32)
    inv = (rng.uniform(size=mesh.V) < 0.05).astype(np.uint8)
    case2 = Case(mesh, costs, edge_cost_factor=1.0, invalid=inv)
    ok = np.flatnonzero((inv == 0) & (costs <= 0.8))
    check(case2, rng.choice(ok, 5, replace=False), rng.choice(ok, 5, replace=False), cost_limit=0.8, tile=120, jacobi=3)
    m = meshgen.punched(72, 0.1, 4, drop=0.15)
    deg = np.bincount(m.edges.ravel(), minlength=m.V)
    okp = np.flatnonzero(deg > 0)
    check(Case(m), rng.choice(okp, 4, replace=False), rng.choice(okp, 4, replace=False), tile=120, jacobi=3)
    f = meshgen.fan_field(spokes=40, rings=6, seed=1)
    r = check(Case(f), [1, f.V - 1], [f.V - 2, 0], tile=120, jacobi=3)
    assert r["max_sweeps"] <= 40


def test_pair_division_by_multiply_high():
    """k_tb_scan splits a listed pair (tile * blocks + block) by one multiply-high and a shift (mnav::tb_div_magic): exact for
    every numerator below 2^31 -- the host refuses a flag matrix beyond that -- and every number of blocks a batch can have."""
    import ctypes as C
    f = O.model_lib().tbm_div_magic_check
    f.restype = C.c_ulonglong
    f.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    for d in list(range(1, 1100)) + [4095, 4096, 4097, 65535, 65536]:
        assert f(d, 0, 1 << 16, 1, 0) == 0, d
        assert f(d, (1 << 31) - (1 << 16), 1 << 31, 1, 0) == 0, d
        assert f(d, 0, 1 << 31, 104729, 0) == 0, d
    for d in (2, 3, 7, 64, 112, 113, 127, 128, 1000, 1024):
        assert f(d, (1 << 31) - (1 << 22), 1 << 31, 1 << 30, 1) == 0, d

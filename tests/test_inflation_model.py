"""The device's inflation wave (mnav_layer_inflation: InflationLayer::waveCostInflation, inflation_layer.cpp:341-491,
replayed on the ordered-wave engine) checked WITHOUT a GPU: the CPU model runs the very same rule code
(mesh_navigation_amd/csrc/mnav_eval.h: eval_cvp with Plan.seed_mask, infl_sethian, infl_candidate) and must reproduce
the sequential oracle -- which tests/test_ref_pins_oracle.py pins to the reference's own InflationLayer -- bit for
bit, under every evaluation order the model can emulate (list order, reversed, shuffled, Jacobi)."""
import numpy as np
import pytest

from mesh_navigation_amd import meshgen
from oracle import oracle as O
from tests.common import Case


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def run(case, lethal, radius, invalid=None, orders=(0, 1, 2, 3), delta=None):
    cfg = O.InflationCfg.defaults()
    cfg.inflation_radius = radius
    _, dist, _ = case.om.inflation(lethal, case.edge_dist, cfg, invalid=invalid)
    m = case.mesh
    out = []
    for order in orders:
        r = O.schedule_model_inflation(m.faces, m.edges, case.edge_dist, lethal, radius, delta=delta, order=order,
                                       invalid=invalid, max_steps=20000)
        assert r["code"] == 0 and r["verify_bad"] == 0 and r["verify_flags"] == 0, (order, r["verify_bad"])
        assert np.array_equal(bits(r["dist"]), bits(dist)), (order, int((bits(r["dist"]) != bits(dist)).sum()))
        out.append(r)
    return dist, out


@pytest.mark.parametrize("N,seed", [(40, 3), (72, 7)])
def test_steepness_lethals_default_radius(N, seed):
    """The C3 stack at small size: lethal = too steep, InflationLayer defaults (radius 0.4)."""
    case = Case(meshgen.terrain(N, 0.1, seed))
    _, lethal = case.om.steepness(case.vn, 0.3)
    dist, out = run(case, lethal, 0.4)
    assert np.isfinite(dist).sum() > lethal.sum()              # the wave did inflate something
    assert all(r["bands"] <= 3 for r in out)                    # one band per radius: the wave dies out within ~2


def test_sparse_sources_wide_radius_and_invalid_vertices():
    """Lethal EDGES scattered over the mesh and a radius of 13 edge lengths: waves from different sources meet and wrap
    around each other (values set below the popping value -> cascades); invalid vertices are popped but never fixed
    (:417-422).  Isolated lethal vertices never start a wave (no face ever has two fixed corners)."""
    rng = np.random.default_rng(0)
    case = Case(meshgen.terrain(48, 0.1, 2))
    m = case.mesh
    lethal = np.zeros(m.V, np.uint8)
    lethal[m.edges[rng.choice(m.E, m.E // 200, replace=False)].ravel()] = 1
    invalid = np.zeros(m.V, np.uint8)
    invalid[rng.choice(m.V, m.V // 40, replace=False)] = 1
    run(case, lethal, 1.3, orders=(0, 2, 3))
    run(case, lethal, 1.3, invalid=invalid, orders=(0, 2))
    run(case, lethal, 1.3, orders=(2,), delta=0.2)             # narrower bands: same result
    iso = np.zeros(m.V, np.uint8)
    iso[[m.vertex_at(0.3, 0.3), m.vertex_at(0.7, 0.6)]] = 1
    dist, _ = run(case, iso, 0.4, orders=(0,))
    assert np.isfinite(dist).sum() == 2


def test_deep_cascades_are_healed_by_the_verification_sweeps():
    """Randomly scattered lethal vertices (many isolated, a few adjacent pairs): the few waves that do start reach
    the isolated zero-distance vertices late and fill the region around them BACKWARDS, several cascade levels deep.
    A member of such a cascade whose far ancestor moved is not re-queued by the work-list iteration; the fixing
    verification sweeps (verify_entry) must find and repair it -- the result is still the reference's."""
    rng = np.random.default_rng(0)
    healed = 0
    for s, trial in ((2, 2), (5, 5)):                           # found by fuzzing: stale deep-cascade keys without the sweeps
        case = Case(meshgen.terrain(48, 0.1, s))
        m = case.mesh
        rng = np.random.default_rng(1000 * s + trial)
        lethal = np.zeros(m.V, np.uint8)
        lethal[rng.choice(m.V, m.V // 30, replace=False)] = 1
        _, out = run(case, lethal, 1.3, orders=(0, 2))
        healed += sum(r["verify_sweeps"] for r in out)
    assert healed >= 0


def test_reference_gtest_vector_through_the_product_rule():
    """mesh_layers/test/inflation_layer_test.cpp:38-84 (test_wave_front_update), the reference's own known answer, on
    the PRODUCT's update arithmetic: triangle (0,0,0), (0.5,0,0), (0,0.5,0); distances {v0: 0, v1: |v0 v1|}; the update
    of v2 from (v0, v1) must give EXPECT_FLOAT_EQ 0.5 and return true (max_dist 5).  Then the rule must agree bit for
    bit with the oracle's restatement on random triangles."""
    e01, e02, e12 = np.float32(0.5), np.float32(0.5), np.float32(np.sqrt(np.float32(0.5)))
    v, requeue = O.product_inflation_update(0.0, float(e01), float(e12), float(e02), float(e01), 5.0)   # a=|v1v2| b=|v0v2| c=|v0v1|
    assert requeue and abs(v - 0.5) <= 4 * np.spacing(np.float32(0.5))                  # EXPECT_FLOAT_EQ = 4 ulp
    assert O.inflation_fading(_gtest_cfg(), v) == pytest.approx(0.9, rel=1e-6)          # :82-84 on the oracle's fading
    rng = np.random.default_rng(4)
    n_fin = 0
    for _ in range(4000):
        p = rng.uniform(-1, 1, (3, 2)).astype(np.float32)
        a = np.float32(np.linalg.norm(p[1] - p[2])); b = np.float32(np.linalg.norm(p[0] - p[2])); c = np.float32(np.linalg.norm(p[0] - p[1]))
        u1 = np.float32(rng.uniform(0, 1)); u2 = np.float32(u1 + rng.uniform(-1, 1) * c)
        dot = np.float32((a * a + b * b - c * c) / (2 * a * b))
        want = O.lib().mo_inflation_sethian(u1, u2, a, b, dot, np.float32(1.0))
        got, _ = O.product_inflation_update(u1, u2, a, b, c, 0.4)
        if np.isfinite(want):
            n_fin += 1
            assert np.float32(got).view(np.uint32) == np.float32(want).view(np.uint32)
        else:
            assert np.isnan(got)
    assert n_fin > 1000


def _gtest_cfg():
    cfg = O.InflationCfg.defaults()
    cfg.inflation_radius, cfg.inscribed_radius, cfg.lethal_value, cfg.inscribed_value, cfg.cost_scaling_factor = 1.5, 0.5, 1.0, 0.9, 1.0
    return cfg


def same_vectors(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)).all(axis=1) | (np.isnan(a).any(axis=1) & np.isnan(b).any(axis=1))


@pytest.mark.parametrize("kind", ["terrain", "punched"])
def test_repulsive_vector_field_is_the_reference_restatement_bit_for_bit(kind):
    """vector_map_ (inflation_layer.cpp:277-309): the order-dependent accumulation over the lethal contours (every face
    with two lethal corners is visited four times, in the order of the lethal corners' pops and of the walk around each)
    and the assignments from the supports of each vertex's last lowering update -- computed from the converged wave
    (mnav_eval.h infl_accumulate / infl_assign, what k_infl_accum / k_infl_assign run) against the sequential oracle.
    The punched mesh has boundary vertices and vertices whose faces form several fans (the walk crosses the gaps)."""
    rng = np.random.default_rng(1)
    for s in range(2):
        mesh = meshgen.terrain(40, 0.1, s) if kind == "terrain" else meshgen.punched(40, 0.1, s, drop=0.15)
        case = Case(mesh)
        for trial in range(2):
            lethal = np.zeros(mesh.V, np.uint8)
            if trial == 0:
                lethal[mesh.edges[rng.choice(mesh.E, mesh.E // 100, replace=False)].ravel()] = 1
            else:
                _, lethal = case.om.steepness(case.vn, 0.5)
                lethal[rng.choice(mesh.V, mesh.V // 60, replace=False)] = 1
            invalid = np.zeros(mesh.V, np.uint8)
            invalid[rng.choice(mesh.V, mesh.V // 30, replace=False)] = 1
            for radius, inv in ((0.4, None), (1.0, invalid)):
                cfg = O.InflationCfg.defaults()
                cfg.inflation_radius = radius
                _, dist, vec = case.om.inflation(lethal, case.edge_dist, cfg, invalid=inv)
                r = O.schedule_model_inflation(mesh.faces, mesh.edges, case.edge_dist, lethal, radius, order=2, invalid=inv,
                                               max_steps=20000, xyz=mesh.xyz)
                assert r["code"] == 0 and np.array_equal(bits(r["dist"]), bits(dist))
                ok = same_vectors(r["vec"], vec)
                assert ok.all(), (kind, s, trial, radius, int((~ok).sum()))
                wave = int(np.isfinite(dist).sum()) > int(lethal.sum())     # scattered single lethal vertices start no wave
                assert (r["has_vec"].sum() > 0) == wave


@pytest.mark.parametrize("i,order", [(84, 3), (62, 4), (476, 6)])
def test_a_band_that_no_order_of_evaluation_settles_is_popped_one_vertex_at_a_time(i, order):
    """Configurations 84 (Jacobi order), 62 and 476 (a seeded mixture of snapshot and in-place reads, model orders >= 4: closer to
    the device's racy evaluation than pure Jacobi) of tools/gpu_infl_fuzz.py: a narrow band holds a cascade whose members support
    each other with provisional keys -- it re-hangs itself under every order of evaluation, a sequential pass from a clean state
    included (what rounds 5-6 tried first: 2-3 %, then 0.5 % of random sparse-lethal maps refused on the device).  The controller then hands the band to the exact band routine (mnav_eval.h exact_*: the
    reference's own procedure, one pop at a time, a vertex supports others only once its state is final), the steps resume from a
    fixed point of their own rule.  The wave is the reference's bit for bit, and the verification sweep finds nothing."""
    from tests.test_gpu_layers import _sparse_lethal_case
    case, lethal, inv, radius = _sparse_lethal_case(i)
    cfg = O.InflationCfg.defaults()
    cfg.inflation_radius = radius
    _, dist, _ = case.om.inflation(lethal, case.edge_dist, cfg, invalid=inv)
    m = case.mesh
    r = O.schedule_model_inflation(m.faces, m.edges, case.edge_dist, lethal, radius, order=order, invalid=inv, max_steps=20000)
    assert r["code"] == 0 and r["verify_bad"] == 0 and r["verify_flags"] == 0
    assert np.array_equal(bits(r["dist"]), bits(dist))

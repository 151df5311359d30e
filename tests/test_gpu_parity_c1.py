"""GPU parity, BASELINE config C1 (50k-vertex terrain): HIP path through the C ABI vs the CPU oracle."""
import numpy as np
import pytest

from tests.common import terrain_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c1(gpu_ctx_factory):
    case = terrain_case(224, 1)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    return case, ctx


def test_dijkstra_c1_bit_exact(c1):
    case, ctx = c1
    m = case.mesh
    seed, target = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
    ref = case.om.dijkstra(case.weights, case.costs, seed, target)
    out = ctx.plan_dijkstra(seed, target, want_fields=True, want_vecmap=True)
    assert out.code == ref.code == 0
    assert np.array_equal(out.dist.view(np.uint32), ref.dist.view(np.uint32)), "potential must be bit-exact"
    assert np.array_equal(out.pred, ref.pred)
    assert np.array_equal(out.path, ref.path), "vertex-index path must be identical"
    assert out.stats["goal_dist"] == ref.stats["goal_dist"]
    vm = case.om.dijkstra_vector_map(ref.pred)
    assert np.array_equal(out.vecmap.view(np.uint32), vm.view(np.uint32))


def test_cvp_c1(c1):
    case, ctx = c1
    m = case.mesh
    seed, target = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
    sp = m.xyz[seed] + np.array([0.03, 0.02, 0.0], np.float32)
    tp = m.xyz[target] + np.array([0.03, 0.02, 0.0], np.float32)
    sf, _ = case.om.containing_face(sp)
    tf, _ = case.om.containing_face(tp)
    ref = case.om.cvp(case.weights, case.costs, case.vn, sp, sf, tf)
    out = ctx.plan_cvp(sp, sf, tf)
    assert out.code == ref.code == 0
    fin = np.isfinite(ref.dist)
    assert np.array_equal(np.isfinite(out.dist), fin)
    rel = np.abs(out.dist[fin] - ref.dist[fin]) / np.maximum(ref.dist[fin], 1e-12)
    assert rel.max() <= 1e-5, f"CVP potential tolerance 1e-5 relative (north_star); got {rel.max()}"
    assert (out.pred != ref.pred).mean() < 1e-4

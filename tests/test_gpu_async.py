"""GPU parity of the asynchronous tile engine (engine 'async', mesh_navigation_amd/csrc/mnav_async.h) -- OPT-IN.

The engine was written after round 4's GPU minutes were spent: its protocol is checked on the CPU model
(tests/test_async_model.py), its kernel has never run on hardware.  Until it has (tools/gpu_async_engine.py), these tests only run
with MNAV_TEST_ASYNC=1, 'auto' never selects the engine, and no number in DESIGN.md comes from it."""
import os

import numpy as np
import pytest

from tests.common import terrain_case
from tests.test_gpu_planners import assert_dijkstra_equal

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.environ.get("MNAV_TEST_ASYNC"), reason="opt-in: MNAV_TEST_ASYNC=1")]


def test_async_engine_matches_the_oracle(gpu_ctx_factory):
    case = terrain_case(128, 3)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    ctx.set_dijkstra_engine("async")
    m = case.mesh
    rng = np.random.default_rng(2)
    robot = m.vertex_at(0.85, 0.8)
    goals = rng.choice(m.V, 24, replace=False).astype(np.uint32)
    goals = goals[goals != robot]
    for off in (0.3, 0.0, float("inf"), -0.2):
        for g in goals[:2]:
            assert_dijkstra_equal(ctx.plan_dijkstra(int(g), robot, goal_dist_offset=off),
                                  case.om.dijkstra(case.weights, case.costs, int(g), robot, goal_dist_offset=off))
    targets = np.full(goals.shape[0], robot, np.uint32)
    refs = [case.om.dijkstra(case.weights, case.costs, int(g), robot) for g in goals]
    for fields in (True, False):
        b = ctx.plan_dijkstra_batch(goals, targets, want_fields=fields)
        for k, ref in enumerate(refs):
            assert b["codes"][k] == ref.code
            assert np.array_equal(b["paths"][k], ref.path), (fields, k)
            if fields:
                assert np.array_equal(b["dist"][k].view(np.uint32), ref.dist.view(np.uint32)), k
                assert np.array_equal(b["pred"][k], ref.pred), k
    ctx.set_dijkstra_engine("auto")

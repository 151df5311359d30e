"""ONE Dijkstra plan on a mesh that is range-partitioned over several GPUs (BASELINE config 4).

One process per GPU.  Every process holds the whole (read-only) mesh description but OWNS a contiguous range of
the Morton-ordered LDS tiles and, with them, their vertices: only the owner relaxes into a vertex.  The loop is

    repeat:  R local tile rounds on the own tiles                      (k_tile_round, mnav_shard_rounds[_async])
             ONE min-allreduce of the interface buffer                 (RCCL over xGMI: halo-vertex distances
                                                                        + the robot vertex, a few 10^4 floats)
             ghost values that dropped wake the tiles around them      (mnav_shard_apply[_async])
             ONE 3-float min-allreduce: smallest pending wake-up, dist[robot], -status
    until nothing that may still propagate is pending anywhere

With an engine that offers the asynchronous steps (GpuShardEngine) the loop stays on the device: kernels and collectives
are ordered by stream events, the termination words are written by a kernel, and the host reads them back only once
every `check_every` exchanges -- an exchange after convergence changes nothing (values only ever decrease, and the finalize
pass derives everything beyond goal_dist from the popped vertices), so looking late costs a few idle exchanges, not
correctness.
    finalize the own tiles (cut-off semantics + predecessors), min-allreduce dist / pred once

The schedule is label-correcting, so the potential is the unique fixed point of the reference's float32
relaxation (dijkstra_mesh_planner.cpp:331): bit-identical to the single-GPU plan and to the reference.  Messages
are latency bound (tens of microseconds each, xGMI bandwidth is irrelevant at these sizes); expect poor strong
scaling for one plan -- what this mode buys is a mesh that no longer has to fit one GPU's caches.

`run_sharded_plan` is written against a small engine protocol so that the same loop drives the GPU engine
(`GpuShardEngine`, the C ABI mnav_shard_*) with torch.distributed, several engines inside one process
(`plan_virtual_ranks`: one GPU standing in for several, used by the GPU tests) and the CPU model of the tests.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Sequence

import numpy as np

SUCCESS, CANCELED, NO_PATH_FOUND, INTERNAL_ERROR = 0, 51, 54, 60


@dataclass
class ShardedResult:
    code: int
    dist: np.ndarray | None       # V float32 (gathered)
    pred: np.ndarray | None       # V uint32 (gathered)
    path: np.ndarray              # dijkstra() list order: seed first ... pred[target]
    exchanges: int
    rounds: int


def walk_path(pred: np.ndarray, seed: int, target: int) -> tuple[int, np.ndarray]:
    """dijkstra_mesh_planner.cpp:358-373 on the gathered predecessor array."""
    if int(pred[target]) == target:
        return NO_PATH_FOUND, np.zeros(0, np.uint32)
    out = []
    v = target
    while v != seed and len(out) <= pred.shape[0]:
        v = int(pred[v])
        out.append(v)
    return SUCCESS, np.asarray(out[::-1], np.uint32)


def run_sharded_plan(engine, allreduce_min: Callable, seed: int, target: int, goal_dist_offset: float = 0.3,
                     rounds_per_exchange: int = 8, max_exchanges: int = 100_000, gather: bool = True, check_every: int = 8,
                     device_loop: bool | None = None) -> ShardedResult:
    """The loop above for ONE rank.  `engine`: begin/rounds/apply/finalize (see GpuShardEngine), optionally
    rounds_async/apply_async/read_control for the device-resident loop.
    `allreduce_min(x)`: in-place elementwise MIN over all ranks of a buffer the engine handed out (a torch tensor
    or a numpy array, the engine decides) -- the only communication there is."""
    engine.begin(seed, target, goal_dist_offset)
    exchanges = rounds = 0
    # 3 floats: [smallest pending wake-up, dist[target], -status].  The status of every rank (0 ok, 1 cancelled, 2 error) rides
    # on the termination reduce, so that a rank-local cancel or failure ends the plan on ALL ranks in the same exchange
    # instead of leaving the others blocked in the next collective.
    ctl = engine.control_buffer()
    if device_loop is None:
        device_loop = hasattr(engine, "apply_async")

    def failed(st: int) -> ShardedResult:
        if hasattr(engine, "drain"):
            engine.drain()                                            # kernels / collectives queued behind the failure finish before the next plan starts
        return ShardedResult(CANCELED if st == 1 else INTERNAL_ERROR, None, None, np.zeros(0, np.uint32), exchanges, rounds)

    while True:
        if device_loop:
            for _ in range(max(1, check_every)):                      # nothing in here waits on the host
                buf = engine.rounds_async(rounds_per_exchange)
                rounds += rounds_per_exchange
                allreduce_min(buf)
                engine.apply_async(buf, ctl)
                allreduce_min(ctl)
                exchanges += 1
            gmin, gtarget, st = engine.read_control(ctl)              # ONE read-back per block of exchanges
        else:
            buf = engine.rounds(rounds_per_exchange)
            rounds += rounds_per_exchange
            allreduce_min(buf)                                        # halo-vertex distances (+ robot vertex)
            local_min, target_dist = engine.apply(buf)
            ctl[0] = local_min
            ctl[1] = target_dist
            ctl[2] = -float(getattr(engine, "status", 0))
            allreduce_min(ctl)                                        # termination + status: 12 bytes
            exchanges += 1
            gmin, gtarget, st = float(ctl[0]), float(ctl[1]), int(round(-float(ctl[2])))
        if st:
            return failed(st)
        if not np.isfinite(gmin) or gmin > np.float32(np.float64(np.float32(gtarget)) + goal_dist_offset):
            break                                                     # nothing left that may still propagate
        if exchanges >= max_exchanges:
            raise RuntimeError("sharded plan did not terminate")      # (the count is the same on every rank)

    def agreed_status() -> int:
        return int(round(-float(ctl[2])))

    dist_buf, pred_buf = engine.finalize()                            # a failing fixed-point check sets engine.status, it does not raise
    ctl[2] = -float(getattr(engine, "status", 0))
    allreduce_min(ctl)
    if agreed_status():
        return failed(agreed_status())
    if not gather:
        return ShardedResult(SUCCESS, None, None, np.zeros(0, np.uint32), exchanges, rounds)
    allreduce_min(dist_buf)                                           # every vertex has exactly one owner
    allreduce_min(pred_buf)
    dist = engine.to_numpy(dist_buf).view(np.float32)
    pred = engine.to_numpy(pred_buf).view(np.uint32)
    code, path = walk_path(pred, seed, target)
    return ShardedResult(code, dist, pred, path, exchanges, rounds)


class GpuShardEngine:
    """The C ABI mnav_shard_* on one GPU; exchange buffers are torch CUDA tensors (RCCL reduces them in place)."""

    def __init__(self, ctx, rank: int, world: int, cost_limit: float = 1.0, device=None):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.cost_limit = cost_limit
        self.n = ctx.shard_setup(rank, world)
        self.info = ctx.shard_info()
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.buf = torch.empty(self.n, dtype=torch.float32, device=dev)
        self.ctl = torch.zeros(3, dtype=torch.float32, device=dev)
        self.status = 0                                                # 0 ok, 1 cancelled, 2 error: agreed on by all ranks in run_sharded_plan
        self.dist = torch.empty(ctx.V, dtype=torch.float32, device=dev)
        # predecessors travel as int32 bit patterns (RCCL has no uint32 MIN in torch): ids < 2^31 keep their order,
        # and the neutral element 0xFFFFFFFF is -1, which MIN would prefer -> flip the sign bit around the collective
        self.pred = torch.empty(ctx.V, dtype=torch.int32, device=dev)

    def begin(self, seed, target, offset):
        self.status = 0
        self.ctx.shard_begin(seed, target, offset, self.cost_limit)

    def control_buffer(self):
        return self.ctl

    def rounds(self, r):
        self.torch.cuda.synchronize()
        try:
            if self.ctx.shard_rounds(r, self.buf.data_ptr()) == 1:     # mnav_cancel arrived on this rank
                self.status = max(self.status, 1)
        except RuntimeError:
            self.status = 2
        return self.buf

    def apply(self, buf):
        self.torch.cuda.synchronize()
        try:
            return self.ctx.shard_apply(buf.data_ptr())
        except RuntimeError:                                           # rank-local failure: reported through the status word, the
            self.status = 2                                            # other ranks must not be left waiting in the next all-reduce
            return float("inf"), float("inf")

    def drain(self):
        self.torch.cuda.synchronize()

    # -- the device-resident loop: kernels linked to torch's current stream by events, no host synchronisation
    def _stream(self) -> int:
        return int(self.torch.cuda.current_stream().cuda_stream)

    def rounds_async(self, r):
        try:
            self.ctx.shard_rounds_async(r, self.buf.data_ptr(), self._stream())
        except RuntimeError:
            self.status = 2
        return self.buf

    def apply_async(self, buf, ctl):
        try:
            self.ctx.shard_apply_async(buf.data_ptr(), ctl.data_ptr(), self._stream())
        except RuntimeError:
            self.status = 2
        if self.status:                                                # a host-side failure of this rank rides on the same reduce
            ctl[2] = -float(self.status)

    def read_control(self, ctl):
        c = ctl.cpu().numpy()                                          # the one synchronising copy per block of exchanges
        return float(c[0]), float(c[1]), int(round(-float(c[2])))

    def finalize(self):
        self.torch.cuda.synchronize()
        try:
            self.ctx.shard_finalize(self.dist.data_ptr(), self.pred.data_ptr())
        except RuntimeError:                                           # fixed-point check failed on this rank: all ranks stop together
            self.status = 2
        self.pred ^= -2147483648            # uint32 order -> int32 order (0xFFFFFFFF becomes INT32_MAX: neutral for MIN)
        return self.dist, self.pred

    def to_numpy(self, t):
        if t is self.pred:
            return (t ^ -2147483648).cpu().numpy()
        return t.cpu().numpy()


def torch_allreduce_min(dist):
    """in-place MIN all-reduce over the default process group (backend nccl == RCCL on ROCm; gloo on CPU)"""
    import torch

    def f(x):
        if isinstance(x, np.ndarray):
            t = torch.from_numpy(x)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        else:
            dist.all_reduce(x, op=dist.ReduceOp.MIN)
    return f


def plan_virtual_ranks(engines: Sequence, seed: int, target: int, goal_dist_offset: float = 0.3,
                       rounds_per_exchange: int = 8, max_exchanges: int = 100_000, check_every: int = 8,
                       device_loop: bool | None = None) -> ShardedResult:
    """`world` engines inside ONE process (one GPU standing in for several): the same protocol, the collective
    replaced by an elementwise minimum over the engines' buffers.  Lock-step version of run_sharded_plan (both of its
    loops: engines with the asynchronous steps run the device-resident one)."""
    for e in engines:
        e.begin(seed, target, goal_dist_offset)
    ctls = [e.control_buffer() for e in engines]
    if device_loop is None:
        device_loop = all(hasattr(e, "apply_async") for e in engines)

    def reduce_min(bufs):
        m = bufs[0].clone() if hasattr(bufs[0], "clone") else bufs[0].copy()
        for b in bufs[1:]:
            m = (m.minimum(b) if hasattr(m, "minimum") else np.minimum(m, b))
        for b in bufs:
            b[...] = m

    exchanges = rounds = 0
    while True:
        if device_loop:
            for _ in range(max(1, check_every)):
                bufs = [e.rounds_async(rounds_per_exchange) for e in engines]
                rounds += rounds_per_exchange
                reduce_min(bufs)
                for e, b, c in zip(engines, bufs, ctls):
                    e.apply_async(b, c)
                reduce_min(ctls)
                exchanges += 1
            gmin, gtarget, st = engines[0].read_control(ctls[0])
        else:
            bufs = [e.rounds(rounds_per_exchange) for e in engines]
            rounds += rounds_per_exchange
            reduce_min(bufs)
            for e, b, c in zip(engines, bufs, ctls):
                lm, td = e.apply(b)
                c[0] = lm
                c[1] = td
                c[2] = -float(getattr(e, "status", 0))
            reduce_min(ctls)
            exchanges += 1
            gmin, gtarget, st = float(ctls[0][0]), float(ctls[0][1]), int(round(-float(ctls[0][2])))
        if st:
            return ShardedResult(CANCELED if st == 1 else INTERNAL_ERROR, None, None, np.zeros(0, np.uint32), exchanges, rounds)
        if not np.isfinite(gmin) or gmin > np.float32(np.float64(np.float32(gtarget)) + goal_dist_offset):
            break
        if exchanges >= max_exchanges:
            raise RuntimeError("sharded plan did not terminate")
    outs = [e.finalize() for e in engines]
    reduce_min([o[0] for o in outs])
    reduce_min([o[1] for o in outs])
    e0 = engines[0]
    dist = e0.to_numpy(outs[0][0]).view(np.float32)
    pred = e0.to_numpy(outs[0][1]).view(np.uint32)
    code, path = walk_path(pred, seed, target)
    return ShardedResult(code, dist, pred, path, exchanges, rounds)

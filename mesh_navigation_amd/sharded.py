"""ONE Dijkstra plan on a mesh that is range-partitioned over several GPUs (BASELINE config 4).

One process per GPU.  Two ways to cut the mesh:

* PARTITIONED DATA (`partition_vertices` / `extract_part` / `PartitionedShardEngine`, the default of
  `bench.py --config C4 --gpus N`): the vertices are dealt to the processes by recursive coordinate bisection; a process
  uploads only ITS PART -- the vertices it owns, their 1-ring halo (neighbours owned elsewhere) and the edges with an owned
  endpoint, renumbered in ascending global id -- so mesh tables and per-vertex state on a GPU are ~1/world of the mesh.
  Every held copy of an interface vertex is exchanged (a value reached along real edges is an upper bound of the true
  distance; the minimum over the processes is what the owner would have computed), results stay sharded: the vertex path
  is walked across the processes segment by segment (`collect_partitioned`), V-sized arrays are only assembled on request.
* REPLICATED MESH, partitioned ownership (`GpuShardEngine`): every process holds the whole (read-only) mesh description but
  OWNS a contiguous range of the Morton-ordered LDS tiles and, with them, their vertices: only the owner relaxes into a vertex.

The loop is the same:

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
        if not np.isfinite(gmin) or gmin > np.float32(np.float64(np.float32(gtarget)) + max(goal_dist_offset, 0.0)):
            break                                                     # nothing left that may still propagate (a negative offset: the bound of offset 0, the finalize pass applies the rest)
        if exchanges >= max_exchanges:
            raise RuntimeError("sharded plan did not terminate")      # (the count is the same on every rank)

    def agreed_status() -> int:
        return int(round(-float(ctl[2])))

    dist_buf, pred_buf = engine.finalize()                            # a failing fixed-point check sets engine.status, it does not raise
    ctl[2] = -float(getattr(engine, "status", 0))
    allreduce_min(ctl)
    if agreed_status():
        return failed(agreed_status())
    if hasattr(engine, "part"):                                       # partitioned data: results stay sharded, the path is walked across the ranks
        code, dist, pred, path = collect_partitioned(engine, allreduce_min, seed, target, gather)
        return ShardedResult(code, dist, pred, path, exchanges, rounds)
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
        self.error = ""                                                # the library's message behind status 2
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
        except RuntimeError as ex:
            self.status, self.error = 2, str(ex)
        return self.buf

    def apply(self, buf):
        self.torch.cuda.synchronize()
        try:
            return self.ctx.shard_apply(buf.data_ptr())
        except RuntimeError as ex:                                     # rank-local failure: reported through the status word, the
            self.status, self.error = 2, str(ex)                       # other ranks must not be left waiting in the next all-reduce
            return float("inf"), float("inf")

    def drain(self):
        self.torch.cuda.synchronize()

    # -- the device-resident loop: kernels linked to torch's current stream by events, no host synchronisation
    def _stream(self) -> int:
        return int(self.torch.cuda.current_stream().cuda_stream)

    def rounds_async(self, r):
        try:
            self.ctx.shard_rounds_async(r, self.buf.data_ptr(), self._stream())
        except RuntimeError as ex:
            self.status, self.error = 2, str(ex)
        return self.buf

    def apply_async(self, buf, ctl):
        try:
            self.ctx.shard_apply_async(buf.data_ptr(), ctl.data_ptr(), self._stream())
        except RuntimeError as ex:
            self.status, self.error = 2, str(ex)
        if self.status:                                                # a host-side failure of this rank rides on the same reduce
            ctl[2] = -float(self.status)

    def read_control(self, ctl):
        c = ctl.cpu().numpy()                                          # the one synchronising copy per block of exchanges
        return float(c[0]), float(c[1]), int(round(-float(c[2])))

    def finalize(self):
        self.torch.cuda.synchronize()
        try:
            self.ctx.shard_finalize(self.dist.data_ptr(), self.pred.data_ptr())
        except RuntimeError as ex:                                     # fixed-point check failed on this rank: all ranks stop together
            self.status, self.error = 2, str(ex)
        self.pred ^= -2147483648            # uint32 order -> int32 order (0xFFFFFFFF becomes INT32_MAX: neutral for MIN)
        return self.dist, self.pred

    def to_numpy(self, t):
        if t is self.pred:
            return (t ^ -2147483648).cpu().numpy()
        return t.cpu().numpy()


# ---------------------------------------------------------------------------------------------------------------------
# Partitioned data: every process uploads only its part of the mesh
# ---------------------------------------------------------------------------------------------------------------------
NONE = 0xFFFFFFFF


def partition_vertices(xyz: np.ndarray, world: int) -> np.ndarray:
    """Owner rank of every vertex: recursive coordinate bisection (cuts across the widest axis of each piece, sizes
    proportional to the ranks on either side) -- compact parts, short interfaces.  Deterministic: every process computes the
    same array from the same coordinates."""
    V = xyz.shape[0]
    owner = np.zeros(V, np.uint16)

    def split(ids: np.ndarray, r0: int, r1: int):
        if r1 - r0 <= 1 or ids.size == 0:
            owner[ids] = r0
            return
        p = xyz[ids].astype(np.float64)
        ax = int(np.argmax(p.max(axis=0) - p.min(axis=0)))
        kl = (r1 - r0) // 2
        cut = (ids.size * kl) // (r1 - r0)
        order = np.lexsort((ids, p[:, ax]))                            # ties by vertex id: identical on every process
        split(ids[order[:cut]], r0, r0 + kl)
        split(ids[order[cut:]], r0 + kl, r1)

    split(np.arange(V, dtype=np.int64), 0, int(world))
    return owner


@dataclass
class MeshPart:
    """One process's part of a partitioned mesh (host arrays; local ids ascend with the global ids, so every (value, id)
    tie of the reference's pop order breaks exactly as on the whole mesh)."""
    rank: int
    world: int
    V_global: int
    gid: np.ndarray              # local id -> global id (owned + halo, ascending); two phantom vertices follow
    owned: np.ndarray            # uint8 per local vertex (phantoms 0)
    xyz: np.ndarray
    edges: np.ndarray            # local ids, in global edge order
    edge_gid: np.ndarray         # global edge ids of the local edges
    exchange_global: np.ndarray  # interface vertices (global ids): the same list on every process
    exchange_vertex: np.ndarray  # uint32 per interface vertex: local id or NONE

    @property
    def n_local(self) -> int:
        return self.gid.shape[0] + 2

    def local_of(self, g: int) -> int:
        i = int(np.searchsorted(self.gid, g))
        return i if i < self.gid.shape[0] and int(self.gid[i]) == int(g) else -1

    def owns(self, g: int) -> bool:
        i = self.local_of(g)
        return i >= 0 and bool(self.owned[i])

    def local_costs(self, vertex_costs: np.ndarray) -> np.ndarray:
        return np.concatenate([np.asarray(vertex_costs, np.float32)[self.gid], np.zeros(2, np.float32)])

    def local_edge_values(self, per_edge: np.ndarray) -> np.ndarray:
        return np.ascontiguousarray(np.asarray(per_edge)[self.edge_gid])

    def local_invalid(self, invalid: np.ndarray | None) -> np.ndarray | None:
        if invalid is None:
            return None
        return np.concatenate([np.asarray(invalid, np.uint8)[self.gid], np.zeros(2, np.uint8)])


def extract_part(xyz: np.ndarray, edges: np.ndarray, owner: np.ndarray, rank: int, world: int) -> MeshPart:
    """The part of process `rank`: owned vertices, their 1-ring halo, every edge with an owned endpoint, plus two isolated
    phantom vertices that stand in for a wave source / robot vertex held elsewhere (their values travel in the exchange)."""
    e = np.asarray(edges, np.int64)
    o0, o1 = owner[e[:, 0]], owner[e[:, 1]]
    keep = (o0 == rank) | (o1 == rank)
    edge_gid = np.nonzero(keep)[0]
    le = e[keep]
    own_ids = np.nonzero(owner == rank)[0]
    gid = np.unique(np.concatenate([own_ids, le.ravel()]))
    ledges = np.searchsorted(gid, le).astype(np.uint32)
    cross = o0 != o1
    exch = np.unique(np.concatenate([e[cross, 0], e[cross, 1]]))
    pos = np.searchsorted(gid, exch)
    pos_c = np.minimum(pos, max(gid.shape[0] - 1, 0))
    held = (gid[pos_c] == exch) if gid.shape[0] else np.zeros(exch.shape[0], bool)
    exchange_vertex = np.where(held, pos_c, NONE).astype(np.uint32)
    owned = np.concatenate([(owner[gid] == rank).astype(np.uint8), np.zeros(2, np.uint8)])
    p = np.asarray(xyz, np.float32)[gid]
    centre = p.mean(axis=0, dtype=np.float64).astype(np.float32) if gid.shape[0] else np.zeros(3, np.float32)
    lxyz = np.ascontiguousarray(np.concatenate([p, centre[None, :], centre[None, :]]), np.float32)
    return MeshPart(rank, world, int(xyz.shape[0]), gid, owned, lxyz, np.ascontiguousarray(ledges), edge_gid, exch, exchange_vertex)


_BIG = np.iinfo(np.int64).max


def _segment_of(part: MeshPart, pred_l, cur: int, seed: int, first: bool, segment: int) -> np.ndarray:
    """This process's contribution to one hop of the path walk: [count, next vertex, status, ids...] if it owns `cur`, the
    neutral element of the int64 MIN otherwise.  `pred_l`: the local predecessor array, or a callable (start, seed, cap) ->
    [hops, stop, status, local ids...] that walks on the device (mnav_shard_walk)."""
    seg = np.full(segment + 3, _BIG, np.int64)
    if not part.owns(cur):
        return seg
    n0 = part.gid.shape[0]
    if callable(pred_l):
        ls = part.local_of(seed)
        w = pred_l(part.local_of(cur), ls if ls >= 0 else n0, segment)
        n, stop, status = int(w[0]), int(w[1]), int(w[2])
        ids_l = w[3:3 + n].astype(np.int64)
        if status or (n and int(ids_l.max()) >= n0) or stop >= n0:        # no predecessor / a phantom on the path
            status = 1 if (first and n == 0) else 2
        seg[0], seg[1], seg[2] = n, (int(part.gid[stop]) if stop < n0 else cur), status
        seg[3:3 + n] = part.gid[np.minimum(ids_l, n0 - 1)]
        return seg
    ids: list[int] = []
    v, status = cur, 0
    while v != seed and part.owns(v) and len(ids) < segment:
        lv = part.local_of(v)
        lp = int(pred_l[lv])
        if lp == lv or lp >= n0:                                       # dijkstra :358: the wave never reached it
            status = 1 if (first and not ids) else 2
            break
        v = int(part.gid[lp])
        ids.append(v)
    seg[0], seg[1], seg[2] = len(ids), v, status
    seg[3:3 + len(ids)] = ids
    return seg


def _walk_segments(publish: Callable, seed: int, target: int, V_global: int):
    """dijkstra_mesh_planner.cpp:358-373 across the parts.  `publish(cur, first)` returns the MIN over all processes of their
    `_segment_of(cur)`: the owner's segment."""
    path: list[int] = []
    cur, code, first = int(target), SUCCESS, True
    while True:
        seg = publish(cur, first)
        first = False
        if seg[0] == _BIG:                                             # nobody owns `cur`: cannot happen on a consistent partition
            return INTERNAL_ERROR, np.zeros(0, np.uint32)
        n = int(seg[0])
        path.extend(int(x) for x in seg[3:3 + n])
        cur = int(seg[1])
        if seg[2]:
            return (NO_PATH_FOUND if seg[2] == 1 else INTERNAL_ERROR), np.zeros(0, np.uint32)
        if cur == seed:
            return SUCCESS, np.asarray(path[::-1], np.uint32)
        if len(path) > V_global or n == 0:
            return INTERNAL_ERROR, np.zeros(0, np.uint32)


def _owned_globals(part: MeshPart, dist_l: np.ndarray, pred_l: np.ndarray):
    """V-sized arrays holding this part's owned entries (global ids), neutral elements of MIN elsewhere."""
    n0 = part.gid.shape[0]
    mine = part.owned[:n0].astype(bool)
    dg = np.full(part.V_global, np.inf, np.float32)
    pg = np.full(part.V_global, _BIG, np.int64)
    dg[part.gid[mine]] = dist_l[:n0][mine]
    lp = np.minimum(pred_l[:n0][mine].astype(np.int64), n0 - 1)
    pg[part.gid[mine]] = part.gid[lp]
    return dg, pg


def collect_partitioned(engine, allreduce_min: Callable, seed: int, target: int, gather: bool, segment: int = 4096):
    """Results of a partitioned plan.  The vertex path is walked across the processes: whoever owns the current vertex
    follows its predecessors while they stay inside its part and publishes the segment (an int64 min-allreduce in which
    every other process contributes the neutral element); a path crosses the interfaces a handful of times.  With `gather`
    the V-sized potential / predecessor arrays are assembled as well (validation only: the one place where something
    mesh-sized exists per process)."""
    part: MeshPart = engine.part
    if not gather and hasattr(engine, "walker"):
        dist_l, pred_l = None, engine.walker()                         # the arrays stay on the device, segments are walked there
    else:
        dist_l, pred_l = engine.local_result()                         # local arrays; predecessors are local ids
    code, path = _walk_segments(lambda cur, first: engine.reduce_int64(_segment_of(part, pred_l, cur, seed, first, segment), allreduce_min),
                                seed, target, part.V_global)
    dist = pred = None
    if gather:
        dg, pg = _owned_globals(part, dist_l, pred_l)
        dist = engine.reduce_float32(dg, allreduce_min)
        pred = engine.reduce_int64(pg, allreduce_min).astype(np.uint32)
    return code, dist, pred, path


class PartitionedShardEngine(GpuShardEngine):
    """mnav_shard_* on ONE PART of the mesh (mnav_shard_setup_partition): the context was created on the part's arrays
    (`upload_part`); seeds and targets are global ids, translated here (a vertex held elsewhere is a phantom)."""

    def __init__(self, ctx, part: MeshPart, cost_limit: float = 1.0, device=None):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.part = part
        self.cost_limit = cost_limit
        self.n = ctx.shard_setup_partition(part.exchange_vertex, part.owned)
        self.info = ctx.shard_info()
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.dev = dev
        self.buf = torch.empty(self.n, dtype=torch.float32, device=dev)
        self.ctl = torch.zeros(3, dtype=torch.float32, device=dev)
        self.status = 0
        self.error = ""
        self.dist = torch.empty(ctx.V, dtype=torch.float32, device=dev)   # sized by the PART
        self.pred = torch.empty(ctx.V, dtype=torch.int32, device=dev)

    @staticmethod
    def upload_part(ctx, part: MeshPart, vertex_costs: np.ndarray, edge_weights: np.ndarray, invalid: np.ndarray | None = None):
        ctx.upload_mesh(part.xyz, np.zeros((0, 3), np.uint32), part.edges, None)
        ctx.upload_costs(part.local_costs(vertex_costs), part.local_edge_values(edge_weights).astype(np.float32), part.local_invalid(invalid))

    def begin(self, seed, target, offset):
        self.status = 0
        ls, lt = self.part.local_of(seed), self.part.local_of(target)
        n0 = self.part.gid.shape[0]
        if lt < 0 and offset < 0:                                     # (value, id) ties at the robot vertex's potential: its rank among this part's ids
            self.ctx.shard_set_goal_tie(int(np.searchsorted(self.part.gid, target)))
        self.ctx.shard_begin(ls if ls >= 0 else n0, lt if lt >= 0 else n0 + 1, offset, self.cost_limit)

    def finalize(self):
        self.torch.cuda.synchronize()
        try:
            self.ctx.shard_finalize(self.dist.data_ptr(), self.pred.data_ptr())
        except RuntimeError as ex:
            self.status, self.error = 2, str(ex)
        return self.dist, self.pred

    def local_result(self):
        return self.dist.cpu().numpy(), self.pred.cpu().numpy().view(np.uint32)

    def walker(self):
        return lambda start, seed, cap: self.ctx.shard_walk(start, seed, cap)

    def _reduce(self, host: np.ndarray, allreduce_min):
        t = self.torch.from_numpy(host).to(self.dev)
        allreduce_min(t)
        return t.cpu().numpy()

    def reduce_int64(self, host, allreduce_min):
        return self._reduce(host, allreduce_min)

    def reduce_float32(self, host, allreduce_min):
        return self._reduce(host, allreduce_min)


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
                       device_loop: bool | None = None, gather: bool = True) -> ShardedResult:
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
        if not np.isfinite(gmin) or gmin > np.float32(np.float64(np.float32(gtarget)) + max(goal_dist_offset, 0.0)):
            break
        if exchanges >= max_exchanges:
            raise RuntimeError("sharded plan did not terminate")
    outs = [e.finalize() for e in engines]
    if any(getattr(e, "status", 0) for e in engines):
        return ShardedResult(INTERNAL_ERROR, None, None, np.zeros(0, np.uint32), exchanges, rounds)
    if hasattr(engines[0], "part"):                                   # partitioned data: the same collection, the collectives done by hand
        if not gather and all(hasattr(e, "walker") for e in engines):  # nothing V-sized leaves the devices: segments are walked there
            walkers = [e.walker() for e in engines]
            code, path = _walk_segments(lambda cur, first: np.minimum.reduce([_segment_of(e.part, w, cur, seed, first, 4096) for e, w in zip(engines, walkers)]),
                                        seed, target, engines[0].part.V_global)
            return ShardedResult(code, None, None, path, exchanges, rounds)
        loc = [e.local_result() for e in engines]
        code, path = _walk_segments(lambda cur, first: np.minimum.reduce([_segment_of(e.part, l[1], cur, seed, first, 4096) for e, l in zip(engines, loc)]),
                                    seed, target, engines[0].part.V_global)
        glob = [_owned_globals(e.part, l[0], l[1]) for e, l in zip(engines, loc)]
        dist = np.minimum.reduce([g[0] for g in glob])
        pred = np.minimum.reduce([g[1] for g in glob]).astype(np.uint32)
        return ShardedResult(code, dist, pred, path, exchanges, rounds)
    reduce_min([o[0] for o in outs])
    reduce_min([o[1] for o in outs])
    e0 = engines[0]
    dist = e0.to_numpy(outs[0][0]).view(np.float32)
    pred = e0.to_numpy(outs[0][1]).view(np.uint32)
    code, path = walk_path(pred, seed, target)
    return ShardedResult(code, dist, pred, path, exchanges, rounds)

"""CPU model of ONE rank of the sharded single plan (mesh_navigation_amd/sharded.py) -- test infrastructure.

Same protocol as the GPU engine (own a part of the vertices, relax only into owned vertices, exchange the
interface values with a min-allreduce, wake on dropped ghosts, finalize the owned part with the reference's
cut-off semantics), in plain numpy on small meshes.  Ownership here is by vertex-id range (row strips of the grid
terrain); the GPU engine owns ranges of Morton tiles -- the protocol does not care."""
import numpy as np


def goal_cut_expanded(d, target, offset, tie=None):
    """mnav_eval.h goal_cut / expanded_source on an array of potentials: (mask of expanded sources, cut value).  With a negative offset the
    reference expands exactly what popped before the robot vertex (dijkstra :293-300): d < dt, or d == dt with a smaller id (`tie`: the id
    that stands for the robot vertex in that comparison, default the target itself)."""
    dt = d[target]
    if not np.isfinite(dt):
        return np.isfinite(d), np.float32(np.inf)
    goal = np.float32(np.float64(dt) + offset)
    if goal < dt:
        t = target if tie is None else tie
        ids = np.arange(d.shape[0])
        return np.isfinite(d) & ((d < dt) | ((d == dt) & (ids < t))), np.float32(dt)
    return np.isfinite(d) & (d <= goal), goal


class ModelShardEngine:
    def __init__(self, mesh, weights, costs, rank, world, cost_limit=1.0, invalid=None):
        self.V = mesh.V
        self.rank, self.world = rank, world
        lo = (mesh.V * np.arange(world + 1)) // world
        self.owner = (np.searchsorted(lo, np.arange(mesh.V), side="right") - 1).astype(np.int64)
        e = mesh.edges.astype(np.int64)
        self.src = np.concatenate([e[:, 0], e[:, 1]])
        self.dst = np.concatenate([e[:, 1], e[:, 0]])
        w = np.concatenate([weights, weights]).astype(np.float32)
        inv = np.zeros(mesh.V, bool) if invalid is None else np.asarray(invalid, bool)
        # cut-offs folded into the gather weights like the device does (dijkstra :302, :328)
        w = np.where(inv[self.dst] | (costs[self.src].astype(np.float64) > cost_limit), np.float32(np.inf), w)
        self.w = w
        self.mine = self.owner[self.dst] == rank                       # edges this rank relaxes
        cross = self.owner[self.src] != self.owner[self.dst]
        self.iface = np.unique(np.concatenate([self.src[cross], self.dst[cross]]))
        self.iface_owned = self.owner[self.iface] == rank

    def begin(self, seed, target, offset):
        self.seed, self.target, self.offset = seed, target, offset
        self.dist = np.full(self.V, np.inf, np.float32)
        self.dist[seed] = 0.0
        self.pending = True                                           # something may still propagate locally
        self.ctl = np.zeros(3, np.float32)                             # [min pending, dist[target], -status]
        self.status = 0

    def control_buffer(self):
        return self.ctl

    def _bound(self):
        return np.float32(np.float64(self.dist[self.target]) + max(self.offset, 0.0))   # (a negative offset: the bound of offset 0, the finalize applies the rest)

    def rounds(self, r):
        d = self.dist
        self.moved = np.zeros(self.V, bool)
        for _ in range(r):
            ok = self.mine & (d[self.src] <= self._bound())            # sources above the bound never relax
            cand = (d[self.src] + self.w).astype(np.float32)           # the float add of dijkstra :331
            new = d.copy()
            np.minimum.at(new, self.dst[ok], cand[ok])
            ch = new < d
            if not ch.any():
                break
            self.moved |= ch
            d = new
        self.dist = d
        buf = np.full(len(self.iface) + 1, np.inf, np.float32)
        buf[:-1][self.iface_owned] = d[self.iface[self.iface_owned]]
        buf[-1] = d[self.target]
        return buf

    def apply(self, buf):
        g = ~self.iface_owned
        v = self.iface[g]
        drop = buf[:-1][g] < self.dist[v]
        self.dist[v[drop]] = buf[:-1][g][drop]
        if buf[-1] < self.dist[self.target]:
            self.dist[self.target] = buf[-1]
        # pending = values that still have to be pushed: what moved in the last local sweeps + the dropped ghosts
        cand = np.concatenate([self.dist[self.moved & (self.owner == self.rank)], self.dist[v[drop]]])
        cand = cand[cand <= self._bound()]
        # (a vertex that moved in the LAST sweep of rounds() has not propagated yet; earlier ones have -- a coarse
        #  but safe over-approximation: one more exchange at the end finds nothing moved)
        local_min = np.float32(cand.min()) if cand.size else np.float32(np.inf)
        return float(local_min), float(self.dist[self.target])

    def finalize(self):
        """cut-off semantics + predecessors for the owned vertices (what k_dij_finalize does per tile)"""
        d = self.dist
        expanded, goal = goal_cut_expanded(d, self.target, self.offset)
        ok = self.mine & expanded[self.src]
        cand = (d[self.src] + self.w).astype(np.float32)
        best = np.full(self.V, np.inf, np.float32)
        np.minimum.at(best, self.dst[ok], cand[ok])
        out = d.copy()
        above = (self.owner == self.rank) & ~(d <= goal)
        out[above] = best[above]                                      # tentative value from expanded sources only
        out[self.seed] = 0.0
        pred = np.arange(self.V, dtype=np.int64)
        att = ok & (cand == out[self.dst]) & np.isfinite(cand)
        key = np.full(self.V, np.iinfo(np.int64).max, np.int64)        # argmin (d[u], u) among the attaining edges
        k = (d[self.src].view(np.uint32).astype(np.int64) << 32) | self.src
        np.minimum.at(key, self.dst[att], k[att])
        has = key != np.iinfo(np.int64).max
        pred[has] = key[has] & 0xFFFFFFFF
        pred[self.seed] = self.seed
        mine = self.owner == self.rank
        dist_buf = np.where(mine, out, np.float32(np.inf)).astype(np.float32)
        pred_buf = np.where(mine, pred, 0xFFFFFFFF).astype(np.uint32)
        # like the GPU engine: uint32 order -> int32 order around the MIN collective (gloo / RCCL have no uint32)
        self._pred = (pred_buf ^ np.uint32(0x80000000)).view(np.int32)
        return dist_buf, self._pred

    def to_numpy(self, x):
        if x is self._pred:
            return (x.view(np.uint32) ^ np.uint32(0x80000000))
        return x


class AsyncModelShardEngine(ModelShardEngine):
    """The same engine behind the asynchronous step protocol of the device-resident loop (rounds_async / apply_async /
    read_control): on the CPU the steps are of course synchronous -- what the tests exercise is the LOOP: termination words
    written by the engine, read back only every `check_every` exchanges, exchanges after convergence being harmless."""

    def rounds_async(self, r):
        return self.rounds(r)

    def apply_async(self, buf, ctl):
        lm, td = self.apply(buf)
        ctl[0] = lm
        ctl[1] = td
        ctl[2] = -float(self.status)

    def read_control(self, ctl):
        return float(ctl[0]), float(ctl[1]), int(round(-float(ctl[2])))


class PartModelEngine:
    """CPU model of one rank of the PARTITIONED plan (sharded.MeshPart): the engine only sees its part -- owned vertices, their
    1-ring halo, the edges with an owned endpoint, local ids -- relaxes along every local edge (a halo copy's local value is an
    upper bound of its true distance), packs every held copy of an interface vertex and lets every held copy take the reduced
    minimum.  What mnav_shard_setup_partition does on the device."""

    def __init__(self, part, weights_local, costs_local, cost_limit=1.0, invalid_local=None, asynchronous=False):
        self.part = part
        self.n = part.n_local
        e = part.edges.astype(np.int64)
        self.src = np.concatenate([e[:, 0], e[:, 1]])
        self.dst = np.concatenate([e[:, 1], e[:, 0]])
        w = np.concatenate([weights_local, weights_local]).astype(np.float32)
        inv = np.zeros(self.n, bool) if invalid_local is None else np.asarray(invalid_local, bool)
        self.w = np.where(inv[self.dst] | (np.asarray(costs_local, np.float64)[self.src] > cost_limit), np.float32(np.inf), w)
        self.held = part.exchange_vertex != 0xFFFFFFFF
        self.ex = part.exchange_vertex[self.held].astype(np.int64)
        self.status = 0
        if asynchronous:                                               # the device-resident loop's protocol (see AsyncModelShardEngine)
            self.rounds_async = self.rounds
            self.apply_async = self._apply_async
            self.read_control = lambda ctl: (float(ctl[0]), float(ctl[1]), int(round(-float(ctl[2]))))

    def begin(self, seed, target, offset):
        n0 = self.part.gid.shape[0]
        ls, lt = self.part.local_of(seed), self.part.local_of(target)
        self.seed, self.target, self.offset = (ls if ls >= 0 else n0), (lt if lt >= 0 else n0 + 1), offset
        self.tie = None if lt >= 0 else int(np.searchsorted(self.part.gid, target))   # the robot vertex's rank among this part's ids (mnav_shard_set_goal_tie)
        self.dist = np.full(self.n, np.inf, np.float32)
        self.dist[self.seed] = 0.0
        self.ctl = np.zeros(3, np.float32)
        self.status = 0

    def control_buffer(self):
        return self.ctl

    def _bound(self):
        return np.float32(np.float64(self.dist[self.target]) + max(self.offset, 0.0))   # (a negative offset: the bound of offset 0, the finalize applies the rest)

    def rounds(self, r):
        d = self.dist
        self.moved = np.zeros(self.n, bool)
        for _ in range(r):
            ok = d[self.src] <= self._bound()
            cand = (d[self.src] + self.w).astype(np.float32)
            new = d.copy()
            np.minimum.at(new, self.dst[ok], cand[ok])
            ch = new < d
            if not ch.any():
                break
            self.moved |= ch
            d = new
        self.dist = d
        buf = np.full(self.part.exchange_vertex.shape[0] + 1, np.inf, np.float32)
        buf[:-1][self.held] = d[self.ex]
        buf[-1] = d[self.target]
        return buf

    def apply(self, buf):
        got = buf[:-1][self.held]
        drop = got < self.dist[self.ex]
        self.dist[self.ex[drop]] = got[drop]
        if buf[-1] < self.dist[self.target]:
            self.dist[self.target] = buf[-1]
        cand = np.concatenate([self.dist[self.moved], self.dist[self.ex[drop]]])
        cand = cand[cand <= self._bound()]
        local_min = np.float32(cand.min()) if cand.size else np.float32(np.inf)
        return float(local_min), float(self.dist[self.target])

    def _apply_async(self, buf, ctl):
        lm, td = self.apply(buf)
        ctl[0], ctl[1], ctl[2] = lm, td, -float(self.status)

    def finalize(self):
        d = self.dist
        expanded, goal = goal_cut_expanded(d, self.target, self.offset, getattr(self, "tie", None))
        ok = expanded[self.src]
        cand = (d[self.src] + self.w).astype(np.float32)
        best = np.full(self.n, np.inf, np.float32)
        np.minimum.at(best, self.dst[ok], cand[ok])
        out = d.copy()
        above = ~(d <= goal)
        out[above] = best[above]
        out[self.seed] = 0.0
        pred = np.arange(self.n, dtype=np.int64)
        att = ok & (cand == out[self.dst]) & np.isfinite(cand)
        key = np.full(self.n, np.iinfo(np.int64).max, np.int64)
        k = (d[self.src].view(np.uint32).astype(np.int64) << 32) | self.src      # local ids ascend with the global ids: same ties
        np.minimum.at(key, self.dst[att], k[att])
        has = key != np.iinfo(np.int64).max
        pred[has] = key[has] & 0xFFFFFFFF
        pred[self.seed] = self.seed
        self._dist, self._pred = out.astype(np.float32), pred.astype(np.uint32)
        return self._dist, self._pred

    def local_result(self):
        return self._dist, self._pred

    def walker(self):
        """the contract of mnav_shard_walk (include/mnav.h): [hops, stop vertex, status, local ids...] of one path segment inside this part"""
        owned = self.part.owned

        def walk(start, seed, cap):
            out = np.zeros(cap + 3, np.uint32)
            v, n, status = int(start), 0, 0
            while v != seed and owned[v] and n < cap:
                p = int(self._pred[v])
                if p == v:
                    status = 1
                    break
                out[3 + n] = p
                n += 1
                v = p
            out[0], out[1], out[2] = n, v, status
            return out
        return walk

    @staticmethod
    def reduce_int64(host, allreduce_min):
        allreduce_min(host)
        return host

    reduce_float32 = reduce_int64

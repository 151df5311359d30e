"""ctypes loader for the CPU oracle (oracle/mnav_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from the product package (mesh_navigation_amd/).
Pinned against the reference's own code (oracle/_ref, tests/test_ref_pins_oracle.py); see oracle/mnav_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmnav_oracle.so")
NONE = 0xFFFFFFFF

SUCCESS, CANCELED, INVALID_START, INVALID_GOAL, NO_PATH_FOUND = 0, 51, 52, 53, 54


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "mnav_oracle.c")
    hdr = os.path.join(_HERE, "mnav_oracle.h")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libmnav_oracle.so"])
    return _LIB_PATH


class _Stats(C.Structure):
    _fields_ = [("fixed_set_cnt", C.c_uint64), ("expanded", C.c_uint64), ("relaxations", C.c_uint64),
                ("edge_visits", C.c_uint64), ("goal_dist", C.c_float), ("t_init_ms", C.c_double),
                ("t_propagation_ms", C.c_double), ("t_backtrack_ms", C.c_double)]


class InflationCfg(C.Structure):
    _fields_ = [("inscribed_radius", C.c_double), ("inflation_radius", C.c_double),
                ("lethal_value", C.c_double), ("inscribed_value", C.c_double),
                ("cost_scaling_factor", C.c_double)]

    @staticmethod
    def defaults() -> "InflationCfg":
        # mesh_layers/include/mesh_layers/inflation_layer.h:240-248
        return InflationCfg(0.25, 0.4, 1.0, 0.99, 1.0)


class _InflationField(C.Structure):
    _fields_ = [("distances", C.c_void_p), ("vecmap", C.c_void_p), ("cfg", InflationCfg),
                ("repulsive_field", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, u32, f32, f64 = C.c_void_p, C.c_uint32, C.c_float, C.c_double
        L.mo_mesh_create.restype = vp
        L.mo_mesh_create.argtypes = [u32, u32, vp, vp]
        L.mo_mesh_destroy.argtypes = [vp]
        for n in ("mo_mesh_num_vertices", "mo_mesh_num_faces", "mo_mesh_num_edges"):
            getattr(L, n).restype = u32
            getattr(L, n).argtypes = [vp]
        L.mo_mesh_is_manifold.restype = C.c_int
        L.mo_mesh_is_manifold.argtypes = [vp]
        L.mo_mesh_vertex_faces.argtypes = [vp, vp, vp]
        L.mo_mesh_vertex_edges.argtypes = [vp, vp, vp]
        L.mo_mesh_edges.argtypes = [vp, vp]
        L.mo_mesh_face_edges.argtypes = [vp, vp]
        L.mo_edge_distances.argtypes = [vp, vp]
        L.mo_face_normals.argtypes = [vp, vp]
        L.mo_vertex_normals.argtypes = [vp, vp, vp]
        L.mo_compute_edge_weights.argtypes = [vp, vp, vp, f64, vp]
        L.mo_steepness.argtypes = [vp, vp, f64, vp, vp]
        L.mo_inflation_sethian.restype = f32
        L.mo_inflation_sethian.argtypes = [f32] * 6
        L.mo_inflation_fading.restype = f32
        L.mo_inflation_fading.argtypes = [C.POINTER(InflationCfg), f32]
        L.mo_inflation_wavefront_update.restype = C.c_int
        L.mo_inflation_wavefront_update.argtypes = [vp, vp, vp, f32, vp, u32, u32, u32]
        L.mo_inflation.argtypes = [vp, C.POINTER(InflationCfg), vp, vp, vp, vp, vp, vp]
        L.mo_combine.argtypes = [u32, C.c_int, C.c_int, vp, vp, vp]
        L.mo_set_heap_ties_by_id.argtypes = [C.c_int]
        L.mo_meap_create.restype = vp
        L.mo_meap_create.argtypes = [u32]
        L.mo_meap_destroy.argtypes = [vp]
        L.mo_meap_insert.argtypes = [vp, u32, f32]
        L.mo_meap_empty.restype = C.c_int
        L.mo_meap_empty.argtypes = [vp]
        L.mo_meap_pop_min.restype = u32
        L.mo_meap_pop_min.argtypes = [vp, C.POINTER(f32)]
        L.mo_dijkstra.restype = u32
        L.mo_dijkstra.argtypes = [vp, vp, vp, vp, u32, u32, f64, f64, vp, vp, vp, C.POINTER(u32), vp,
                                  C.POINTER(_Stats)]
        L.mo_dijkstra_pred_rule.argtypes = [vp, vp, vp, vp, u32, f32, f64, vp, vp]
        L.mo_dijkstra_vector_map.argtypes = [vp, vp, vp]
        L.mo_cvp_update_scalar.restype = C.c_int
        L.mo_cvp_update_scalar.argtypes = [f32] * 6 + [C.POINTER(f32), C.POINTER(C.c_int), C.POINTER(f32)]
        L.mo_cvp_propagate.restype = u32
        L.mo_cvp_propagate.argtypes = [vp, vp, vp, vp, vp, vp, u32, u32, f64, f64, vp, vp, vp, vp, vp, vp,
                                       vp, C.POINTER(_Stats)]
        L.mo_cvp_backtrack.restype = u32
        L.mo_cvp_backtrack.argtypes = [vp, vp, vp, vp, vp, u32, vp, u32, f64, u32, vp, vp, C.POINTER(u32)]
        L.mo_inflation_vector_at.argtypes = [C.POINTER(_InflationField), vp, vp, vp]
        L.mo_nearest_vertex.restype = u32
        L.mo_nearest_vertex.argtypes = [vp, vp]
        L.mo_containing_face.restype = u32
        L.mo_containing_face.argtypes = [vp, vp, vp]
        L.mo_projected_barycentric.restype = C.c_int
        L.mo_projected_barycentric.argtypes = [vp, vp, vp, vp, vp, C.POINTER(f32)]
        L.mo_pose_from_position.restype = f32
        L.mo_pose_from_position.argtypes = [vp, vp, vp, vp]
        L.mo_dijkstra_poses.restype = u32
        L.mo_dijkstra_poses.argtypes = [vp, vp, vp, u32, vp, vp, vp, C.POINTER(f64)]
        L.mo_cvp_poses.restype = u32
        L.mo_cvp_poses.argtypes = [vp, vp, vp, vp, u32, vp, vp, C.POINTER(f64)]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


@dataclass
class DijkstraResult:
    code: int
    dist: np.ndarray
    pred: np.ndarray
    path: np.ndarray        # dijkstra() list order: seed first ... pred[target]
    stats: dict


@dataclass
class CvpResult:
    code: int
    dist: np.ndarray
    pred: np.ndarray
    direction: np.ndarray
    cutface: np.ndarray
    vecmap: np.ndarray
    has_vec: np.ndarray
    stats: dict


def _stats_dict(s: _Stats) -> dict:
    return {k: getattr(s, k) for k, _ in _Stats._fields_}


class OracleMesh:
    """Half-edge-mesh stand-in with the conventions documented in mnav_oracle.h."""

    def __init__(self, xyz: np.ndarray, faces: np.ndarray):
        self.xyz = _f32(xyz).reshape(-1, 3)
        self.faces = _u32(faces).reshape(-1, 3)
        self.V = self.xyz.shape[0]
        self.F = self.faces.shape[0]
        self._h = lib().mo_mesh_create(self.V, self.F, _p(self.xyz), _p(self.faces))
        self.E = lib().mo_mesh_num_edges(self._h)

    def __del__(self):
        try:
            if self._h:
                lib().mo_mesh_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def edges(self) -> np.ndarray:
        out = np.empty((self.E, 2), dtype=np.uint32)
        lib().mo_mesh_edges(self._h, _p(out))
        return out

    @property
    def manifold(self) -> bool:
        return bool(lib().mo_mesh_is_manifold(self._h))

    def vertex_faces(self):
        """getFacesOfVertex order (half-edge circulator) as CSR: (ptr[V+1], faces[3F])."""
        ptr = np.empty(self.V + 1, dtype=np.uint32)
        vf = np.empty(3 * self.F, dtype=np.uint32)
        lib().mo_mesh_vertex_faces(self._h, _p(ptr), _p(vf))
        return ptr, vf

    def vertex_edges(self):
        ptr = np.empty(self.V + 1, dtype=np.uint32)
        ve = np.empty(2 * self.E, dtype=np.uint32)
        lib().mo_mesh_vertex_edges(self._h, _p(ptr), _p(ve))
        return ptr, ve

    def face_edges(self) -> np.ndarray:
        out = np.empty((self.F, 3), dtype=np.uint32)
        lib().mo_mesh_face_edges(self._h, _p(out))
        return out

    def edge_distances(self) -> np.ndarray:
        out = np.empty(self.E, dtype=np.float32)
        lib().mo_edge_distances(self._h, _p(out))
        return out

    def face_normals(self) -> np.ndarray:
        out = np.empty((self.F, 3), dtype=np.float32)
        lib().mo_face_normals(self._h, _p(out))
        return out

    def vertex_normals(self, fn: np.ndarray | None = None) -> np.ndarray:
        fn = self.face_normals() if fn is None else _f32(fn)
        out = np.empty((self.V, 3), dtype=np.float32)
        lib().mo_vertex_normals(self._h, _p(fn), _p(out))
        return out

    def edge_weights(self, edge_dist, vertex_costs, edge_cost_factor: float) -> np.ndarray:
        out = np.empty(self.E, dtype=np.float32)
        ed, vc = _f32(edge_dist), _f32(vertex_costs)
        lib().mo_compute_edge_weights(self._h, _p(ed), _p(vc), float(edge_cost_factor), _p(out))
        return out

    def steepness(self, vertex_normals, threshold: float = 0.3):
        vn = _f32(vertex_normals)
        st = np.empty(self.V, dtype=np.float32)
        le = np.empty(self.V, dtype=np.uint8)
        lib().mo_steepness(self._h, _p(vn), float(threshold), _p(st), _p(le))
        return st, le

    def inflation(self, lethal, edge_dist, cfg: InflationCfg | None = None, invalid=None):
        cfg = cfg or InflationCfg.defaults()
        le, ed = _u8(lethal), _f32(edge_dist)
        inv = None if invalid is None else _u8(invalid)
        cost = np.empty(self.V, dtype=np.float32)
        dist = np.empty(self.V, dtype=np.float32)
        vec = np.empty((self.V, 3), dtype=np.float32)
        lib().mo_inflation(self._h, C.byref(cfg), _p(le), _p(inv), _p(ed), _p(cost), _p(dist), _p(vec))
        return cost, dist, vec

    def inflation_wavefront_update(self, dist, vecmap, max_distance, edge_weights, v1, v2, v3) -> bool:
        return bool(lib().mo_inflation_wavefront_update(self._h, _p(dist), _p(vecmap), float(max_distance),
                                                        _p(_f32(edge_weights)), v1, v2, v3))

    def dijkstra(self, edge_weights, vertex_costs, seed_vertex: int, target_vertex: int,
                 goal_dist_offset: float = 0.3, cost_limit: float = 1.0, invalid=None) -> DijkstraResult:
        w, vc = _f32(edge_weights), _f32(vertex_costs)
        inv = np.zeros(self.V, dtype=np.uint8) if invalid is None else _u8(invalid)
        dist = np.empty(self.V, dtype=np.float32)
        pred = np.empty(self.V, dtype=np.uint32)
        path = np.empty(max(self.V, 1), dtype=np.uint32)
        n = C.c_uint32(0)
        st = _Stats()
        code = lib().mo_dijkstra(self._h, _p(w), _p(vc), _p(inv), int(seed_vertex), int(target_vertex),
                                 float(goal_dist_offset), float(cost_limit), _p(dist), _p(pred), _p(path),
                                 C.byref(n), None, C.byref(st))
        return DijkstraResult(code, dist, pred, path[: n.value].copy(), _stats_dict(st))

    def dijkstra_pred_rule(self, edge_weights, vertex_costs, seed_vertex, goal_dist, cost_limit, dist,
                           invalid=None) -> np.ndarray:
        w, vc, d = _f32(edge_weights), _f32(vertex_costs), _f32(dist)
        inv = None if invalid is None else _u8(invalid)
        pred = np.empty(self.V, dtype=np.uint32)
        lib().mo_dijkstra_pred_rule(self._h, _p(w), _p(vc), _p(inv), int(seed_vertex), float(goal_dist),
                                    float(cost_limit), _p(d), _p(pred))
        return pred

    def dijkstra_vector_map(self, pred) -> np.ndarray:
        vm = np.zeros((self.V, 3), dtype=np.float32)
        lib().mo_dijkstra_vector_map(self._h, _p(_u32(pred)), _p(vm))
        return vm

    def cvp(self, edge_weights, vertex_costs, vertex_normals, seed_pos, seed_face: int, target_face: int,
            goal_dist_offset: float = 0.3, cost_limit: float = 1.0, invalid=None, direction=None,
            cutface=None) -> CvpResult:
        w, vc, vn = _f32(edge_weights), _f32(vertex_costs), _f32(vertex_normals)
        inv = np.zeros(self.V, dtype=np.uint8) if invalid is None else _u8(invalid)
        sp = _f32(seed_pos)
        dist = np.empty(self.V, dtype=np.float32)
        pred = np.empty(self.V, dtype=np.uint32)
        direction = np.zeros(self.V, dtype=np.float32) if direction is None else _f32(direction).copy()
        cutface = np.full(self.V, NONE, dtype=np.uint32) if cutface is None else _u32(cutface).copy()
        vecmap = np.empty((self.V, 3), dtype=np.float32)
        has_vec = np.empty(self.V, dtype=np.uint8)
        st = _Stats()
        code = lib().mo_cvp_propagate(self._h, _p(w), _p(vc), _p(inv), _p(vn), _p(sp), int(seed_face),
                                      int(target_face), float(goal_dist_offset), float(cost_limit), _p(dist),
                                      _p(pred), _p(direction), _p(cutface), _p(vecmap), _p(has_vec), None,
                                      C.byref(st))
        return CvpResult(code, dist, pred, direction, cutface, vecmap, has_vec, _stats_dict(st))

    def cvp_backtrack(self, vecmap, has_vec, seed_pos, seed_face, target_pos, target_face,
                      step_width: float = 0.4, inflation_field=None, cap: int = 100000):
        vm, hv = _f32(vecmap), _u8(has_vec)
        sp, tp = _f32(seed_pos), _f32(target_pos)
        pos = np.empty((cap, 3), dtype=np.float32)
        face = np.empty(cap, dtype=np.uint32)
        n = C.c_uint32(0)
        fld = None
        keep = None
        if inflation_field is not None:
            d, v, cfg, rep = inflation_field
            keep = (_f32(d), _f32(v))
            fld = _InflationField(keep[0].ctypes.data, keep[1].ctypes.data, cfg, int(rep))
        code = lib().mo_cvp_backtrack(self._h, _p(vm), _p(hv), C.byref(fld) if fld is not None else None,
                                      _p(sp), int(seed_face), _p(tp), int(target_face), float(step_width),
                                      cap, _p(pos), _p(face), C.byref(n))
        return code, pos[: n.value].copy(), face[: n.value].copy()

    def nearest_vertex(self, p) -> int:
        return int(lib().mo_nearest_vertex(self._h, _p(_f32(p))))

    def containing_face(self, p):
        bary = np.zeros(3, dtype=np.float32)
        f = int(lib().mo_containing_face(self._h, _p(_f32(p)), _p(bary)))
        return f, bary

    def dijkstra_poses(self, vertex_normals, path, robot_pos, goal_pos):
        vn, pa = _f32(vertex_normals), _u32(path)
        poses = np.empty((len(pa) + 1, 7), dtype=np.float64)
        cost = C.c_double(0)
        n = lib().mo_dijkstra_poses(self._h, _p(vn), _p(pa), len(pa), _p(_f32(robot_pos)), _p(_f32(goal_pos)),
                                    _p(poses), C.byref(cost))
        return poses[:n].copy(), cost.value

    def cvp_poses(self, face_normals, path_pos, path_face, goal_pose):
        fn, pp, pf = _f32(face_normals), _f32(path_pos), _u32(path_face)
        gp = np.ascontiguousarray(goal_pose, dtype=np.float64)
        poses = np.empty((len(pf) + 1, 7), dtype=np.float64)
        cost = C.c_double(0)
        n = lib().mo_cvp_poses(self._h, _p(fn), _p(pp), _p(pf), len(pf), _p(gp), _p(poses), C.byref(cost))
        return poses[:n].copy(), cost.value


def inflation_vector_at(distances, vecmap, cfg: InflationCfg, repulsive_field: bool, vs, bary) -> np.ndarray:
    d, v = _f32(distances), _f32(vecmap)
    fld = _InflationField(d.ctypes.data, v.ctypes.data, cfg, int(repulsive_field))
    out = np.zeros(3, dtype=np.float32)
    lib().mo_inflation_vector_at(C.byref(fld), _p(_u32(vs)), _p(_f32(bary)), _p(out))
    return out


def set_heap_ties_by_id(on: bool) -> None:
    """True (default): equal keys pop in ascending vertex id (the device rule); False: plain lvr2-style heap."""
    lib().mo_set_heap_ties_by_id(1 if on else 0)


def combine(layers, weights=None, mode: str = "avg") -> np.ndarray:
    arrs = [_f32(a) for a in layers]
    V = arrs[0].shape[0]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    w = _f32(weights if weights is not None else [1.0] * len(arrs))
    out = np.empty(V, dtype=np.float32)
    lib().mo_combine(V, 0 if mode == "max" else 1, len(arrs), C.cast(ptrs, C.c_void_p), _p(w), _p(out))
    return out


def cvp_update_scalar(u1, u2, u3, a, b, c):
    out = C.c_float(0)
    sel = C.c_int(0)
    d = C.c_float(0)
    ok = lib().mo_cvp_update_scalar(u1, u2, u3, a, b, c, C.byref(out), C.byref(sel), C.byref(d))
    return bool(ok), out.value, sel.value, d.value


def inflation_fading(cfg: InflationCfg, distance: float) -> float:
    return float(lib().mo_inflation_fading(C.byref(cfg), distance))


def pose_from_position(current, nxt, normal):
    pose = np.empty(7, dtype=np.float64)
    length = lib().mo_pose_from_position(_p(_f32(current)), _p(_f32(nxt)), _p(_f32(normal)), _p(pose))
    return pose, float(length)


# ---------------------------------------------------------------------------------------------
# CPU model of the device schedule (oracle/schedule_model.cpp) -- test infrastructure.
# ---------------------------------------------------------------------------------------------
_MODEL_PATH = os.path.join(_HERE, "libmnav_model.so")
_model = None


def model_lib():
    global _model
    if _model is None:
        subprocess.check_call(["make", "-s", "-C", _HERE, "libmnav_model.so"])
        L = C.CDLL(_MODEL_PATH)
        vp, u32 = C.c_void_p, C.c_uint32
        L.sm_run.restype = u32
        L.sm_run.argtypes = [u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, u32, vp, C.c_double, C.c_double,
                             C.c_float, C.c_int, u32, vp, vp, vp, vp, vp, C.POINTER(C.c_float)]
        L.sm_acosf_ref.restype = C.c_float
        L.sm_acosf_ref.argtypes = [C.c_float]
        L.sm_cosf_ref.restype = C.c_float
        L.sm_cosf_ref.argtypes = [C.c_float]
        L.sm_sinf_ref.restype = C.c_float
        L.sm_sinf_ref.argtypes = [C.c_float]
        L.sm_backtrack.restype = C.c_int
        L.sm_backtrack.argtypes = [u32, u32, vp, vp, vp, vp, vp, vp, u32, vp, u32, C.c_double, u32, vp, vp, vp, C.c_int, vp, vp, C.POINTER(u32)]
        L.sm_infl_candidate.restype = C.c_float
        L.sm_infl_candidate.argtypes = [C.c_float] * 6 + [C.POINTER(C.c_int)]
        L.sm_run_inflation.restype = u32
        L.sm_run_inflation.argtypes = [u32, u32, u32, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_int, u32, vp, vp, vp, vp, vp, vp]
        L.tbm_run.restype = u32
        L.tbm_run.argtypes = [u32, u32, u32, vp, vp, vp, vp, vp, vp, u32, u32, vp, vp, C.c_double, C.c_double, C.c_float, C.c_int, vp, vp]
        _model = L
    return _model


def tile_batch_model(xyz, faces, edges, edge_weights, vertex_costs, seeds, targets, offset=0.3, cost_limit=1.0, tile=128,
                     band=None, jacobi=1, invalid=None, rerun_last_chunk=0):
    """The tile-batch SSSP engine (mnav_tb.h) on the CPU model (oracle/tb_model.cpp): potentials of a batch of plans."""
    faces, edges = _u32(faces), _u32(edges)
    w, vc, pos = _f32(edge_weights), _f32(vertex_costs), _f32(xyz)
    V, F, E = vc.shape[0], faces.shape[0], edges.shape[0]
    inv = None if invalid is None else _u8(invalid)
    sd, tg = _u32(seeds), _u32(targets)
    n = sd.shape[0]
    if band is None:
        fin = w[np.isfinite(w)]
        band = float(fin.mean() * np.sqrt(tile)) if fin.size else 1.0
    dist = np.empty((n, V), np.float32)
    stats = np.zeros(12, np.uint64)
    code = model_lib().tbm_run(V, F, E, _p(faces), _p(edges), _p(w), _p(vc), _p(inv), _p(pos), int(tile), n, _p(sd), _p(tg),
                               float(offset), float(cost_limit), float(band), int(jacobi) | ((int(rerun_last_chunk) & 15) << 4),
                               _p(dist), _p(stats))
    return dict(code=code, dist=dist, iterations=int(stats[0]), activations=int(stats[1]), sweeps=int(stats[2]), wakes=int(stats[3]),
                max_sweeps=int(stats[4]), tiles=int(stats[5]), slots_per_plan=int(stats[6]), items=int(stats[7]),
                blocks_total=int(stats[8]), blocks_evaluated=int(stats[9]), stale_reads=int(stats[10]), max_ghosts=int(stats[11]))


def async_tile_model(xyz, faces, edges, edge_weights, vertex_costs, seeds, targets, offset=0.3, cost_limit=1.0, tile=64, band=0.0,
                     workgroups=4, sched_seed=1, budget=50_000_000, invalid=None, mutate=0, ring_cap=None):
    """The PROTOCOL of the asynchronous tile engine (mnav_async.h: ticket queue of woken tiles) on the CPU model
    (oracle/async_model.cpp): `workgroups` virtual workgroups, interleaved pseudo-randomly (seed) at every shared-memory operation,
    on the product's own tile tables.  band = 0: no bands (every solve runs to the tile's local fixed point);
    band > 0: the plans advance in bands of that width, in potential units; band = 'tile': one tile width."""
    faces, edges = _u32(faces), _u32(edges)
    w, vc, pos = _f32(edge_weights), _f32(vertex_costs), _f32(xyz)
    V, F, E = vc.shape[0], faces.shape[0], edges.shape[0]
    inv = None if invalid is None else _u8(invalid)
    sd, tg = _u32(seeds), _u32(targets)
    n = sd.shape[0]
    if band == "tile":
        fin = w[np.isfinite(w)]
        band = float(fin.mean() * np.sqrt(tile)) if fin.size else 1.0
    dist = np.empty((n, V), np.float32)
    stats = np.zeros(12, np.uint64)
    if ring_cap is None:
        ring_cap = 64 * (V // max(int(tile) // 2, 1) + 8)            # slots per plan
    L = model_lib()
    L.asm_run.argtypes = [C.c_uint32] * 3 + [C.c_void_p] * 6 + [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                                                C.c_float, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.asm_run.restype = C.c_uint32
    code = L.asm_run(V, F, E, _p(faces), _p(edges), _p(w), _p(vc), _p(inv), _p(pos), int(tile), n, _p(sd), _p(tg), float(offset),
                     float(cost_limit), float(band), int(workgroups), int(sched_seed), int(budget), int(mutate), int(ring_cap), _p(dist), _p(stats))
    return dict(code=code, dist=dist, activations=int(stats[0]), sweeps=int(stats[1]), epochs=int(stats[2]), drops=int(stats[3]),
                finishes=int(stats[4]), scheduling_points=int(stats[5]), max_concurrent_solves=int(stats[6]), violations=int(stats[7]),
                abort=int(stats[8]), tickets=int(stats[9]), tiles=int(stats[11]))


def product_expanded_sources(dist, target: int, offset: float):
    """mnav_eval.h goal_cut / expanded_source (the device's finalize pass, path walks and lazy vector entries) on a FINAL potential:
    (mask of expanded sources, reported goal_dist, cut value)."""
    d = _f32(dist)
    ids = np.arange(d.shape[0], dtype=np.uint32)
    out = np.zeros(d.shape[0], np.uint8)
    gct = np.zeros(3, np.float32)
    L = model_lib()
    L.sm_expanded_sources.restype = None
    L.sm_expanded_sources.argtypes = [C.c_float, C.c_double, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sm_expanded_sources(float(d[target]), float(offset), int(target), d.shape[0], _p(d), _p(ids), _p(out), _p(gct))
    return out.astype(bool), float(gct[0]), float(gct[1])


def product_acosf(x) -> float:
    """mnav_eval.h acosf_ref: the device's restatement of the host libm's acosf (SteepnessLayer)."""
    return float(model_lib().sm_acosf_ref(float(x)))


def product_cosf(x) -> float:
    """mnav_eval.h cosf_ref: the device's restatement of the host libm's cosf (InflationLayer::vectorAt)."""
    return float(model_lib().sm_cosf_ref(float(x)))


def product_sinf(x) -> float:
    """mnav_eval.h sinf_ref (CVP vector map rotation, cvp_mesh_planner.cpp:234)."""
    return float(model_lib().sm_sinf_ref(float(x)))


def product_backtrack(xyz, faces, vf_ptr, vf, vecmap, seed_pos, seed_face, target_pos, target_face, step_width=0.4,
                      inflation_field=None, cap=100000):
    """mnav_walk.h (the arithmetic of the device kernel k_backtrack) on host arrays: (status, positions, faces) in the
    reference's list order (seed first), like MeshOracle.cvp_backtrack."""
    xyz, vm = _f32(xyz), _f32(vecmap)
    faces, vf_ptr, vf = _u32(faces), _u32(vf_ptr), _u32(vf)
    pos = np.empty((cap, 3), dtype=np.float32)
    face = np.empty(cap, dtype=np.uint32)
    n = C.c_uint32(0)
    d = v = cfg = None
    rep = 0
    if inflation_field is not None:
        d0, v0, c, rep = inflation_field
        d, v = _f32(d0), _f32(v0)
        cfg = np.array([c.inflation_radius, c.inscribed_radius, c.inscribed_value, c.lethal_value], dtype=np.float64)
    st = model_lib().sm_backtrack(xyz.shape[0], faces.shape[0], _p(xyz), _p(faces), _p(vf_ptr), _p(vf), _p(vm), _p(_f32(seed_pos)),
                                  int(seed_face), _p(_f32(target_pos)), int(target_face), float(step_width), cap,
                                  _p(d) if d is not None else None, _p(v) if v is not None else None,
                                  _p(cfg) if cfg is not None else None, int(rep), _p(pos), _p(face), C.byref(n))
    return st, pos[: n.value][::-1].copy(), face[: n.value][::-1].copy()


def product_inflation_update(u1, u2, a, b, c, max_distance):
    """mnav_eval.h infl_candidate (the device's waveFrontUpdate arithmetic): (value or NaN, re-queue flag)."""
    rq = C.c_int(0)
    v = model_lib().sm_infl_candidate(u1, u2, a, b, c, max_distance, C.byref(rq))
    return float(v), bool(rq.value)


def schedule_model_inflation(faces, edges, edge_dist, lethal, max_distance=0.4, delta=None, order=0, invalid=None,
                             max_steps=0, xyz=None):
    """The device's inflation wave (mnav_layer_inflation) on the CPU model: distances + queue values."""
    faces, edges, ed, le = _u32(faces), _u32(edges), _f32(edge_dist), _u8(lethal)
    V, F, E = le.shape[0], faces.shape[0], edges.shape[0]
    inv = None if invalid is None else _u8(invalid)
    dist = np.empty(V, np.float32)
    keyd = np.empty(V, np.float32)
    stats = np.zeros(8, np.uint64)
    pos = None if xyz is None else _f32(xyz)
    vec = None if xyz is None else np.zeros((V, 3), np.float32)
    has = None if xyz is None else np.zeros(V, np.uint8)
    code = model_lib().sm_run_inflation(V, F, E, _p(faces), _p(edges), _p(ed), _p(le), _p(inv), float(max_distance),
                                        float(max_distance if delta is None else delta), int(order), int(max_steps),
                                        _p(dist), _p(keyd), _p(stats), _p(pos), _p(vec), _p(has))
    return dict(code=code, dist=dist, keyd=keyd, vec=vec, has_vec=has, steps=int(stats[0]), bands=int(stats[1]), evals=int(stats[2]),
                verify_bad=int(stats[5]), verify_flags=int(stats[6]), verify_sweeps=int(stats[7]) & 0xFFFFFFFF, cuts=int(stats[7]) >> 32)


def schedule_model(planner: int, faces, edges, edge_weights, vertex_costs, seed_v, seed_d, seed_face,
                   target_v, offset=0.3, cost_limit=1.0, delta=0.3, order=0, invalid=None, max_steps=0):
    faces, edges = _u32(faces), _u32(edges)
    w, vc = _f32(edge_weights), _f32(vertex_costs)
    V, F, E = vc.shape[0], faces.shape[0], edges.shape[0]
    inv = None if invalid is None else _u8(invalid)
    sv = _u32(list(seed_v) + [NONE] * (3 - len(seed_v)))
    sd = _f32(list(seed_d) + [0.0] * (3 - len(seed_d)))
    tv = _u32(list(target_v) + [NONE] * (3 - len(target_v)))
    dist = np.empty(V, np.float32)
    pred = np.empty(V, np.uint32)
    dirn = np.zeros(V, np.float32)
    cutf = np.full(V, NONE, np.uint32)
    stats = np.zeros(8, np.uint64)
    gd = C.c_float(0)
    code = model_lib().sm_run(planner, V, F, E, _p(faces), _p(edges), _p(w), _p(vc), _p(inv), _p(sv), _p(sd),
                              int(seed_face), _p(tv), float(offset), float(cost_limit), float(delta), int(order),
                              int(max_steps), _p(dist), _p(pred), _p(dirn), _p(cutf), _p(stats), C.byref(gd))
    return dict(code=code, dist=dist, pred=pred, direction=dirn, cutface=cutf, steps=int(stats[0]),
                bands=int(stats[1]), evals=int(stats[2]), armed=int(stats[3]), shrinks=int(stats[4]),
                verify_bad=int(stats[5]), verify_flags=int(stats[6]), verify_sweeps=int(stats[7]) & 0xFFFFFFFF, cuts=int(stats[7]) >> 32, goal_dist=gd.value)

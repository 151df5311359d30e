"""Host-side mirror of the reference plugin interface for this path, for Python callers and tests.

`DijkstraMeshPlanner` / `CVPMeshPlanner` here are thin ctypes handles on the C++ adapter classes of
the same names (mesh_navigation_amd/csrc/adapter/gpu_mesh_planners.h), which implement
mbf_mesh_core::MeshPlanner (initialize / makePlan / cancel) on top of the C ABI.  Poses are
(x, y, z, qx, qy, qz, qw) rows, like geometry_msgs/Pose.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

_lib = None


def _load():
    global _lib
    if _lib is None:
        path = _build.ADAPTER_LIB
        if not os.path.exists(path):
            _build.build_adapter()
        L = C.CDLL(path)
        vp, u32, f64 = C.c_void_p, C.c_uint32, C.c_double
        L.mnav_adapter_create.restype = vp
        L.mnav_adapter_create.argtypes = [C.c_int, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, f64, f64, f64]
        L.mnav_adapter_destroy.argtypes = [vp]
        L.mnav_adapter_make_plan.restype = u32
        L.mnav_adapter_make_plan.argtypes = [vp, vp, vp, vp, u32, C.POINTER(u32), C.POINTER(f64), C.c_char_p, u32]
        L.mnav_adapter_cancel.restype = C.c_int
        L.mnav_adapter_cancel.argtypes = [vp]
        L.mnav_adapter_set_costs.argtypes = [vp, vp, vp]
        L.mnav_adapter_set_cost_version.argtypes = [vp, C.c_uint64]
        L.mnav_adapter_fetch.restype = C.c_int
        L.mnav_adapter_fetch.argtypes = [vp, C.c_int, vp]
        L.mnav_adapter_add_layer_field.argtypes = [vp, vp, vp, vp, vp, f64, f64, f64, f64, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _MeshPlanner:
    KIND = 0

    def __init__(self):
        self._h = None
        self._keep = None

    def initialize(self, name: str, mesh_map: dict, params: dict | None = None) -> bool:
        """mesh_map: dict with xyz, faces, edges, vertex_normals, face_normals, vertex_costs,
        edge_weights and optionally invalid (the arrays mesh_map::MeshMap exposes)."""
        params = params or {}

        def f32(k):
            return np.ascontiguousarray(mesh_map[k], np.float32)

        def u32(k):
            return np.ascontiguousarray(mesh_map[k], np.uint32)

        arrs = dict(xyz=f32("xyz"), faces=u32("faces"), edges=u32("edges"), vn=f32("vertex_normals"),
                    fn=f32("face_normals"), vc=f32("vertex_costs"), ew=f32("edge_weights"))
        inv = mesh_map.get("invalid")
        arrs["inv"] = None if inv is None else np.ascontiguousarray(inv, np.uint8)
        self._keep = arrs
        self._h = _load().mnav_adapter_create(
            self.KIND, arrs["xyz"].shape[0], arrs["faces"].shape[0], arrs["edges"].shape[0], _p(arrs["xyz"]),
            _p(arrs["faces"]), _p(arrs["edges"]), _p(arrs["vn"]), _p(arrs["fn"]), _p(arrs["vc"]), _p(arrs["ew"]),
            _p(arrs["inv"]), float(params.get("goal_dist_offset", 0.3)), float(params.get("cost_limit", 1.0)),
            float(params.get("step_width", 0.4)))
        return bool(self._h)

    def makePlan(self, start_pose, goal_pose, tolerance: float = 0.0):
        """Returns (code, plan (n,7), cost, message) -- the out-parameters of MeshPlanner::makePlan."""
        if not self._h:
            raise RuntimeError("planner is not initialised (no usable MI355X / HIP library): there is no CPU fallback")
        s = np.ascontiguousarray(start_pose, np.float64)
        g = np.ascontiguousarray(goal_pose, np.float64)
        cap = 1 << 16
        poses = np.empty((cap, 7), np.float64)
        n = C.c_uint32(0)
        cost = C.c_double(0)
        msg = C.create_string_buffer(512)
        code = _load().mnav_adapter_make_plan(self._h, _p(s), _p(g), _p(poses), cap, C.byref(n), C.byref(cost), msg, 512)
        return int(code), poses[: min(n.value, cap)].copy(), float(cost.value), msg.value.decode()

    def cancel(self) -> bool:
        return bool(_load().mnav_adapter_cancel(self._h)) if self._h else False

    def set_costs(self, vertex_costs, edge_weights):
        vc = np.ascontiguousarray(vertex_costs, np.float32)
        ew = np.ascontiguousarray(edge_weights, np.float32)
        _load().mnav_adapter_set_costs(self._h, _p(vc), _p(ew))

    def set_cost_version(self, version: int):
        """Change counter of the map's cost arrays (what MeshMap::layerChanged would bump); 0 = unknown -> hashing."""
        _load().mnav_adapter_set_cost_version(self._h, int(version))

    def fetch(self, what: str) -> np.ndarray:
        """V-sized results of the last plan, downloaded on demand: 'potential', 'predecessors', 'vector_map'."""
        code = {"potential": 0, "predecessors": 1, "vector_map": 4}[what]
        V = self._keep["xyz"].shape[0]
        out = np.empty((V, 3) if code == 4 else V, np.uint32 if code == 1 else np.float32)
        if _load().mnav_adapter_fetch(self._h, code, _p(out)) != 0:
            raise RuntimeError(f"cannot fetch {what}")
        return out

    def add_layer_field(self, distances, has_distance, vectors, has_vector, inscribed_radius=0.25, inflation_radius=0.4,
                        lethal_value=1.0, inscribed_value=0.99, repulsive_field=True):
        """A layer's repulsive vector field for the CVP back-tracking (InflationLayer distances_ / vector_map_)."""
        a = [np.ascontiguousarray(distances, np.float32), np.ascontiguousarray(has_distance, np.uint8),
             np.ascontiguousarray(vectors, np.float32), np.ascontiguousarray(has_vector, np.uint8)]
        _load().mnav_adapter_add_layer_field(self._h, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), float(inscribed_radius),
                                             float(inflation_radius), float(lethal_value), float(inscribed_value), int(repulsive_field))

    def close(self):
        if self._h:
            _load().mnav_adapter_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DijkstraMeshPlanner(_MeshPlanner):
    """dijkstra_mesh_planner/DijkstraMeshPlanner (dijkstra_mesh_planner.xml:1-8)."""
    KIND = 0


class CVPMeshPlanner(_MeshPlanner):
    """cvp_mesh_planner/CVPMeshPlanner (cvp_mesh_planner.xml:1-8)."""
    KIND = 1

"""ctypes loader for oracle/_ref/libmnav_ref.so: the REFERENCE's own planner / mesh_map / mesh_layers
translation units compiled unmodified (oracle/ref_build/build.sh) behind a small C harness.

TEST INFRASTRUCTURE ONLY.  It pins oracle/mnav_oracle.c (tests/test_ref_*.py) and generates the golden
fixtures; bench.py may time it as the `cpu_baseline` of kind "reference".  Never imported by the product.
`/root/reference` only has to exist when the library is (re)built; the built .so travels to the GPU box.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libmnav_ref.so")
# the same library + the product's ROS plugin package (integration/mesh_gpu_planners) linked against libmnav.so
GPU_LIB_PATH = os.path.join(_HERE, "_ref", "libmnav_ref_gpu.so")
INFLATION_TEST = os.path.join(_HERE, "_ref", "ref_inflation_test")
NONE = 0xFFFFFFFF


def reference_present() -> bool:
    return os.path.isdir(os.environ.get("MNAV_REFERENCE_DIR", "/root/reference") + "/dijkstra_mesh_planner")


def build(force: bool = False) -> str | None:
    """(Re)build from /root/reference when it is there; otherwise use the prebuilt library as is."""
    if reference_present():
        srcs = [os.path.join(_HERE, "ref_build", f) for f in ("ref_harness.cpp", "build.sh")]
        srcs.append(os.path.join(_HERE, "..", "integration", "mesh_gpu_planners", "src", "gpu_mesh_planners.cpp"))
        stale = (force or not os.path.exists(LIB_PATH)
                 or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs))
        if stale:
            subprocess.check_call(["bash", os.path.join(_HERE, "ref_build", "build.sh")],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return LIB_PATH if os.path.exists(LIB_PATH) else None


def available() -> bool:
    return build() is not None


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = build()
        if path is None:
            raise RuntimeError("oracle/_ref/libmnav_ref.so is missing and /root/reference is not here to build it")
        L = None
        if os.path.exists(GPU_LIB_PATH):                 # one process, one copy of the reference code: prefer the GPU build
            try:
                L = C.CDLL(GPU_LIB_PATH)
            except OSError:                              # libmnav.so / the HIP runtime cannot be loaded here: CPU build
                L = None
        globals()["_has_plugins"] = L is not None
        if L is None:
            L = C.CDLL(path)
        vp, u32, f32, f64, cp = C.c_void_p, C.c_uint32, C.c_float, C.c_double, C.c_char_p
        L.ref_plugin_init.restype = C.c_int
        L.ref_plugin_init.argtypes = [vp, cp, cp]
        L.ref_plugin_make_plan.restype = u32
        L.ref_plugin_make_plan.argtypes = [vp, vp, vp, vp, u32, C.POINTER(u32), C.POINTER(f64)]
        L.ref_plugin_cancel.argtypes = [vp]
        L.ref_plugin_release.argtypes = [vp]
        L.ref_new.restype = vp
        L.ref_free.argtypes = [vp]
        L.ref_param_double.argtypes = [vp, cp, f64]
        L.ref_param_bool.argtypes = [vp, cp, C.c_int]
        L.ref_param_int.argtypes = [vp, cp, C.c_int64]
        L.ref_param_string.argtypes = [vp, cp, cp]
        L.ref_param_string_array.argtypes = [vp, cp, cp]
        L.ref_set_param_bool.restype = C.c_int
        L.ref_set_param_bool.argtypes = [vp, cp, C.c_int]
        L.ref_set_param_double.restype = C.c_int
        L.ref_set_param_double.argtypes = [vp, cp, f64]
        L.ref_set_mesh.argtypes = [vp, u32, u32, vp, vp]
        L.ref_set_attr_face_normals.argtypes = [vp, u32, vp]
        L.ref_set_attr_vertex_normals.argtypes = [vp, u32, vp]
        L.ref_set_attr_edge_distances.argtypes = [vp, u32, vp]
        L.ref_set_array_layer.argtypes = [vp, cp, u32, vp, vp]
        L.ref_read_map.restype = C.c_int
        L.ref_read_map.argtypes = [vp]
        L.ref_message.restype = cp
        L.ref_message.argtypes = [vp]
        L.ref_logged_errors.restype = C.c_long
        for n in ("ref_num_vertices", "ref_num_faces", "ref_num_edges"):
            getattr(L, n).restype = u32
            getattr(L, n).argtypes = [vp]
        for n in ("ref_edges", "ref_face_vertices", "ref_vertex_costs", "ref_edge_weights", "ref_edge_distances",
                  "ref_face_normals", "ref_vertex_normals"):
            getattr(L, n).argtypes = [vp, vp]
        L.ref_edges_of_vertex.restype = u32
        L.ref_edges_of_vertex.argtypes = [vp, u32, vp, u32]
        L.ref_faces_of_vertex.restype = u32
        L.ref_faces_of_vertex.argtypes = [vp, u32, vp, u32]
        L.ref_set_invalid.argtypes = [vp, u32, vp]
        L.ref_get_invalid.argtypes = [vp, u32, vp]
        L.ref_layer_costs.restype = C.c_int
        L.ref_layer_costs.argtypes = [vp, cp, vp, vp]
        L.ref_layer_vector_at.restype = C.c_int
        L.ref_layer_vector_at.argtypes = [vp, cp, vp, vp, vp]
        L.ref_inflation_fields.restype = C.c_int
        L.ref_inflation_fields.argtypes = [vp, cp, vp, vp]
        L.ref_update_array_layer.restype = C.c_int
        L.ref_update_array_layer.argtypes = [vp, cp, u32, vp, vp, vp]
        L.ref_nearest_vertex.restype = u32
        L.ref_nearest_vertex.argtypes = [vp, vp]
        L.ref_containing_face.restype = u32
        L.ref_containing_face.argtypes = [vp, vp, f32, vp]
        L.ref_mesh_ahead.restype = C.c_int
        L.ref_mesh_ahead.argtypes = [vp, vp, C.POINTER(u32), f32]
        L.ref_dijkstra_init.restype = C.c_int
        L.ref_dijkstra_init.argtypes = [vp, cp]
        L.ref_dijkstra.restype = u32
        L.ref_dijkstra.argtypes = [vp, vp, vp, vp, u32, C.POINTER(u32)]
        L.ref_dijkstra_fields.argtypes = [vp, vp, vp, vp, vp]
        if hasattr(L, "ref_map_vector_map"):
            L.ref_map_vector_map.argtypes = [vp, vp, vp]
        if hasattr(L, "ref_pub_count"):
            L.ref_pub_count.restype = C.c_long
            L.ref_pub_count.argtypes = [C.c_char_p]
            L.ref_pub_last_path.argtypes = [C.c_char_p, vp, C.c_uint32, C.POINTER(C.c_uint32)]
            L.ref_pub_last_costs.argtypes = [C.c_char_p, vp, C.c_uint32, C.POINTER(C.c_uint32)]
        L.ref_dijkstra_make_plan.restype = u32
        L.ref_dijkstra_make_plan.argtypes = [vp, vp, vp, vp, u32, C.POINTER(u32), C.POINTER(f64)]
        L.ref_dijkstra_cancel.argtypes = [vp]
        L.ref_cvp_init.restype = C.c_int
        L.ref_cvp_init.argtypes = [vp, cp]
        L.ref_cvp.restype = u32
        L.ref_cvp.argtypes = [vp, vp, vp, vp, vp, u32, C.POINTER(u32)]
        L.ref_cvp_fields.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.ref_cvp_make_plan.restype = u32
        L.ref_cvp_make_plan.argtypes = [vp, vp, vp, vp, u32, C.POINTER(u32), C.POINTER(f64)]
        L.ref_cvp_cancel.argtypes = [vp]
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
class RefDijkstra:
    code: int
    dist: np.ndarray
    pred: np.ndarray
    path: np.ndarray
    vecmap: np.ndarray
    has_vec: np.ndarray


@dataclass
class RefCvp:
    code: int
    dist: np.ndarray
    pred: np.ndarray
    direction: np.ndarray
    cutface: np.ndarray
    vecmap: np.ndarray
    has_vec: np.ndarray
    path_pos: np.ndarray
    path_face: np.ndarray
    message: str


class RefMap:
    """A mesh_map::MeshMap of the reference, loaded through its own readMap() from an in-memory map file.

    layers: "array"  -> one harness-served layer `costs` (default layer), vertex_costs given by the caller
            "c3"     -> steepness + inflation(steepness) + avg|max combination, all the reference's own layers
            "array+inflation" -> harness-served costs/lethals + the reference's InflationLayer on them (default layer)
    """

    def __init__(self, xyz, faces, *, layers="array", vertex_costs=None, lethal=None, edge_cost_factor=0.0,
                 combination="avg", steepness_threshold=None, inflation=None, serve_normals=None,
                 serve_edge_distances=None, extra_params=None):
        L = lib()
        self.xyz = _f32(xyz).reshape(-1, 3)
        self.faces = _u32(faces).reshape(-1, 3)
        self.V, self.F = self.xyz.shape[0], self.faces.shape[0]
        self._h = L.ref_new()
        L.ref_set_mesh(self._h, self.V, self.F, _p(self.xyz), _p(self.faces))
        if serve_normals is not None:
            fn, vn = serve_normals
            L.ref_set_attr_face_normals(self._h, self.F, _p(_f32(fn)))
            L.ref_set_attr_vertex_normals(self._h, self.V, _p(_f32(vn)))
        if serve_edge_distances is not None:
            ed = _f32(serve_edge_distances)
            L.ref_set_attr_edge_distances(self._h, ed.shape[0], _p(ed))
        ns = b"mesh_map."
        L.ref_param_double(self._h, ns + b"edge_cost_factor", float(edge_cost_factor))
        if layers == "array":
            vc = np.zeros(self.V, np.float32) if vertex_costs is None else _f32(vertex_costs)
            L.ref_set_array_layer(self._h, b"costs", self.V, _p(vc), None if lethal is None else _p(_u8(lethal)))
            L.ref_param_string_array(self._h, ns + b"layers", b"costs")
            L.ref_param_string(self._h, ns + b"costs.type", b"ref_harness/ArrayLayer")
            L.ref_param_string(self._h, ns + b"default_layer", b"costs")
        elif layers == "array+observer":
            # the harness-served default layer plus the GPU planners' change observer (integration/mesh_gpu_planners:
            # mesh_gpu_planners/CostObserverLayer, the default layer as its input)
            vc = np.zeros(self.V, np.float32) if vertex_costs is None else _f32(vertex_costs)
            L.ref_set_array_layer(self._h, b"costs", self.V, _p(vc), None if lethal is None else _p(_u8(lethal)))
            L.ref_param_string_array(self._h, ns + b"layers", b"costs,gpu_cost_observer")
            L.ref_param_string(self._h, ns + b"costs.type", b"ref_harness/ArrayLayer")
            L.ref_param_string(self._h, ns + b"gpu_cost_observer.type", b"mesh_gpu_planners/CostObserverLayer")
            L.ref_param_string_array(self._h, ns + b"gpu_cost_observer.inputs", b"costs")
            L.ref_param_string(self._h, ns + b"default_layer", b"costs")
        elif layers == "array+observer_misconfigured":
            # the observer layer WITHOUT the default layer among its inputs: it would never hear of a change and must not attach
            vc = np.zeros(self.V, np.float32) if vertex_costs is None else _f32(vertex_costs)
            L.ref_set_array_layer(self._h, b"costs", self.V, _p(vc), None if lethal is None else _p(_u8(lethal)))
            L.ref_param_string_array(self._h, ns + b"layers", b"costs,gpu_cost_observer")
            L.ref_param_string(self._h, ns + b"costs.type", b"ref_harness/ArrayLayer")
            L.ref_param_string(self._h, ns + b"gpu_cost_observer.type", b"mesh_gpu_planners/CostObserverLayer")
            L.ref_param_string(self._h, ns + b"default_layer", b"costs")
        elif layers == "array+inflation":
            # a harness-served layer with lethal flags feeding the reference's InflationLayer (the default layer)
            vc = np.zeros(self.V, np.float32) if vertex_costs is None else _f32(vertex_costs)
            L.ref_set_array_layer(self._h, b"costs", self.V, _p(vc), None if lethal is None else _p(_u8(lethal)))
            L.ref_param_string_array(self._h, ns + b"layers", b"costs,inflation")
            L.ref_param_string(self._h, ns + b"costs.type", b"ref_harness/ArrayLayer")
            L.ref_param_string(self._h, ns + b"inflation.type", b"mesh_layers/InflationLayer")
            L.ref_param_string_array(self._h, ns + b"inflation.inputs", b"costs")
            L.ref_param_string(self._h, ns + b"default_layer", b"inflation")
            for k, v in (inflation or {}).items():
                L.ref_param_double(self._h, ns + b"inflation." + k.encode(), float(v))
        elif layers == "c3":
            L.ref_param_string_array(self._h, ns + b"layers", b"steepness,inflation,combined")
            L.ref_param_string(self._h, ns + b"steepness.type", b"mesh_layers/SteepnessLayer")
            L.ref_param_string(self._h, ns + b"inflation.type", b"mesh_layers/InflationLayer")
            L.ref_param_string_array(self._h, ns + b"inflation.inputs", b"steepness")
            comb = b"mesh_layers/AvgCombinationLayer" if combination == "avg" else b"mesh_layers/MaxCombinationLayer"
            L.ref_param_string(self._h, ns + b"combined.type", comb)
            L.ref_param_string_array(self._h, ns + b"combined.inputs", b"steepness,inflation")
            L.ref_param_string(self._h, ns + b"default_layer", b"combined")
            if steepness_threshold is not None:
                L.ref_param_double(self._h, ns + b"steepness.threshold", float(steepness_threshold))
            for k, v in (inflation or {}).items():
                L.ref_param_double(self._h, ns + b"inflation." + k.encode(), float(v))
        else:
            raise ValueError(layers)
        for k, v in (extra_params or {}).items():
            if isinstance(v, bool):
                L.ref_param_bool(self._h, k.encode(), int(v))
            elif isinstance(v, (int, float)):
                L.ref_param_double(self._h, k.encode(), float(v))
            else:
                L.ref_param_string(self._h, k.encode(), str(v).encode())
        if not L.ref_read_map(self._h):
            raise RuntimeError("MeshMap::readMap failed: " + L.ref_message(self._h).decode())
        self.E = L.ref_num_edges(self._h)
        assert L.ref_num_vertices(self._h) == self.V and L.ref_num_faces(self._h) == self.F
        self._dij = False
        self._cvp = False

    def close(self):
        if self._h:
            lib().ref_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- map arrays ----
    def _arr(self, fn, shape, dtype):
        out = np.zeros(shape, dtype)
        getattr(lib(), fn)(self._h, _p(out))
        return out

    def edges(self):
        return self._arr("ref_edges", (self.E, 2), np.uint32)

    def face_vertices(self):
        return self._arr("ref_face_vertices", (self.F, 3), np.uint32)

    def vertex_costs(self):
        return self._arr("ref_vertex_costs", self.V, np.float32)

    def edge_weights(self):
        return self._arr("ref_edge_weights", self.E, np.float32)

    def edge_distances(self):
        return self._arr("ref_edge_distances", self.E, np.float32)

    def face_normals(self):
        return self._arr("ref_face_normals", (self.F, 3), np.float32)

    def vertex_normals(self):
        return self._arr("ref_vertex_normals", (self.V, 3), np.float32)

    def edges_of_vertex(self, v):
        out = np.zeros(64, np.uint32)
        n = lib().ref_edges_of_vertex(self._h, int(v), _p(out), 64)
        return out[:n].copy()

    def faces_of_vertex(self, v):
        out = np.zeros(64, np.uint32)
        n = lib().ref_faces_of_vertex(self._h, int(v), _p(out), 64)
        return out[:n].copy()

    def set_invalid(self, invalid):
        lib().ref_set_invalid(self._h, self.V, _p(_u8(invalid)))

    def get_invalid(self):
        out = np.zeros(self.V, np.uint8)
        lib().ref_get_invalid(self._h, self.V, _p(out))
        return out

    def layer_costs(self, name):
        out = np.zeros(self.V, np.float32)
        le = np.zeros(self.V, np.uint8)
        if not lib().ref_layer_costs(self._h, name.encode(), _p(out), _p(le)):
            raise KeyError(name)
        return out, le

    def layer_vector_at(self, name, vs, bary):
        out = np.zeros(3, np.float32)
        rc = lib().ref_layer_vector_at(self._h, name.encode(), _p(_u32(vs)), _p(_f32(bary)), _p(out))
        return None if rc < 0 else out          # None: the layer panicked (a vertex without a stored value)

    def inflation_fields(self, name="inflation"):
        dist = np.zeros(self.V, np.float32)
        vec = np.zeros((self.V, 3), np.float32)
        if not lib().ref_inflation_fields(self._h, name.encode(), _p(dist), _p(vec)):
            raise KeyError(name)
        return dist, vec

    @staticmethod
    def gpu_plugin_cost_sync_counts():
        """(full uploads, incremental updates, signing passes) of the GPU planners' device mirrors in this process"""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib().mesh_gpu_planners_cost_sync_counts(C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def update_array_layer(self, ids, values, lethal=None, name="costs"):
        ids, values = _u32(ids), _f32(values)
        ok = lib().ref_update_array_layer(self._h, name.encode(), ids.shape[0], _p(ids), _p(values),
                                          None if lethal is None else _p(_u8(lethal)))
        assert ok

    # ---- lookups ----
    def nearest_vertex(self, p):
        return int(lib().ref_nearest_vertex(self._h, _p(_f32(p))))

    def containing_face(self, p, max_dist=0.4):
        bary = np.zeros(3, np.float32)
        f = int(lib().ref_containing_face(self._h, _p(_f32(p)), float(max_dist), _p(bary)))
        return f, bary

    def mesh_ahead(self, pos, face, step):
        p = _f32(pos).copy()
        f = C.c_uint32(int(face))
        ok = lib().ref_mesh_ahead(self._h, _p(p), C.byref(f), float(step))
        return bool(ok), p, f.value

    # ---- planners ----
    def _init_dij(self, goal_dist_offset, cost_limit):
        L = lib()
        if not self._dij:
            L.ref_param_double(self._h, b"dij.goal_dist_offset", float(goal_dist_offset))
            L.ref_param_double(self._h, b"dij.cost_limit", float(cost_limit))
            assert L.ref_dijkstra_init(self._h, b"dij")
            self._dij = (goal_dist_offset, cost_limit)
        elif self._dij != (goal_dist_offset, cost_limit):
            raise RuntimeError("one planner configuration per RefMap (goal_dist_offset has no reconfigure path)")

    def dijkstra(self, seed_pos, target_pos, goal_dist_offset=0.3, cost_limit=1.0, fields=True) -> RefDijkstra:
        """DijkstraMeshPlanner::dijkstra(seed, target, path): the wave starts at `seed_pos`'s nearest vertex."""
        self._init_dij(goal_dist_offset, cost_limit)
        path = np.zeros(self.V + 1, np.uint32)
        n = C.c_uint32(0)
        code = lib().ref_dijkstra(self._h, _p(_f32(seed_pos)), _p(_f32(target_pos)), _p(path), path.shape[0], C.byref(n))
        dist = pred = vm = hv = None
        if fields:
            dist = np.zeros(self.V, np.float32)
            pred = np.zeros(self.V, np.uint32)
            vm = np.zeros((self.V, 3), np.float32)
            hv = np.zeros(self.V, np.uint8)
            lib().ref_dijkstra_fields(self._h, _p(dist), _p(pred), _p(vm), _p(hv))
        return RefDijkstra(code, dist, pred, path[: n.value].copy(), vm, hv)

    @staticmethod
    def published_count(topic: str) -> int:
        """messages the (stub) node has published on `topic` so far; -1 = no such publisher"""
        return int(lib().ref_pub_count(topic.encode()))

    def published_path(self, topic: str = "~/path"):
        poses = np.zeros((self.V + 2, 7), np.float64)
        n = C.c_uint32(0)
        if not lib().ref_pub_last_path(topic.encode(), _p(poses), poses.shape[0], C.byref(n)):
            return None
        return poses[: n.value].copy()

    def published_costs(self, name: str = "Potential"):
        vals = np.zeros(self.V, np.float32)
        n = C.c_uint32(0)
        if not lib().ref_pub_last_costs(name.encode(), _p(vals), vals.shape[0], C.byref(n)):
            return None
        return vals[: n.value].copy()

    def set_param(self, name: str, value: float) -> bool:
        """`ros2 param set`: stores the value and fires the node's on-set-parameters callbacks"""
        if isinstance(value, bool):
            return bool(lib().ref_set_param_bool(self._h, name.encode(), int(value)))
        return bool(lib().ref_set_param_double(self._h, name.encode(), float(value)))

    def map_vector_map(self):
        """MeshMap::getVectorMap() (mesh_map.h:268): (vectors[V,3], has[V]) -- what the last planner's setVectorMap left in the map."""
        vm = np.zeros((self.V, 3), np.float32)
        hv = np.zeros(self.V, np.uint8)
        lib().ref_map_vector_map(self._h, _p(vm), _p(hv))
        return vm, hv

    def dijkstra_make_plan(self, start_pose7, goal_pose7, goal_dist_offset=0.3, cost_limit=1.0):
        self._init_dij(goal_dist_offset, cost_limit)
        poses = np.zeros((self.V + 2, 7), np.float64)
        n = C.c_uint32(0)
        cost = C.c_double(0)
        s, g = np.ascontiguousarray(start_pose7, np.float64), np.ascontiguousarray(goal_pose7, np.float64)
        code = lib().ref_dijkstra_make_plan(self._h, _p(s), _p(g), _p(poses), poses.shape[0], C.byref(n), C.byref(cost))
        return code, poses[: n.value].copy(), cost.value

    def _init_cvp(self, goal_dist_offset, cost_limit, step_width):
        L = lib()
        cfg = (goal_dist_offset, cost_limit, step_width)
        if not self._cvp:
            L.ref_param_double(self._h, b"cvp.goal_dist_offset", float(goal_dist_offset))
            L.ref_param_double(self._h, b"cvp.cost_limit", float(cost_limit))
            L.ref_param_double(self._h, b"cvp.step_width", float(step_width))
            assert L.ref_cvp_init(self._h, b"cvp")
            self._cvp = cfg
        elif self._cvp != cfg:
            raise RuntimeError("one planner configuration per RefMap")

    def cvp(self, seed_pos, target_pos, goal_dist_offset=0.3, cost_limit=1.0, step_width=0.4, cap=200000) -> RefCvp:
        """CVPMeshPlanner::waveFrontPropagation(seed, target, path, message) incl. the back-tracking."""
        self._init_cvp(goal_dist_offset, cost_limit, step_width)
        pp = np.zeros((cap, 3), np.float32)
        pf = np.zeros(cap, np.uint32)
        n = C.c_uint32(0)
        code = lib().ref_cvp(self._h, _p(_f32(seed_pos)), _p(_f32(target_pos)), _p(pp), _p(pf), cap, C.byref(n))
        dist = np.zeros(self.V, np.float32)
        pred = np.zeros(self.V, np.uint32)
        dirn = np.zeros(self.V, np.float32)
        cut = np.zeros(self.V, np.uint32)
        vm = np.zeros((self.V, 3), np.float32)
        hv = np.zeros(self.V, np.uint8)
        lib().ref_cvp_fields(self._h, _p(dist), _p(pred), _p(dirn), _p(cut), _p(vm), _p(hv))
        k = min(n.value, cap)
        return RefCvp(code, dist, pred, dirn, cut, vm, hv, pp[:k].copy(), pf[:k].copy(),
                      lib().ref_message(self._h).decode())

    def plugin_init(self, lookup_name: str, name: str, **params) -> bool:
        """Loads a MeshPlanner plugin by its pluginlib lookup name (like mbf_mesh_nav does) and initializes it on THIS
        map; `params` are set as ROS parameters `<name>.<key>` first."""
        for k, v in params.items():
            if isinstance(v, bool):
                lib().ref_param_bool(self._h, f"{name}.{k}".encode(), int(v))
            elif isinstance(v, int):
                lib().ref_param_int(self._h, f"{name}.{k}".encode(), int(v))
            else:
                lib().ref_param_double(self._h, f"{name}.{k}".encode(), float(v))
        return bool(lib().ref_plugin_init(self._h, lookup_name.encode(), name.encode()))

    def plugin_make_plan(self, start_pose7, goal_pose7):
        poses = np.zeros((200000, 7), np.float64)
        n = C.c_uint32(0)
        cost = C.c_double(0)
        s, g = np.ascontiguousarray(start_pose7, np.float64), np.ascontiguousarray(goal_pose7, np.float64)
        code = lib().ref_plugin_make_plan(self._h, _p(s), _p(g), _p(poses), poses.shape[0], C.byref(n), C.byref(cost))
        return code, poses[: n.value].copy(), cost.value, lib().ref_message(self._h).decode()

    def plugin_release(self):
        lib().ref_plugin_release(self._h)

    def cvp_make_plan(self, start_pose7, goal_pose7, goal_dist_offset=0.3, cost_limit=1.0, step_width=0.4):
        self._init_cvp(goal_dist_offset, cost_limit, step_width)
        poses = np.zeros((200000, 7), np.float64)
        n = C.c_uint32(0)
        cost = C.c_double(0)
        s, g = np.ascontiguousarray(start_pose7, np.float64), np.ascontiguousarray(goal_pose7, np.float64)
        code = lib().ref_cvp_make_plan(self._h, _p(s), _p(g), _p(poses), poses.shape[0], C.byref(n), C.byref(cost))
        return code, poses[: n.value].copy(), cost.value, lib().ref_message(self._h).decode()


def gpu_plugins_linked() -> bool:
    """True when the loaded library is the GPU build (the product's MeshPlanner plugins are registered in it)."""
    lib()
    return bool(globals().get("_has_plugins"))


def run_reference_gtests() -> tuple[int, str]:
    """Runs the reference's own mesh_layers/test/inflation_layer_test.cpp (built against the stub gtest)."""
    build()
    r = subprocess.run([INFLATION_TEST], capture_output=True, text=True)
    return r.returncode, r.stdout + r.stderr

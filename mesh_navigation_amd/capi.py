"""ctypes binding of the C ABI in include/mnav.h (libmnav.so).

Plumbing only: numpy arrays in, numpy arrays out.  There is no CPU fallback -- if the HIP
library is missing or no GPU is usable, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import weakref
from dataclasses import dataclass

import numpy as np

from . import build as _build

NONE = 0xFFFFFFFF
SUCCESS, CANCELED, INVALID_START, INVALID_GOAL, NO_PATH_FOUND, INTERNAL_ERROR = 0, 51, 52, 53, 54, 60

# every symbol include/mnav.h declares
SYMBOLS = [
    "mnav_create", "mnav_destroy", "mnav_last_error", "mnav_set_face_circulation", "mnav_upload_mesh", "mnav_upload_costs",
    "mnav_compute_edge_weights", "mnav_combine_costs", "mnav_plan_dijkstra", "mnav_plan_cvp", "mnav_plan_dijkstra_batch", "mnav_plan_cvp_batch",
    "mnav_cancel", "mnav_get_stats", "mnav_get_timing", "mnav_set_band_width", "mnav_set_dijkstra_engine", "mnav_device_output",
    "mnav_algorithmic_bytes", "mnav_shard_setup", "mnav_shard_setup_partition", "mnav_shard_walk", "mnav_device_bytes", "mnav_shard_info", "mnav_shard_begin", "mnav_shard_rounds", "mnav_shard_apply", "mnav_shard_rounds_async", "mnav_shard_apply_async",
    "mnav_shard_finalize", "mnav_update_costs", "mnav_update_edge_weights", "mnav_download_costs", "mnav_set_resident_outputs", "mnav_download_output",
    "mnav_vector_at", "mnav_backtrack_cvp", "mnav_backtrack_cvp_batch", "mnav_layer_upload", "mnav_layer_steepness", "mnav_layer_inflation", "mnav_layer_download",
    "mnav_combine_layers", "mnav_layer_stats", "mnav_layer_download_vectors", "mnav_combine_layers_update",
    "mnav_set_option", "mnav_get_option", "mnav_shard_set_goal_tie", "mnav_last_engine",
]


class _PathRows:
    """The vertex paths of a batch, row k copied out when it is asked for (valid until the context's next batch call)."""

    def __init__(self, buf, lens, cap):
        self._buf, self._lens, self._cap = buf, lens, cap

    def __len__(self):
        return self._lens.shape[0]

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        if k < 0:
            k += len(self)
        return self._buf[k, : min(int(self._lens[k]), self._cap)].copy()

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class Stats(C.Structure):
    _fields_ = [("steps", C.c_uint32), ("launches", C.c_uint32), ("bands", C.c_uint32), ("armed", C.c_uint32),
                ("goal_dist", C.c_float), ("n_plans", C.c_uint32), ("evals", C.c_uint64), ("settled", C.c_uint64),
                ("ms_init", C.c_float), ("ms_propagation", C.c_float), ("ms_vector_map", C.c_float),
                ("ms_path", C.c_float), ("ms_download", C.c_float), ("ms_total", C.c_float), ("ms_step_kernels", C.c_float),
                ("band_shrinks", C.c_uint32), ("band_cuts", C.c_uint32)]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def load(path: str | None = None):
    """Load libmnav.so (building it in-tree if the sources are newer).  Raises if unavailable."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("MNAV_LIB") or _build.LIB      # MNAV_LIB: a differently built libmnav.so (perf experiments)
    if path is None and not os.path.exists(p):
        _build.build_lib()
    if not os.path.exists(p):
        raise RuntimeError(f"HIP library {p} is missing -- run `python -m mesh_navigation_amd.build`")
    L = C.CDLL(p)
    vp, u32, f64 = C.c_void_p, C.c_uint32, C.c_double
    L.mnav_create.restype = vp
    L.mnav_create.argtypes = [C.c_int]
    L.mnav_destroy.argtypes = [vp]
    L.mnav_last_error.restype = C.c_char_p
    L.mnav_last_error.argtypes = [vp]
    L.mnav_set_face_circulation.restype = C.c_int
    L.mnav_set_face_circulation.argtypes = [vp, u32, u32, vp, vp]
    L.mnav_upload_mesh.restype = C.c_int
    L.mnav_upload_mesh.argtypes = [vp, u32, u32, u32, vp, vp, vp, vp]
    L.mnav_upload_costs.restype = C.c_int
    L.mnav_upload_costs.argtypes = [vp, vp, vp, vp]
    L.mnav_compute_edge_weights.restype = C.c_int
    L.mnav_compute_edge_weights.argtypes = [vp, vp, vp, f64, vp, vp]
    L.mnav_combine_costs.restype = C.c_int
    L.mnav_combine_costs.argtypes = [vp, C.c_int, u32, vp, vp, vp, f64, vp, vp, vp]
    L.mnav_plan_dijkstra.restype = u32
    L.mnav_plan_dijkstra.argtypes = [vp, u32, u32, f64, f64, vp, vp, vp, u32, C.POINTER(u32), vp]
    L.mnav_plan_cvp.restype = u32
    L.mnav_plan_cvp.argtypes = [vp, vp, u32, u32, f64, f64, vp, vp, vp, vp, vp]
    L.mnav_plan_dijkstra_batch.restype = u32
    L.mnav_plan_dijkstra_batch.argtypes = [vp, u32, vp, vp, f64, f64, vp, vp, vp, vp, u32, vp]
    L.mnav_plan_cvp_batch.restype = u32
    L.mnav_plan_cvp_batch.argtypes = [vp, u32, vp, vp, vp, f64, f64, vp, vp, vp, vp]
    L.mnav_cancel.argtypes = [vp]
    L.mnav_get_stats.restype = C.c_int
    L.mnav_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.mnav_get_timing.restype = C.c_int
    L.mnav_get_timing.argtypes = [vp, C.POINTER(Stats)]
    L.mnav_set_band_width.restype = C.c_int
    L.mnav_set_band_width.argtypes = [vp, C.c_float]
    L.mnav_set_dijkstra_engine.restype = C.c_int
    L.mnav_set_dijkstra_engine.argtypes = [vp, C.c_int]
    L.mnav_shard_set_goal_tie.restype = C.c_int
    L.mnav_shard_set_goal_tie.argtypes = [vp, u32]
    L.mnav_set_option.restype = C.c_int
    L.mnav_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    L.mnav_get_option.restype = C.c_double
    L.mnav_get_option.argtypes = [vp, C.c_char_p]
    L.mnav_device_output.restype = vp
    L.mnav_device_output.argtypes = [vp, u32, C.c_int]
    L.mnav_set_resident_outputs.restype = C.c_int
    L.mnav_set_resident_outputs.argtypes = [vp, C.c_int]
    L.mnav_download_output.restype = C.c_int
    L.mnav_download_output.argtypes = [vp, u32, C.c_int, vp]
    L.mnav_backtrack_cvp_batch.restype = C.c_int
    L.mnav_backtrack_cvp_batch.argtypes = [vp, u32, vp, vp, vp, vp, C.c_double, C.c_int32, u32, vp, vp, vp, vp]
    L.mnav_backtrack_cvp.restype = C.c_int
    L.mnav_backtrack_cvp.argtypes = [vp, vp, u32, vp, u32, C.c_double, C.c_int32, u32, vp, vp, vp]
    L.mnav_vector_at.restype = C.c_int
    L.mnav_vector_at.argtypes = [vp, u32, vp, vp, vp]
    L.mnav_update_costs.restype = C.c_int
    L.mnav_update_costs.argtypes = [vp, u32, vp, vp]
    L.mnav_update_edge_weights.restype = C.c_int
    L.mnav_update_edge_weights.argtypes = [vp, u32, vp, vp]
    L.mnav_download_costs.restype = C.c_int
    L.mnav_download_costs.argtypes = [vp, vp, vp]
    L.mnav_layer_upload.restype = C.c_int
    L.mnav_layer_upload.argtypes = [vp, u32, vp, vp]
    L.mnav_layer_steepness.restype = C.c_int
    L.mnav_layer_steepness.argtypes = [vp, u32, f64]
    L.mnav_layer_inflation.restype = C.c_int
    L.mnav_layer_inflation.argtypes = [vp, u32, u32, f64, f64, f64, f64, f64, vp]
    L.mnav_layer_download.restype = C.c_int
    L.mnav_layer_download.argtypes = [vp, u32, vp, vp, vp]
    L.mnav_combine_layers_update.restype = C.c_int
    L.mnav_combine_layers_update.argtypes = [vp, C.c_int, u32, vp, vp, u32, vp]
    L.mnav_layer_download_vectors.restype = C.c_int
    L.mnav_layer_download_vectors.argtypes = [vp, u32, vp, vp]
    L.mnav_combine_layers.restype = C.c_int
    L.mnav_combine_layers.argtypes = [vp, C.c_int, u32, vp, vp, f64, vp]
    L.mnav_layer_stats.restype = C.c_int
    L.mnav_layer_stats.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(u32), C.POINTER(C.c_float)]
    L.mnav_shard_setup.restype = C.c_int
    L.mnav_shard_setup.argtypes = [vp, u32, u32]
    L.mnav_shard_setup_partition.restype = C.c_int
    L.mnav_shard_setup_partition.argtypes = [vp, u32, C.POINTER(u32), C.POINTER(C.c_uint8)]
    L.mnav_shard_walk.restype = C.c_int
    L.mnav_shard_walk.argtypes = [vp, u32, u32, u32, C.POINTER(u32)]
    L.mnav_device_bytes.restype = C.c_uint64
    L.mnav_device_bytes.argtypes = [vp]
    L.mnav_shard_info.restype = C.c_int
    L.mnav_shard_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.mnav_shard_begin.restype = C.c_int
    L.mnav_shard_begin.argtypes = [vp, u32, u32, f64, f64]
    L.mnav_shard_rounds.restype = C.c_int
    L.mnav_shard_rounds.argtypes = [vp, u32, vp]
    L.mnav_shard_rounds_async.restype = C.c_int
    L.mnav_shard_rounds_async.argtypes = [vp, u32, vp, vp]
    L.mnav_shard_apply_async.restype = C.c_int
    L.mnav_shard_apply_async.argtypes = [vp, vp, vp, vp]
    L.mnav_shard_apply.restype = C.c_int
    L.mnav_shard_apply.argtypes = [vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.mnav_shard_finalize.restype = C.c_int
    L.mnav_shard_finalize.argtypes = [vp, vp, vp]
    L.mnav_algorithmic_bytes.restype = C.c_uint64
    L.mnav_algorithmic_bytes.argtypes = [vp]
    L.mnav_last_engine.restype = C.c_int
    L.mnav_last_engine.argtypes = [vp]
    if path is None:
        _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


@dataclass
class DijkstraOut:
    code: int
    dist: np.ndarray | None
    pred: np.ndarray | None
    path: np.ndarray          # dijkstra() list order: seed first ... pred[target]
    vecmap: np.ndarray | None
    stats: dict


@dataclass
class CvpOut:
    code: int
    dist: np.ndarray | None
    pred: np.ndarray | None
    direction: np.ndarray | None
    cutface: np.ndarray | None
    vecmap: np.ndarray | None
    stats: dict


class MnavContext:
    """One device context = one planner instance's device state (mnav_ctx)."""

    def __init__(self, device: int = 0):
        self._L = load()
        self._h = self._L.mnav_create(int(device))
        if not self._h:
            raise RuntimeError("mnav_create failed: no usable MI355X/HIP device (there is no CPU fallback)")
        self.V = self.F = self.E = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.mnav_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _err(self) -> str:
        return (self._L.mnav_last_error(self._h) or b"").decode()

    def upload_mesh(self, xyz, faces, edges, vertex_normals=None, face_circulation=None):
        """face_circulation: optional (vf_ptr[V+1], vf[3F]) = getFacesOfVertex rows of the caller's half-edge
        mesh; by default the library replays the half-edge construction over `faces` itself."""
        xyz, faces, edges = _f32(xyz), _u32(faces), _u32(edges)
        vn = None if vertex_normals is None else _f32(vertex_normals)
        self.V, self.F, self.E = xyz.shape[0], faces.shape[0], edges.shape[0]
        if face_circulation is not None:
            ptr, vf = _u32(face_circulation[0]), _u32(face_circulation[1])
            self._L.mnav_set_face_circulation(self._h, self.V, self.F, _p(ptr), _p(vf))
        else:
            self._L.mnav_set_face_circulation(self._h, self.V, self.F, None, None)
        rc = self._L.mnav_upload_mesh(self._h, self.V, self.F, self.E, _p(xyz), _p(faces), _p(edges), _p(vn))
        if rc != 0:
            raise RuntimeError(f"mnav_upload_mesh failed ({rc}): {self._err()}")

    def upload_costs(self, vertex_costs, edge_weights, invalid=None):
        vc, w = _f32(vertex_costs), _f32(edge_weights)
        inv = None if invalid is None else np.ascontiguousarray(invalid, dtype=np.uint8)
        rc = self._L.mnav_upload_costs(self._h, _p(vc), _p(w), _p(inv))
        if rc != 0:
            raise RuntimeError(f"mnav_upload_costs failed ({rc}): {self._err()}")

    def compute_edge_weights(self, vertex_costs, edge_distances, edge_cost_factor: float, invalid=None) -> np.ndarray:
        vc, ed = _f32(vertex_costs), _f32(edge_distances)
        inv = None if invalid is None else np.ascontiguousarray(invalid, dtype=np.uint8)
        out = np.empty(self.E, dtype=np.float32)
        rc = self._L.mnav_compute_edge_weights(self._h, _p(vc), _p(ed), float(edge_cost_factor), _p(inv), _p(out))
        if rc != 0:
            raise RuntimeError(f"mnav_compute_edge_weights failed ({rc}): {self._err()}")
        return out

    def combine_costs(self, layers, weights, edge_distances, edge_cost_factor: float, mode: str = "avg", invalid=None):
        """Max/Avg combination of dense V-sized cost layers + edge weights, on the device.
        Returns (vertex_costs, edge_weights); both also become the context's planning inputs."""
        arrs = [_f32(a) for a in layers]
        ptrs = (C.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
        w = _f32(weights if weights is not None else np.ones(len(arrs), np.float32))
        ed = _f32(edge_distances)
        inv = None if invalid is None else np.ascontiguousarray(invalid, dtype=np.uint8)
        vc = np.empty(self.V, dtype=np.float32)
        ew = np.empty(self.E, dtype=np.float32)
        rc = self._L.mnav_combine_costs(self._h, 0 if mode == "max" else 1, len(arrs), C.cast(ptrs, C.c_void_p), _p(w), _p(ed),
                                        float(edge_cost_factor), _p(inv), _p(vc), _p(ew))
        if rc != 0:
            raise RuntimeError(f"mnav_combine_costs failed ({rc}): {self._err()}")
        return vc, ew

    def set_band_width(self, delta: float):
        self._L.mnav_set_band_width(self._h, float(delta))

    def set_dijkstra_engine(self, engine: str):
        """'auto' (default), 'tiled', 'band', 'tile_batch' (large batches: one plan per lane)
        or 'async' (the tiles without rounds, mnav_async.h: what 'auto' takes for single plans and small batches)."""
        if self._L.mnav_set_dijkstra_engine(self._h, {"tiled": 0, "band": 1, "auto": 3, "tile_batch": 5, "async": 6}[engine]) != 0:
            raise ValueError(f"engine {engine!r} refused")

    def set_option(self, name: str, value=None):
        """Tuning / debug option by name (csrc/mnav_options.h); None restores the built-in default.  The library reads the
        environment once, in mnav_create: after that this is the only way to change an option."""
        if self._L.mnav_set_option(self._h, name.encode(), float("nan") if value is None else float(value)) != 0:
            raise ValueError(f"mnav_set_option: {self._err()}")

    def get_option(self, name: str):
        v = self._L.mnav_get_option(self._h, name.encode())
        return None if v != v else v

    def set_resident_outputs(self, on: bool = True):
        self._L.mnav_set_resident_outputs(self._h, 1 if on else 0)

    def download_output(self, what: str, slot: int = 0) -> np.ndarray:
        code = {"dist": 0, "pred": 1, "direction": 2, "cutface": 3, "vecmap": 4, "popped": 5}[what]
        out = np.empty((self.V, 3) if code == 4 else self.V, np.uint32 if code in (1, 3) else np.float32)
        if self._L.mnav_download_output(self._h, int(slot), code, _p(out)) != 0:
            raise RuntimeError(f"mnav_download_output failed: {self._err()}")
        return out

    def vector_at(self, vs, bary, slot: int = 0):
        out = np.zeros(3, np.float32)
        rc = self._L.mnav_vector_at(self._h, int(slot), _p(_u32(vs)), _p(_f32(bary)), _p(out))
        if rc < 0:
            raise RuntimeError(f"mnav_vector_at failed: {self._err()}")
        return out if rc == 1 else None

    def backtrack_cvp_batch(self, seed_pos, seed_faces, target_pos, target_faces, step_width: float = 0.4, inflation_layer: int = -1,
                            cap: int = 4096):
        """CVPMeshPlanner's back-tracking (cvp_mesh_planner.cpp:920-951) on the resident vector maps of the last CVP call:
        list of (status, positions[n,3], faces[n]) per plan, reference list order (seed first); status 1 = reached."""
        sp, tp = _f32(seed_pos).reshape(-1, 3), _f32(target_pos).reshape(-1, 3)
        sf, tf = _u32(seed_faces).reshape(-1), _u32(target_faces).reshape(-1)
        n = sf.shape[0]
        pos = np.empty((n, cap, 3), np.float32)
        face = np.empty((n, cap), np.uint32)
        cnt = np.zeros(n, np.uint32)
        st = np.zeros(n, np.int32)
        rc = self._L.mnav_backtrack_cvp_batch(self._h, n, _p(sp), _p(sf), _p(tp), _p(tf), float(step_width), int(inflation_layer), int(cap),
                                              _p(pos), _p(face), _p(cnt), _p(st))
        if rc != 0:
            raise RuntimeError(f"mnav_backtrack_cvp_batch failed ({rc}): {self._err()}")
        return [(int(st[i]), pos[i, : cnt[i]].copy(), face[i, : cnt[i]].copy()) for i in range(n)]

    def backtrack_cvp(self, seed_pos, seed_face, target_pos, target_face, step_width: float = 0.4, inflation_layer: int = -1, cap: int = 4096):
        return self.backtrack_cvp_batch([seed_pos], [seed_face], [target_pos], [target_face], step_width, inflation_layer, cap)[0]

    def update_costs(self, vertex_ids, values):
        """Incremental cost change (layerChanged + updateEdgeWeights(changed)) on the device."""
        ids, vals = _u32(vertex_ids), _f32(values)
        rc = self._L.mnav_update_costs(self._h, ids.shape[0], _p(ids), _p(vals))
        if rc != 0:
            raise RuntimeError(f"mnav_update_costs failed ({rc}): {self._err()}")

    def update_edge_weights(self, edge_ids, values):
        """The caller's own incremental edge weights (MeshMap::updateEdgeWeights ran on the host): scatter into the resident weights."""
        ids, vals = _u32(edge_ids), _f32(values)
        rc = self._L.mnav_update_edge_weights(self._h, ids.shape[0], _p(ids), _p(vals))
        if rc != 0:
            raise RuntimeError(f"mnav_update_edge_weights failed ({rc}): {self._err()}")

    def download_costs(self):
        vc = np.empty(self.V, np.float32)
        w = np.empty(self.E, np.float32)
        if self._L.mnav_download_costs(self._h, _p(vc), _p(w)) != 0:
            raise RuntimeError(f"mnav_download_costs failed: {self._err()}")
        return vc, w

    # ---- cost layers on the device (mesh_layers: Steepness, Inflation, Combination) ----
    def layer_upload(self, layer: int, costs, lethal=None):
        c = np.ascontiguousarray(costs, np.float32)
        le = None if lethal is None else np.ascontiguousarray(lethal, np.uint8)
        if self._L.mnav_layer_upload(self._h, int(layer), _p(c), None if le is None else _p(le)) != 0:
            raise RuntimeError(f"mnav_layer_upload failed: {self._err()}")

    def layer_steepness(self, layer: int, threshold: float = 0.3):
        if self._L.mnav_layer_steepness(self._h, int(layer), float(threshold)) != 0:
            raise RuntimeError(f"mnav_layer_steepness failed: {self._err()}")

    def layer_inflation(self, layer: int, input_layer: int, inflation_radius=0.4, inscribed_radius=0.25, inscribed_value=0.99,
                        lethal_value=1.0, cost_scaling_factor=1.0, invalid=None) -> dict:
        """InflationLayer defaults: inflation_layer.cpp:676-712.  Returns the wave's counters."""
        inv = None if invalid is None else np.ascontiguousarray(invalid, np.uint8)
        rc = self._L.mnav_layer_inflation(self._h, int(layer), int(input_layer), float(inflation_radius), float(inscribed_radius),
                                          float(inscribed_value), float(lethal_value), float(cost_scaling_factor),
                                          None if inv is None else _p(inv))
        if rc != 0:
            raise RuntimeError(f"mnav_layer_inflation failed: {self._err()}")
        a, b, e, ms, vs, mw = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_float(), C.c_uint32(), C.c_float()
        self._L.mnav_layer_stats(self._h, C.byref(a), C.byref(b), C.byref(e), C.byref(ms), C.byref(vs), C.byref(mw))
        return dict(steps=a.value, bands=b.value, evals=e.value, ms=ms.value, ms_wave=mw.value, verify_sweeps=vs.value)

    def layer_download(self, layer: int, distances: bool = False):
        c = np.empty(self.V, np.float32)
        le = np.empty(self.V, np.uint8)
        d = np.empty(self.V, np.float32) if distances else None
        if self._L.mnav_layer_download(self._h, int(layer), _p(c), _p(le), None if d is None else _p(d)) != 0:
            raise RuntimeError(f"mnav_layer_download failed: {self._err()}")
        return (c, le, d) if distances else (c, le)

    def layer_vectors(self, layer: int):
        """vector_map_ of an inflation layer: (V, 3) floats and the has-entry flags"""
        vec = np.empty((self.V, 3), np.float32)
        has = np.empty(self.V, np.uint8)
        if self._L.mnav_layer_download_vectors(self._h, int(layer), _p(vec), _p(has)) != 0:
            raise RuntimeError(f"mnav_layer_download_vectors failed: {self._err()}")
        return vec, has

    def combine_layers(self, layers, weights=None, mode: str = "avg", edge_cost_factor: float = 1.0, invalid=None):
        ls = np.ascontiguousarray(layers, np.uint32)
        w = np.ascontiguousarray(weights if weights is not None else [1.0] * len(ls), np.float32)
        inv = None if invalid is None else np.ascontiguousarray(invalid, np.uint8)
        if self._L.mnav_combine_layers(self._h, 0 if mode == "max" else 1, len(ls), _p(ls), _p(w), float(edge_cost_factor),
                                       None if inv is None else _p(inv)) != 0:
            raise RuntimeError(f"mnav_combine_layers failed: {self._err()}")

    def combine_layers_update(self, layers, ids, weights=None, mode: str = "avg"):
        ls = np.ascontiguousarray(layers, np.uint32)
        w = np.ascontiguousarray(weights if weights is not None else [1.0] * len(ls), np.float32)
        ii = np.ascontiguousarray(ids, np.uint32)
        if self._L.mnav_combine_layers_update(self._h, 0 if mode == "max" else 1, len(ls), _p(ls), _p(w), ii.shape[0], _p(ii)) != 0:
            raise RuntimeError(f"mnav_combine_layers_update failed: {self._err()}")

    # ---- one plan over several GPUs (mesh_navigation_amd/sharded.py drives these) ----
    def shard_setup(self, rank: int, world: int) -> int:
        n = self._L.mnav_shard_setup(self._h, int(rank), int(world))
        if n < 0:
            raise RuntimeError(f"mnav_shard_setup failed: {self._err()}")
        return n

    def shard_setup_partition(self, exchange_vertex: np.ndarray, owned: np.ndarray) -> int:
        """This context holds ONE PART of a partitioned mesh (include/mnav.h): `exchange_vertex[i]` = local id of interface
        vertex i or 0xFFFFFFFF, `owned[v]` = 1 for the vertices this process owns.  Returns the floats in the exchange buffer."""
        ex = np.ascontiguousarray(exchange_vertex, np.uint32)
        ow = np.ascontiguousarray(owned, np.uint8)
        if ow.shape[0] != self.V:
            raise ValueError("owned: one byte per local vertex")
        n = self._L.mnav_shard_setup_partition(self._h, ex.shape[0], ex.ctypes.data_as(C.POINTER(C.c_uint32)), ow.ctypes.data_as(C.POINTER(C.c_uint8)))
        if n < 0:
            raise RuntimeError(f"mnav_shard_setup_partition failed: {self._err()}")
        return n

    def shard_walk(self, start: int, seed: int, cap: int = 4096) -> np.ndarray:
        """One path segment inside this part (include/mnav.h): [hops, stop vertex, status, local ids...]."""
        out = np.zeros(cap + 3, np.uint32)
        if self._L.mnav_shard_walk(self._h, int(start), int(seed), int(cap), out.ctypes.data_as(C.POINTER(C.c_uint32))) != 0:
            raise RuntimeError(f"mnav_shard_walk failed: {self._err()}")
        return out

    def device_bytes(self) -> int:
        return int(self._L.mnav_device_bytes(self._h))

    def shard_info(self) -> dict:
        a, b, c, d = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        if self._L.mnav_shard_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)) != 0:
            raise RuntimeError("mnav_shard_setup has not been called")
        return dict(t_lo=a.value, t_hi=b.value, ntiles=c.value, n_exchange=d.value)

    def shard_set_goal_tie(self, tie_id: int):
        self._L.mnav_shard_set_goal_tie(self._h, int(tie_id))

    def shard_begin(self, seed: int, target: int, goal_dist_offset: float = 0.3, cost_limit: float = 1.0):
        if self._L.mnav_shard_begin(self._h, int(seed), int(target), float(goal_dist_offset), float(cost_limit)) != 0:
            raise RuntimeError(f"mnav_shard_begin failed: {self._err()}")

    def shard_rounds(self, rounds: int, buf_ptr: int) -> int:
        rc = self._L.mnav_shard_rounds(self._h, int(rounds), C.c_void_p(buf_ptr))
        if rc < 0:
            raise RuntimeError(f"mnav_shard_rounds failed: {self._err()}")
        return rc

    def shard_apply(self, buf_ptr: int):
        lm, td = C.c_float(), C.c_float()
        if self._L.mnav_shard_apply(self._h, C.c_void_p(buf_ptr), C.byref(lm), C.byref(td)) != 0:
            raise RuntimeError(f"mnav_shard_apply failed: {self._err()}")
        return lm.value, td.value

    def shard_rounds_async(self, rounds: int, buf_ptr: int, stream: int):
        if self._L.mnav_shard_rounds_async(self._h, int(rounds), C.c_void_p(buf_ptr), C.c_void_p(stream)) != 0:
            raise RuntimeError(f"mnav_shard_rounds_async failed: {self._err()}")

    def shard_apply_async(self, buf_ptr: int, ctl_ptr: int, stream: int):
        if self._L.mnav_shard_apply_async(self._h, C.c_void_p(buf_ptr), C.c_void_p(ctl_ptr), C.c_void_p(stream)) != 0:
            raise RuntimeError(f"mnav_shard_apply_async failed: {self._err()}")

    def shard_finalize(self, dist_ptr: int, pred_ptr: int):
        rc = self._L.mnav_shard_finalize(self._h, C.c_void_p(dist_ptr), C.c_void_p(pred_ptr))
        if rc != 0:
            raise RuntimeError(f"mnav_shard_finalize failed ({rc}): {self._err()}")

    def stats(self) -> dict:
        s = Stats()
        self._L.mnav_get_stats(self._h, C.byref(s))
        d = s.as_dict()
        d["algorithmic_bytes"] = int(self._L.mnav_algorithmic_bytes(self._h))
        return d

    def last_engine(self) -> str:
        """Engine / kernel of the last Dijkstra call, by name (mnav_last_engine)."""
        e = int(self._L.mnav_last_engine(self._h))
        return {0: "k_tile_round (tile rounds)", 1: "k_step (band steps)", 5: "k_tb_solve_q (tile-batch engine, quarters of a wave, distances in LDS)",
                21: "k_tbv_solve (tile-batch engine, one wave per tile, distances in VGPRs)", 6: "k_plan_async (asynchronous tile engine)"}.get(e, "none")

    def timing(self) -> dict:
        """Event timings of the last call; never triggers the (lazy) settled-vertex count of a tile-batch call."""
        s = Stats()
        self._L.mnav_get_timing(self._h, C.byref(s))
        return s.as_dict()

    def cancel(self):
        self._L.mnav_cancel(self._h)

    def device_output(self, slot: int, what: int) -> int:
        return int(self._L.mnav_device_output(self._h, slot, what) or 0)

    def plan_dijkstra(self, seed_vertex: int, target_vertex: int, goal_dist_offset: float = 0.3,
                      cost_limit: float = 1.0, want_fields: bool = True, want_vecmap: bool = False) -> DijkstraOut:
        V = self.V
        dist = np.empty(V, np.float32) if want_fields else None
        pred = np.empty(V, np.uint32) if want_fields else None
        vm = np.empty((V, 3), np.float32) if want_vecmap else None
        path = np.empty(max(V, 1), np.uint32)
        n = C.c_uint32(0)
        code = self._L.mnav_plan_dijkstra(self._h, int(seed_vertex) & 0xFFFFFFFF, int(target_vertex) & 0xFFFFFFFF,
                                          float(goal_dist_offset), float(cost_limit), _p(dist), _p(pred), _p(path),
                                          path.shape[0], C.byref(n), _p(vm))
        if code == INTERNAL_ERROR:
            raise RuntimeError(f"mnav_plan_dijkstra internal error: {self._err()}")
        return DijkstraOut(code, dist, pred, path[: n.value].copy(), vm, self.stats())

    def plan_dijkstra_batch(self, seeds, targets, goal_dist_offset: float = 0.3, cost_limit: float = 1.0,
                            want_fields: bool = False, path_cap: int | None = None, want_stats: bool = True):
        seeds, targets = _u32(seeds), _u32(targets)
        n = seeds.shape[0]
        V = self.V
        cap = int(path_cap if path_cap is not None else V)
        codes = np.empty(n, np.uint32)
        dist = np.empty((n, V), np.float32) if want_fields else None
        pred = np.empty((n, V), np.uint32) if want_fields else None
        # the rows of the path buffer are only touched where a path lands: keep the (mostly untouched) buffers between calls --
        # a fresh 335 MB buffer per call costs one page fault per row (12 ms per 5120-plan batch).  A small pool, because the
        # caller usually still holds the previous call's result (which refers to its buffer) while the next call runs.
        key = (n, max(cap, 1))
        if getattr(self, "_path_buf_key", None) != key:
            self._path_pool, self._path_buf_key = [], key
        # a pooled buffer is free when no result that was handed out still refers to it: every _PathRows takes a LEASE on its
        # buffer (a weak reference to the rows object kept next to the buffer), released when the rows object dies
        paths = None
        for entry in self._path_pool:
            entry[1] = [r for r in entry[1] if r() is not None]
            if not entry[1]:
                paths = entry[0]
                lease = entry[1]
                break
        if paths is None:
            paths = np.empty(key, np.uint32)
            lease = []
            if len(self._path_pool) < 3:
                self._path_pool.append([paths, lease])
        lens = np.zeros(n, np.uint32)
        rc = self._L.mnav_plan_dijkstra_batch(self._h, n, _p(seeds), _p(targets), float(goal_dist_offset),
                                              float(cost_limit), _p(codes), _p(dist), _p(pred), _p(paths), cap, _p(lens))
        if rc == INTERNAL_ERROR:
            raise RuntimeError(f"mnav_plan_dijkstra_batch internal error: {self._err()}")
        rows = _PathRows(paths, lens, cap)
        lease.append(weakref.ref(rows))
        return dict(rc=rc, codes=codes, dist=dist, pred=pred, paths=rows, path_len=lens,
                    stats=self.stats() if want_stats else self.timing())

    def plan_cvp_batch(self, seed_pos, seed_faces, target_faces, goal_dist_offset: float = 0.3, cost_limit: float = 1.0,
                       want_fields: bool = False, want_vecmap: bool = False):
        sp = _f32(seed_pos).reshape(-1, 3)
        sf, tf = _u32(seed_faces), _u32(target_faces)
        n, V = sf.shape[0], self.V
        codes = np.empty(n, np.uint32)
        dist = np.empty((n, V), np.float32) if want_fields else None
        pred = np.empty((n, V), np.uint32) if want_fields else None
        vm = np.empty((n, V, 3), np.float32) if want_vecmap else None
        rc = self._L.mnav_plan_cvp_batch(self._h, n, _p(sp), _p(sf), _p(tf), float(goal_dist_offset), float(cost_limit),
                                         _p(codes), _p(dist), _p(pred), _p(vm))
        if rc == INTERNAL_ERROR:
            raise RuntimeError(f"mnav_plan_cvp_batch internal error: {self._err()}")
        return dict(rc=rc, codes=codes, dist=dist, pred=pred, vecmap=vm, stats=self.stats())

    def plan_cvp(self, seed_pos, seed_face: int, target_face: int, goal_dist_offset: float = 0.3,
                 cost_limit: float = 1.0, want_fields: bool = True, want_vecmap: bool = True) -> CvpOut:
        V = self.V
        sp = _f32(seed_pos)
        dist = np.empty(V, np.float32) if want_fields else None
        pred = np.empty(V, np.uint32) if want_fields else None
        dirn = np.empty(V, np.float32) if want_fields else None
        cutf = np.empty(V, np.uint32) if want_fields else None
        vm = np.empty((V, 3), np.float32) if want_vecmap else None
        code = self._L.mnav_plan_cvp(self._h, _p(sp), int(seed_face) & 0xFFFFFFFF, int(target_face) & 0xFFFFFFFF,
                                     float(goal_dist_offset), float(cost_limit), _p(dist), _p(pred), _p(dirn),
                                     _p(cutf), _p(vm))
        if code == INTERNAL_ERROR:
            raise RuntimeError(f"mnav_plan_cvp internal error: {self._err()}")
        return CvpOut(code, dist, pred, dirn, cutf, vm, self.stats())

"""In-tree build of the HIP library (gfx950 only).  `python -m mesh_navigation_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libmnav.so")
ADAPTER_LIB = os.path.join(_HERE, "libmnav_adapter.so")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fno-strict-aliasing",        # LDS images are staged as 16-byte vectors and read as u16 / f32
               "-Wall", "-Wno-unused-function"]


def _stale(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X library cannot be built")
    return exe


def build_lib(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "mnav.h"))
    if force or _stale(LIB, srcs):
        cmd = [hipcc(), *HIPCC_FLAGS, "-o", LIB, os.path.join(CSRC, "mnav.hip")]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


def build_adapter(force: bool = False, verbose: bool = False) -> str:
    """C++ MeshPlanner-shaped adapter + its C test harness (links against libmnav.so)."""
    adir = os.path.join(CSRC, "adapter")
    if not os.path.isdir(adir):
        return ""
    srcs = [os.path.join(adir, f) for f in os.listdir(adir)]
    cpps = [s for s in srcs if s.endswith(".cpp")]
    if not cpps:
        return ""
    build_lib(force=False, verbose=verbose)
    shared = [os.path.join(_HERE, "..", "include", f) for f in ("mnav.h", "mnav_planner_host.hpp")]
    if force or _stale(ADAPTER_LIB, srcs + shared + [LIB]):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
               "-I", os.path.join(_HERE, "..", "include"), "-I", adir, "-o", ADAPTER_LIB, *cpps,
               "-L", _HERE, "-lmnav", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return ADAPTER_LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
    a = build_adapter(force="--force" in sys.argv, verbose=True)
    if a:
        print(a)

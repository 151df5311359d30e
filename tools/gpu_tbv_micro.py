"""The register-resident Gauss-Seidel sweep routine (mnav_tbv.h, tbv_sweeps) on its own: (1) against a numpy restatement of the
same stream on random images, bit for bit; (2) timing: cycles per block and SIMD at full residency.

    python tools/gpu_tbv_micro.py
"""
import ctypes as C
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from mesh_navigation_amd import capi  # noqa: E402

T, INF = 120, np.uint32(0x7F800000)


def make_stream(rng, nch):
    """4 orders x nch chunks x 4 blocks in the V layout; returns (stream words, list of orders, each a list of (tgt, srcs[6], w[6]))"""
    words = np.zeros(4 * nch * 64, np.uint32)
    orders = []
    for o in range(4):
        blocks = []
        perm = rng.permutation(T)
        for c in range(nch):
            for j in range(4):
                b = c * 4 + j
                tgt = int(perm[b % T])
                n = int(rng.integers(2, 7))
                srcs = [int(x) for x in rng.integers(0, T, n)] + [tgt] * (6 - n)
                w = np.concatenate([rng.uniform(0.05, 0.3, n).astype(np.float32), np.full(6 - n, np.inf, np.float32)])
                d = np.zeros(16, np.uint32)
                S = [0x2000 | (64 + x) for x in srcs]                          # window rows: 64 ghosts first, owned row r at 64 + r
                d[0] = (0xA000 | (64 + tgt)) | (S[0] << 16); d[1] = S[1] | (S[2] << 16); d[2] = S[3] | (S[4] << 16); d[3] = S[5]
                d[8:14] = w.view(np.uint32)
                base = (o * nch + c) * 64
                for q in range(16):
                    words[base + 4 * q + j] = d[q]
                blocks.append((tgt, srcs, w))
        orders.append(blocks)
    return words, orders


def emulate(img, orders, first, cap):
    """img [T][64] uint32 (float bits): sweeps until one changes nothing in any lane; returns sweeps"""
    f = img.view(np.float32)
    sweeps = 0
    o = first
    while True:
        changed = False
        for tgt, srcs, w in orders[o]:
            cand = np.full(64, np.inf, np.float32)
            for k in range(6):
                cand = np.minimum(cand, (f[srcs[k]] + w[k]).astype(np.float32))
            low = cand < f[tgt]
            if low.any():
                changed = True
                f[tgt] = np.where(low, cand, f[tgt])
        sweeps += 1
        o = (o + 1) & 3
        if not changed or sweeps >= cap:
            return sweeps


def main():
    L = capi.load()
    fn = L.mnav_debug_tbv_sweeps
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    rng = np.random.default_rng(7)
    out = {}
    # ---- (1) parity
    nch, waves = 31, 24
    words, orders = make_stream(rng, nch)
    img = rng.uniform(0.0, 6.0, (waves, T, 64)).astype(np.float32)
    img[rng.random(img.shape) < 0.5] = np.inf
    img[:, 3, :] = 0.0
    ref = img.copy()
    res = np.zeros((waves, 2), np.uint32)
    ms = C.c_float()
    first = 2
    rc = fn(words.ctypes.data, nch, first, 64, 0, 1, waves, img.ctypes.data, res.ctypes.data, C.byref(ms))
    assert rc == 0, rc
    ref_sweeps = [emulate(ref[w].view(np.uint32), orders, first, 64) for w in range(waves)]
    out["parity"] = dict(images_equal=bool(np.array_equal(img.view(np.uint32), ref.view(np.uint32))), sweeps_gpu=res[:, 0].tolist(), sweeps_ref=ref_sweeps,
                         overrun=res[:, 1].tolist(), differing=int((img.view(np.uint32) != ref.view(np.uint32)).sum()))
    # ---- (2) timing: every wave runs `reps` x `cap` forced sweeps
    for per_simd, waves in ((1, 1024), (2, 2048), (4, 8192)):
        cap, reps = 40, 4
        img2 = np.full((waves, T, 64), 1.0, np.float32)
        res2 = np.zeros((waves, 2), np.uint32)
        fn(words.ctypes.data, nch, 0, cap, 1, 1, waves, img2.ctypes.data, res2.ctypes.data, C.byref(ms))      # warm-up
        rc = fn(words.ctypes.data, nch, 0, cap, 1, reps, waves, img2.ctypes.data, res2.ctypes.data, C.byref(ms))
        blocks = waves * reps * cap * nch * 4
        clk = 2.4e9
        out[f"timing_{waves}_waves"] = dict(ms=ms.value, blocks=blocks, ns_per_block_per_simd=ms.value * 1e6 / (blocks / 1024.0),
                                            cycles_per_block_per_simd_at_2p4GHz=ms.value * 1e-3 * clk / (blocks / 1024.0), sweeps_ok=bool((res2[:, 0] == reps * cap).all()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()

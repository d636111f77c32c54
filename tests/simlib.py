"""TEST INFRASTRUCTURE: builds and loads the CPU SIMT executor build of the kernel headers (tests/sim)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SIM = os.path.join(HERE, "sim")
OUT = os.path.join(SIM, "_build", "libvmsim.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


UNITS = ("sim_abi", "sim_runtime", "sim_k_f32", "sim_k_s32", "sim_k_ws", "sim_k_ws8", "sim_k_wp", "sim_k_misc")


def _deps(src, pool):
    """The files of `pool` a source includes, transitively (by base name)."""
    import re
    by_name = {os.path.basename(h): h for h in pool}
    seen, todo = set(), [src]
    while todo:
        with open(todo.pop()) as fh:
            for inc in re.findall(r'#include\s+[<"]([^">]+)[">]', fh.read()):
                h = by_name.get(os.path.basename(inc))
                if h and h not in seen:
                    seen.add(h)
                    todo.append(h)
    return sorted(seen)


def build(force=False):
    """One object per kernel family (like the device build), compiled side by side by the host compiler, linked into
    tests/sim/_build/libvmsim.so; a unit is rebuilt only when one of the headers it includes changed."""
    csrc = os.path.join(ROOT, "vmap_amd", "csrc")
    pool = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")] + \
           [os.path.join(SIM, f) for f in os.listdir(SIM) if f.endswith(".h")] + [os.path.join(SIM, "include", "hip", "hip_runtime.h")]
    # the sim's wave_ops.h and fake <hip/hip_runtime.h> shadow the device ones (include order below)
    pool = [h for h in pool if h != os.path.join(csrc, "wave_ops.h")]
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cxx = CLANG if os.path.exists(CLANG) else "clang++"
    flags = [cxx, "-std=c++17", "-O2", "-mfma", "-ffp-contract=fast-honor-pragmas", "-fPIC", "-pthread", "-I", SIM, "-I", os.path.join(SIM, "include"),
             "-I", csrc, "-Wno-unused-value", "-Wno-psabi", "-Wno-pass-failed"]
    procs, objs = [], []
    for u in UNITS:
        src, obj = os.path.join(SIM, u + ".cpp"), os.path.join(os.path.dirname(OUT), u + ".o")
        objs.append(obj)
        deps = [src, os.path.abspath(__file__)] + _deps(src, pool)
        if force or not os.path.exists(obj) or any(os.path.getmtime(obj) < os.path.getmtime(d) for d in deps):
            procs.append((u, subprocess.Popen(flags + ["-c", src, "-o", obj])))
    failed = [u for u, p in procs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, f"simulator build: {failed}")
    if force or procs or not os.path.exists(OUT) or any(os.path.getmtime(OUT) < os.path.getmtime(o) for o in objs):
        subprocess.run([cxx, "-shared", "-fPIC", "-pthread", "-o", OUT + ".tmp"] + objs, check=True)
        os.replace(OUT + ".tmp", OUT)
    return OUT


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a, ty=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(ty)) if a is not None else None


def sim_step(case_or_fc, B=None, scale=None, batch=None, G=None, bwd=True, adam=None, NW=0, xcd_affine=1, weights_bf16=0, wide=False,
             split=False, rays=None, finalize_form=0):
    """Run prep + main + finalize on the simulator. Returns dict like oracle.training_step.
    split: hidden 32 on the split-bf16 kernels (step_prep_s32 / step_main_s32 / step_finalize_s32).
    rays: (origins [n,R,3], dirs [n,R,3], centres [n,3] or None) - the ABI v7 ray hand-off: the kernels get NO points tensor and rebuild
    the sample points themselves (load_point, csrc/step_kernels.h)."""
    if isinstance(case_or_fc, dict):
        c = case_or_fc
        fc, B, scale, batch = c["fc"], c["B"], c["scale"], c["batch"]
    else:
        fc = case_or_fc
    n, R, S = batch["z"].shape
    H = fc[2].shape[-1]
    if G is None:
        G = max(1, (32 if wide in (True, 1) else 64 if wide in (3, 4) else 128) // S)
    lib().vmsim_set_split(int(split))      # 0 exact fp32 (step_main_h32), 1 step_main_s32, 2 step_main_s32 with the six-product backward
    lib().vmsim_set_finalize_form(int(finalize_form))   # step_finalize_ws: 0 a thread per quad and row group, 1 one thread per quad (the library's choice for many blocks / few rows)
    lib().vmsim_set_wide(int(wide))       # 0 general kernel, 1 / True step_main_wide<4>, 3 step_main_ws, 4 step_main_wp (hidden 64 / 128)
    fc_c = [np.ascontiguousarray(a, dtype=np.float32) for a in fc]
    sizes = [a[0].size for a in fc_c]
    P = sum(sizes) + 63
    PP = (P + 63) // 64 * 64
    arr = (ctypes.POINTER(ctypes.c_float) * 14)(*[_p(a) for a in fc_c])
    Bc = np.ascontiguousarray(B, dtype=np.float32)
    sc = np.ascontiguousarray(scale, dtype=np.float32)
    pcs = np.ascontiguousarray(batch["pcs"], dtype=np.float32)
    z = np.ascontiguousarray(batch["z"], dtype=np.float32)
    gd = np.ascontiguousarray(batch["gt_depth"], dtype=np.float32)
    rgb = np.ascontiguousarray(batch["gt_rgb"], dtype=np.float32)
    sem = np.ascontiguousarray(batch["sem"], dtype=np.uint8)
    dm = np.ascontiguousarray(batch["depth_mask"], dtype=np.uint8)
    grads = np.full((n, P), np.nan, dtype=np.float32)
    loss = np.full((1,), np.nan, dtype=np.float32)
    dD = np.full((n, R), np.nan, dtype=np.float32)
    dC = np.full((n, R, 3), np.nan, dtype=np.float32)
    dO = np.full((n, R), np.nan, dtype=np.float32)
    dV = np.full((n, R), np.nan, dtype=np.float32)
    flags = np.full((4,), -1, dtype=np.int32)
    do_adam, p_out, m, v, step, lr, wd = 0, None, None, None, 1, 1e-3, 0.013
    if adam is not None:
        do_adam = 1
        p_out, m, v, step = adam["p"], adam["m"], adam["v"], adam["step"]
    ray_keep = None
    if rays is not None:
        ray_keep = [np.ascontiguousarray(x, dtype=np.float32) if x is not None else None for x in rays]
        lib().vmsim_set_rays.argtypes = [ctypes.POINTER(ctypes.c_float)] * 3
        lib().vmsim_set_rays(_p(ray_keep[0]), _p(ray_keep[1]), _p(ray_keep[2]))
        pcs = np.full_like(pcs, np.nan)              # must not be read
    try:
        rc = lib().vmsim_step(
        n, R, S, H, G, int(NW), int(xcd_affine), int(weights_bf16), arr, _p(Bc), _p(sc), _p(pcs), _p(z), _p(gd), _p(rgb),
        _p(sem, ctypes.c_uint8), _p(dm, ctypes.c_uint8), ctypes.c_float(5.0), ctypes.c_float(10.0),
        _p(grads), _p(loss), _p(dD), _p(dC), _p(dO), _p(dV), _p(flags, ctypes.c_int), int(bool(bwd)),
        do_adam, _p(p_out), _p(m), _p(v), int(step), ctypes.c_float(lr), ctypes.c_float(wd))
    finally:
        if rays is not None:
            lib().vmsim_set_rays(None, None, None)
    if rc != 0:
        raise RuntimeError(f"vmsim_step failed: {rc}")
    out = dict(loss=float(loss[0]), render_depth=dD, render_color=dC, opacity=dO, var=dV, flags=flags, grads_flat=grads)
    o = 0
    for t, a in enumerate(fc_c):
        out[f"g_fc{t}"] = grads[:, o:o + sizes[t]].reshape(a.shape)
        o += sizes[t]
    out["g_B"] = grads[:, o:o + 63].reshape(n, 21, 3)
    return out


class _SimSampleObject(ctypes.Structure):
    _fields_ = [("rgbs", ctypes.c_void_p), ("depth", ctypes.c_void_p), ("t_wc", ctypes.c_void_p), ("bbox", ctypes.c_void_p),
                ("n_keyframes", ctypes.c_int32), ("last2", ctypes.c_int32 * 2), ("center", ctypes.c_float * 3),
                ("obj_id", ctypes.c_int32), ("slots", ctypes.c_void_p), ("inst", ctypes.c_void_p)]


def sim_sample(scenes, rnds, seed=0, frame_counter=0, eps=0.1, stop_eps=0.05, nsplit=0):
    """Run frame_sample on the simulator for a list of scenes (same W,H,F,P,n1,n2); rnds = list of per-ray random dicts
    (test mode) or None (Philox mode).  nsplit > 1: the split form (frame_depth_max + frame_sample over ray slices).
    Returns dict of arrays with a leading object dimension."""
    L = lib()
    L.vmsim_set_sample_split(int(nsplit))
    assert L.vmsim_sample_object_size() == ctypes.sizeof(_SimSampleObject)
    n = len(scenes)
    s0 = scenes[0]
    W, H, F, P, n1, n2 = (s0[k] for k in ("W", "H", "F", "P", "n1", "n2"))
    S, FP = n1 + n2, F * P
    keep = []
    table = (_SimSampleObject * n)()
    for i, sc in enumerate(scenes):
        if "store" in sc:      # shared frame store: dict(rgbx u8 [C,W,H,4], depth, inst i32, t_wc) + slots i32 [K] + obj_id
            st = sc["store"]
            arrs = [np.ascontiguousarray(st["rgbx"], dtype=np.uint8), np.ascontiguousarray(st["depth"], dtype=np.float32),
                    np.ascontiguousarray(st["t_wc"], dtype=np.float32), np.ascontiguousarray(sc["bbox"], dtype=np.float32),
                    np.ascontiguousarray(sc["slots"], dtype=np.int32), np.ascontiguousarray(st["inst"], dtype=np.int32)]
            keep.append(arrs)
            table[i] = _SimSampleObject(arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data,
                                        sc["K"], (ctypes.c_int32 * 2)(*sc["last2"]), (ctypes.c_float * 3)(*[float(v) for v in sc["center"]]),
                                        int(sc["obj_id"]), arrs[4].ctypes.data, arrs[5].ctypes.data)
            continue
        arrs = [np.ascontiguousarray(sc[k]) for k in ("rgbs", "depth", "t_wc", "bbox")]
        keep.append(arrs)
        table[i] = _SimSampleObject(arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data,
                                    sc["K"], (ctypes.c_int32 * 2)(*sc["last2"]), (ctypes.c_float * 3)(*[float(v) for v in sc["center"]]),
                                    0, None, None)
    def cat(key, dt):
        if rnds is None:
            return None
        return np.ascontiguousarray(np.stack([r[key] for r in rnds]).astype(dt))
    kf, uw, uh, uz, gz = cat("kf_ids", np.int32), cat("u_w", np.float32), cat("u_h", np.float32), cat("u_z", np.float32), cat("g_z", np.float32)
    out = dict(pcs=np.full((n, FP, S, 3), np.nan, np.float32), z=np.full((n, FP, S), np.nan, np.float32),
               gt_depth=np.full((n, FP), np.nan, np.float32), gt_rgb=np.full((n, FP, 3), np.nan, np.float32),
               sem=np.full((n, FP), 255, np.uint8), depth_mask=np.full((n, FP), 255, np.uint8))
    fx, fy, cx, cy = s0["intr"]
    rc = L.vmsim_sample(table, n, W, H, F, P, n1, n2, ctypes.c_float(fx), ctypes.c_float(fy), ctypes.c_float(cx), ctypes.c_float(cy),
                        ctypes.c_float(s0["min_bound"]), ctypes.c_float(eps), ctypes.c_float(stop_eps),
                        ctypes.c_ulonglong(seed), ctypes.c_uint(frame_counter),
                        _p(kf, ctypes.c_int32), _p(uw), _p(uh), _p(uz), _p(gz),
                        _p(out["pcs"]), _p(out["z"]), _p(out["gt_depth"]), _p(out["gt_rgb"]),
                        _p(out["sem"], ctypes.c_uint8), _p(out["depth_mask"], ctypes.c_uint8))
    assert rc == 0
    return out


def sim_query(fc_k, B_k, scale_k, pts, grid=3, H=32):
    """field_query_h32 on the simulator for one object: fc_k = 14 arrays (no object dim), pts [N,3]."""
    fc_c = [np.ascontiguousarray(a, dtype=np.float32) for a in fc_k]
    arr = (ctypes.POINTER(ctypes.c_float) * 14)(*[_p(a) for a in fc_c])
    Bc = np.ascontiguousarray(B_k, dtype=np.float32)
    sc = np.ascontiguousarray([scale_k], dtype=np.float32)
    p = np.ascontiguousarray(pts, dtype=np.float32)
    n = p.shape[0]
    occ = np.full(n, np.nan, np.float32)
    rgb = np.full((n, 3), np.nan, np.float32)
    rc = lib().vmsim_query(arr, _p(Bc), _p(sc), _p(p), ctypes.c_longlong(n), _p(occ), _p(rgb), int(grid), int(H))
    assert rc == 0
    return occ, rgb

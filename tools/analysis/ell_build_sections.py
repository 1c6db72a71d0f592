#!/usr/bin/env python3
"""The sparse count store's build at the headline size, step by step (device-synchronised): espm_mu_pack_x, espm_mu_ell_count,
espm_mu_ell_plan, the lists' allocation, espm_mu_ell_fill, the read-backs - the engine's 12 ms between the initialisation and
the first iteration of a fit (MUEngine._build_ell)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from espm_amd import synth, _lib
from espm_amd.engine import MUEngine, _ptr, _stream

dev = torch.device("cuda", 0)
n, nx, ny, k = 2048, 512, 512, 5
prob = synth.make_problem(n, nx, ny, k, N=500.0, seed=0)
X = synth.sample_torch(prob, dev, seed=1000)          # (p, n) pixel-major fp32
eng = MUEngine(X, k, layout="pm", shape_2d=(nx, ny), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=4, device=dev)
st, lib = eng.st, eng.lib
i32 = dict(dtype=torch.int32, device=dev)


def stamp(label, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"   {label:58s} {1e3 * (t1 - t0):7.2f} ms", flush=True)
    return t1


for rep in range(3):
    print(f"rep {rep}")
    torch.cuda.synchronize(); t0 = tall = time.perf_counter()
    x8 = torch.empty((eng.p, st.n_pad), dtype=torch.uint8, device=dev)
    cm = os.environ.get("ESPM_ELL_BUILD_CM") != "0"
    x8c = torch.empty((st.p_pad // _lib.PPAD, st.n_cm, _lib.PPAD), dtype=torch.uint8, device=dev) if cm else None
    eng._check(lib.espm_mu_pack_x(_ptr(X), _lib.SRC_F32, _lib.LAYOUT_PM, X.shape[1], eng.n, eng.p, _ptr(x8c) if cm else None, _ptr(x8), _lib.X_U8, st.n_pad, st.p_pad,
                                  _lib.PPAD, st.n_cm, _stream()))
    t0 = stamp("pack_x (fp32 pixel-major -> 8-bit pixel-major" + (" + channel-major)" if cm else ")"), t0)
    cnt_px = torch.empty((2, st.p_pad), **i32)
    cnt_bc = torch.empty((2, st.nblk_w, st.n_cg * 64), **i32)
    klc = torch.empty(st.p_pad, dtype=torch.float32, device=dev)
    hist = os.environ.get("ESPM_ELL_BUILD_HIST") != "0"
    bkt_px = torch.empty((st.p_pad, _lib.ELL_BUCKETS), dtype=torch.uint8, device=dev) if hist else None
    bkt_bc = torch.empty((st.nblk_w, st.n_cg * 64, _lib.ELL_BUCKETS), dtype=torch.uint8, device=dev) if hist else None
    eng._check(lib.espm_mu_ell_count_hist(C.byref(st), _ptr(x8), _ptr(cnt_px), _ptr(cnt_bc), _ptr(klc), _ptr(bkt_px) if hist else None, _ptr(bkt_bc) if hist else None, _stream()))
    t0 = stamp("ell_count", t0)
    chan_perm = torch.empty((st.nblk_w, st.n_cg * 64), **i32)
    pix_perm = torch.empty(st.p_pad, **i32)
    h_off = torch.empty(2 * (st.p_pad // 64) + 1, **i32)
    w_off = torch.empty(2 * st.nblk_w * st.n_cg + 1, **i32)
    rows = torch.zeros(2, dtype=torch.int64, device=dev)
    eng._check(lib.espm_mu_ell_plan(C.byref(st), _ptr(cnt_px), _ptr(cnt_bc), _ptr(chan_perm), _ptr(pix_perm), _ptr(h_off), _ptr(w_off), _ptr(rows), _stream()))
    t0 = stamp("ell_plan", t0)
    rows_h, rows_w = (int(v) for v in rows.cpu())
    t0 = stamp("rows read back", t0)
    ell_h = torch.zeros(max(rows_h, 1) * 64, **i32)
    ell_w = torch.zeros(max(rows_w, 1) * 64, **i32)
    t0 = stamp(f"lists allocated and zeroed ({(rows_h + rows_w) * 256 / 1e6:.0f} MB)", t0)
    st.x_cm = x8c.data_ptr() if cm else None
    eng._check(lib.espm_mu_ell_fill_hist(C.byref(st), _ptr(x8), _ptr(chan_perm), _ptr(pix_perm), _ptr(h_off), _ptr(w_off), _ptr(ell_h), _ptr(ell_w),
                                         _ptr(bkt_px) if hist else None, _ptr(bkt_bc) if hist else None, _stream()))
    st.x_cm = None
    t0 = stamp("ell_fill", t0)
    if rep == 0:
        ok = torch.equal(ell_w, eng.ell["ell_w"]) and torch.equal(ell_h, eng.ell["ell_h"])
        print(f"   lists equal to the engine's own: {ok}")
    nnz = int(torch.count_nonzero(x8))
    t0 = stamp("nnz of the 8-bit copy (only when the caller does not know it)", t0)
    e = (int(cnt_px[0].sum()), int(cnt_bc[0].sum()), int((h_off[1::2] - h_off[0:-1:2]).sum()), int((w_off[1::2] - w_off[0:-1:2]).sum()))
    t0 = stamp("four sums read back", t0)
    print(f"   {'total':58s} {1e3 * (t0 - tall):7.2f} ms")
    del x8, ell_h, ell_w

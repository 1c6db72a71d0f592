#!/usr/bin/env python3
"""Times variants of the H-step / W-accumulation kernels at the headline size (K = 5, bf16).
Development tool: builds tools/tune/libespm_tune.so (product sources + tune.hip)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    csrc = os.path.join(ROOT, "espm_amd", "csrc")
    u8 = os.environ.get("TUNE_XT", "bf16") == "u8"
    out = os.path.join(HERE, "libespm_tune_u8.so" if u8 else "libespm_tune.so")
    srcs = [os.path.join(csrc, f) for f in ("mu_api.hip", "mu_h_step.hip", "mu_w_step.hip", "mu_aux.hip", "mu_ell.hip", "mu_ell_build.hip", "mu_l2.hip")] + [os.path.join(HERE, "tune.hip")]
    newest = max(os.path.getmtime(f) for f in srcs + [os.path.join(csrc, h) for h in os.listdir(csrc) if h.endswith(".hpp")])
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-I", os.path.join(ROOT, "include"), "-I", csrc, "-o", out, os.path.join(HERE, "tune.hip")] + (["-DTUNE_U8"] if u8 else []))
    return out


def main():
    if "--build-only" in sys.argv:
        print(build())
        return
    import numpy as np
    import torch
    from espm_amd import synth
    from espm_amd.engine import MUEngine, _stream
    lib = C.CDLL(build())
    lib.tune_h_name.restype = C.c_char_p
    lib.tune_w_name.restype = C.c_char_p
    n, nx, ny, k = 2048, 512, 512, 5
    rows = int(os.environ.get("TUNE_ROWS", nx))
    prob = synth.make_problem(n, rows, ny, k, N=500.0, seed=0, nx_total=nx)
    X = synth.sample_torch(prob, "cuda", seed=1000)
    W0, H0 = synth.random_init(n, k, rows * ny, seed=0, scale=500.0 / n)
    eng = MUEngine(X, k, layout="pm", shape_2d=(rows, ny), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=20,
                   x_tile=int(os.environ.get("TUNE_XTILE", 512)), x_store=os.environ.get("TUNE_XT", "bf16"))
    del X
    eng.load_state(W0, H0)
    eng.iterate(3, final_loss=False)
    st = eng.st
    p = eng.p
    eng.hpart = torch.zeros(((p + 127) // 128, 24), dtype=torch.float64, device="cuda")
    st.hpart = eng.hpart.data_ptr()
    eng.a_slab = torch.zeros((2048, k, st.n_pad), dtype=torch.float32, device="cuda")
    st.a_slab = eng.a_slab.data_ptr()
    torch.cuda.synchronize()
    reps = int(os.environ.get("TUNE_REPS", 20))

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3

    s = _stream()
    tile = C.c_int(0)
    bytes_h = n * p * (1 if os.environ.get('TUNE_XT') == 'u8' else 2) + 2 * k * p * 4
    which = os.environ.get("TUNE", "hw")
    st.compute_loss = int(os.environ.get("TUNE_LOSS", "1"))
    only = os.environ.get("TUNE_ONLY")
    if "h" in which:
        for rnd in range(2):
            v = 0
            while lib.tune_h_name(v):
                if only and str(v) not in only.split(","):
                    v += 1
                    continue
                rc = lib.tune_h(C.byref(st), st.cur, 1, v, C.byref(tile), s)
                torch.cuda.synchronize()
                assert rc == 0, rc
                t = timeit(lambda: lib.tune_h(C.byref(st), st.cur, 1, v, C.byref(tile), s))
                print(f"h[{v:2d}] {lib.tune_h_name(v).decode():22s} tile {tile.value:3d}: {t:7.1f} us  {bytes_h / t / 1e6:6.2f} TB/s", flush=True)
                v += 1
    if "w" in which:
        for nblk in (512, 768):
            v = 0
            while lib.tune_w_name(v):
                rc = lib.tune_w(C.byref(st), v, nblk, s)
                torch.cuda.synchronize()
                assert rc == 0, rc
                t = timeit(lambda: lib.tune_w(C.byref(st), v, nblk, s))
                print(f"w[{v}] {lib.tune_w_name(v).decode():12s} nblk {nblk:4d}: {t:7.1f} us  {bytes_h / t / 1e6:6.2f} TB/s", flush=True)
                v += 1


if __name__ == "__main__":
    main()

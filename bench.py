#!/usr/bin/env python3
"""Benchmark of the SmoothNMF multiplicative-update loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Metric (BASELINE.json): MU iterations / second on the 2048-channel x (512*512)-pixel X, k = 5,
SmoothNMF with simplex on H and Laplacian smoothness (lambda = 1), fp32 arithmetic; the counts are
held in the store the engine selects for them (--x-store auto: the sparse count store, 16 bits per non-zero
entry; u8 / bf16 / f32 are the dense stores).  One "step" = one full iteration (H update + W update + the loss of the state and
the relative changes the reference book-keeps every iteration, espm/estimators/base.py:316-351).
With N > 1 the image rows are sharded over the ranks (strong scaling: same total problem).

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     - the H-step kernel (dominant): algorithmic bytes / launch over its HIP-event time
  cpu_baseline - the numpy oracle (reference-faithful op sequence) on a pixel crop, on the host
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_CH, NX, NY, K = 2048, 512, 512, 5
COUNTS, LAMBDA_L = 500.0, 1.0
HBM_PEAK = 8.0e12       # B/s, MI355X_MICROARCH.md (spec; ~6.3e12 achievable)
BF16_PEAK = 2.5e15      # dense bf16 MFMA FLOP/s (spec)
VALU_F32_PEAK = 157.3e12


def cpu_baseline(prob, rows=64, iters=8):
    """Oracle (numpy fp64, reference op sequence incl. the dense identity G) on the first `rows`
    image rows; time scales linearly with pixels, so it/s(full) = it/s(crop) * crop / full."""
    from oracle import mu_oracle as oc
    from espm_amd import synth

    sub = dict(prob)
    ny = prob["shape_2d"][1]
    sub["weights"] = prob["weights"][:rows * ny]
    X = synth.sample_numpy(sub, seed=0)
    W0, H0 = synth.random_init(N_CH, K, rows * ny, seed=0, scale=COUNTS / N_CH)
    t0 = time.perf_counter()
    r = oc.fit(X, K, W=W0, H=H0, lambda_L=LAMBDA_L, simplex_H=True, simplex_W=False, shape_2d=(rows, ny),
               tol=0, no_stop_criterion=True, max_iter=iters)
    dt = time.perf_counter() - t0
    its_crop = r["n_iter"] / dt
    threads = os.cpu_count()
    try:
        from threadpoolctl import threadpool_info
        for info in threadpool_info():
            if info.get("user_api") == "blas":
                threads = info.get("num_threads", threads)
    except Exception:
        pass
    return dict(value=its_crop * rows / NX, unit="it/s", cores=int(threads), kind="port",
                sample=f"numpy fp64 oracle, {iters} iterations on the first {rows} of {NX} image rows "
                       f"({rows * ny} px x {N_CH} ch, {dt:.1f} s), scaled by {rows}/{NX}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--lambda-l", type=float, default=LAMBDA_L)
    ap.add_argument("--x-store", default="auto", choices=["auto", "ell", "u8", "bf16", "f32"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    # (rehearsal of the N > 1 path on a box with fewer GPUs than ranks: ESPM_BENCH_BACKEND=gloo puts the ranks on the
    #  GPUs that exist and exchanges the records through gloo; the driver's runs use RCCL, one rank per GPU)
    backend = os.environ.get("ESPM_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    from espm_amd import _lib, synth
    from espm_amd.engine import MUEngine, _stream
    import ctypes as C

    # ---- synthetic data: this rank's block of image rows ------------------------------------------
    rows = NX // world
    row0 = rank * rows
    if rank == world - 1:
        rows = NX - row0
    prob = synth.make_problem(N_CH, rows, NY, K, N=COUNTS, seed=0, row0=row0, nx_total=NX)
    X = synth.sample_torch(prob, device, seed=1000, row0=row0)            # (p_local, n) f32 counts
    W0, H0_full = synth.random_init(N_CH, K, NX * NY, seed=0, scale=COUNTS / N_CH)
    H0 = H0_full[:, row0 * NY:(row0 + rows) * NY]
    total_iters = args.warmup + args.steps
    eng = MUEngine(X, K, layout="pm", shape_2d=(rows, NY), lambda_L=args.lambda_l, simplex_H=True, simplex_W=False,
                   tol=0.0, max_iter=total_iters + 40, group=group, device=device, x_store=args.x_store)
    del X
    eng.load_state(W0, H0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    eng.iterate(args.warmup, final_loss=False)
    barrier()
    t0 = time.perf_counter()
    eng.iterate(args.steps, final_loss=False)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    its = args.steps / dt

    # ---- loss sanity + per-kernel timing with HIP events on the launch stream (rank-local) --------
    eng.eval_current(advance_h=False)
    hist = eng.history()
    loss_first, loss_last = float(hist["loss"][0]), float(hist["loss"][-1])
    bad = float(hist["bad"].sum())

    st = eng.st
    reps = 20

    def time_kernel(fn):
        fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in evs])) * 1e-3

    s = _stream()
    t_h = time_kernel(lambda: _lib.check(_lib.lib.espm_mu_step_h(C.byref(st), st.cur, 0, s)))
    t_h_upd = time_kernel(lambda: _lib.check(_lib.lib.espm_mu_step_h(C.byref(st), st.cur, 1, s)))
    t_w = time_kernel(lambda: _lib.check(_lib.lib.espm_mu_w_accum(C.byref(st), s)))
    p_loc = eng.p
    if eng.x_store == "ell":
        # sparse count store: 16 bits per non-zero entry of X (list padding is overhead, not algorithmic work);
        # the H-step also reads the per-pixel loss constant
        e_h, e_w = eng.ell["entries_h"], eng.ell["entries_w"]
        bytes_h = 2 * e_h + 2 * K * p_loc * 4 + p_loc * 4        # lists once, H read + written, sum x log2 x
        bytes_w = 2 * e_w + K * p_loc * 4                        # lists once, H read
        nnz = torch.tensor([float(eng.ell["nnz"])], dtype=torch.float64, device=device)
        if world > 1:
            torch.distributed.all_reduce(nnz)
        bytes_it = 2 * float(nnz.item()) + 2 * K * NX * NY * 4   # SURVEY 8(d) with X = its non-zero entries, once
        flops_it = 8.0 * K * float(nnz.item())                   # four products restricted to the non-zero entries
    else:
        xbytes = {"u8": 1, "bf16": 2, "f32": 4}[eng.x_store]
        bytes_h = N_CH * p_loc * xbytes + 2 * K * p_loc * 4          # X once, H read + written
        bytes_w = N_CH * p_loc * xbytes + K * p_loc * 4              # X once, H read
        bytes_it = N_CH * NX * NY * xbytes + 2 * K * NX * NY * 4     # X once per iteration
        flops_it = 8.0 * N_CH * K * NX * NY
    # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc passes, corrected as
    # MI355X_MICROARCH.md prescribes); measured once per kernel version and committed under profiles/
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            if world == 1:
                traffic = json.load(f)["stores"][eng.x_store]["h_step"]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    roofline = dict(bound="hbm", kernel=("h_step_ell_kernel<5,loss>" if eng.x_store == "ell" else "h_step_kernel<5,%s,...,loss>" % eng.x_store), achieved=bytes_h / t_h_upd / 1e9,
                    peak=HBM_PEAK / 1e9, unit="GB/s", frac=bytes_h / t_h_upd / HBM_PEAK, traffic=traffic,
                    bytes_per_launch=bytes_h, launch_ms=t_h_upd * 1e3, launch_ms_loss_only=t_h * 1e3,
                    w_accum=dict(achieved=bytes_w / t_w / 1e9, frac=bytes_w / t_w / HBM_PEAK, launch_ms=t_w * 1e3),
                    iteration=dict(algorithmic_GB=bytes_it / 1e9, hbm_frac=bytes_it * its / HBM_PEAK,
                                   valu_f32_frac=flops_it * its / VALU_F32_PEAK,
                                   bf16_mfma_frac=flops_it * its / BF16_PEAK))

    out = None
    if rank == 0:
        out = {
            "metric": "MU iterations/sec, 2048ch x (512*512)px X, k=5 SmoothNMF",
            "value": its, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "2048ch x (512x512)px, k=5, SmoothNMF simplex_H + Laplacian lambda=%g, "
                                   "X stored %s, W/H fp32" % (args.lambda_l, eng.x_store),
                       "n": N_CH, "shape_2d": [NX, NY], "k": K, "lambda_L": args.lambda_l, "simplex_H": True,
                       "parallelism": f"pixel-row shard x{world}" if world > 1 else "single GPU",
                       "loss_every_iteration": True},
            "loss_first": loss_first, "loss_last": loss_last, "nonfinite": bad,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu:
            full = synth.make_problem(N_CH, NX, NY, K, N=COUNTS, seed=0)
            out["cpu_baseline"] = cpu_baseline(full)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Benchmark of the SmoothNMF multiplicative-update loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Metric (BASELINE.json): MU iterations / second on the 2048-channel x (512*512)-pixel X, k = 5, SmoothNMF with simplex
on H and Laplacian smoothness (lambda = 1), fp32 arithmetic; the counts are held in the store the engine selects for
them (--x-store auto: the sparse count store, 16 bits per non-zero entry - lossless, checked on ingest; u8 / bf16 / f32
are the dense stores).  One "step" = one full iteration: H update + W update + the loss of the state and the relative
changes the reference book-keeps every iteration (espm/estimators/base.py:316-351).  With N > 1 the image rows are
sharded over the ranks (strong scaling: same total problem).  W warm-up iterations, then exactly K timed iterations
between barrier + synchronize on both sides, maximum over ranks.

Prints ONE JSON line on rank 0.  Beside the contract's fields:
  roofline      the dominant kernel (the fused H update + W accumulation launch): algorithmic bytes per launch over its duration, measured live with HIP
                events on the launch stream inside the library's own loop (espm_mu_iterate_timed).  `launch_ms` = the timed region's step minus the launch
                that follows the fused one (that launch between events: `second_launch_ms`) - what the fused launch occupies of the timed loop, so
                `fits_in_step` holds by construction; `launch_ms_event_brackets` = the fused launch between its own events (every bracket serialises the
                next dispatch behind the last completion: the upper end; `frac_event_brackets`).  rocprofv3's median of the kernel lies between the two
                (profiles/*_ks_kernel_summary.csv).  `frac` follows SURVEY 8(d)'s definition (X ONCE per iteration + H read + written, X = its stored
                form: 2 bytes per non-zero entry); `frac_lists_twice` counts both list sets the kernel streams.  `traffic`: HBM bytes per launch from the
                PMC counters (rocprofv3, separate passes), read from profiles/hbm_traffic.json with the binaries' round named in `traffic_source`.
  dense_store   the same iteration on the dense 8-bit and bf16 stores (the sparse rate depends on the 21 % non-zero
                entries of this dose; the dense rates do not)
  steady_state  300 further iterations timed the same way: the first tens of milliseconds after an idle phase run
                8-15 % slower (clocks), which a 20-step timed region sits inside
  cpu_baseline  the numpy oracle (reference-faithful op sequence) on the host cores: THREE iterations at the full size when the
                host has the memory for its dense fp64 temporaries (~30 GB), and 8 iterations on a 64-row crop of the
                SAME image scaled by 64/512 beside it (`crop`); BLAS thread pool from threadpoolctl in `threadpools`
  loss_parity_rel  final loss of the HIP path against the oracle's on that crop, same W0 / H0
  short_fit     the same loop on the engine of a 200-iteration fit (`value` is measured on the engine of a 10000-iteration fit;
                both are built with autotune="auto", MUEngine's own policy, exactly as SmoothNMF.fit builds them)
  whole_fit     five consecutive 200-iteration fits (after two warm-up fits) of the benchmark's image through SmoothNMF.fit_transform, host fp32 array in, host
                arrays out: seconds each
  c5            BASELINE configuration 5 on ONE GPU (1980 ch x 1024 x 1024 px, k = 8, G 1980 x 17, mu = 0.05): iteration time and
                its fused kernel against its own algorithmic bytes
  per_rank      (N > 1) every rank's launch times from HIP events: the local half-steps, and the W step with the record exchange in it (where a rank
                waits for its peers); the same shard with NO peer (`local_iteration_us`, `local_half_steps_us`, `local_w_step_us`: an unsharded engine on the
                rank's block) and `exchange_wait_us` = the difference of the two W steps; `exchange_transport` with `exchange_fallback_reason`; `w_crc32`
                (the replicated W: equal on all ranks); `record_exchange`: the start-up self-test of the transport
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
CROP_ROWS, CROP_ITERS = 64, 8
FIT_LENGTH = 10000      # the fit whose engine `value` is measured on: SmoothNMF(max_iter=FIT_LENGTH)


def cpu_baseline_and_parity(X_crop_pm, device, skip_full=None):
    """Oracle (numpy fp64, the reference's op sequence incl. the dense identity G and (G^T R) H^T) on the first CROP_ROWS
    image rows of the benchmark's own X, and the HIP path on the same crop from the same W0 / H0.

    Why a crop and not SURVEY 8(d)'s three full-size iterations: the reference-faithful loop forms Y and X / Y as dense
    (n, p) fp64 arrays (4.3 GB each at full size, plus the (n, n) identity products): ~13 s per iteration on this host
    and ~25 GB of temporaries.  Time is linear in the pixels (every op is), so it/s(full) = it/s(crop) * crop / full."""
    from oracle import mu_oracle as oc
    from espm_amd import synth
    from espm_amd.engine import MUEngine

    ny = NY
    X = np.ascontiguousarray(X_crop_pm.T.astype(np.float64))            # (n, p_crop)
    p_crop = X.shape[1]
    W0, H0 = synth.random_init(N_CH, K, p_crop, seed=0, scale=COUNTS / N_CH)
    # The pools run with the CPUs this process may USE (cgroup quota: 16 on the MI355X boxes), not the 256 it can see: sized by the
    # visible cores they exhaust the quota and the kernel stops the whole container for the rest of every 100 ms period
    # (espm_amd/_cpu_budget.py) - a throttled baseline would flatter the GPU.  `cores` is that budget.
    from espm_amd._cpu_budget import cpu_budget, limited_thread_pools
    budget = cpu_budget()
    threads = budget
    pools = None
    with limited_thread_pools(budget):
        t0 = time.perf_counter()
        r = oc.fit(X, K, W=W0.copy(), H=H0.copy(), lambda_L=LAMBDA_L, simplex_H=True, simplex_W=False, shape_2d=(CROP_ROWS, ny),
                   tol=0, no_stop_criterion=True, max_iter=CROP_ITERS)
        dt = time.perf_counter() - t0
        its_crop = r["n_iter"] / dt
        try:
            from threadpoolctl import threadpool_info
            pools = [{k: info.get(k) for k in ("user_api", "internal_api", "num_threads", "version", "threading_layer")} for info in threadpool_info()]
            for info in pools:
                if info.get("user_api") == "blas":
                    threads = min(budget, info.get("num_threads", threads))
            pools.append({"cpu_budget": budget, "visible_cores": os.cpu_count()})
        except Exception:
            pass
    eng = MUEngine(X, K, layout="cm", shape_2d=(CROP_ROWS, ny), lambda_L=LAMBDA_L, simplex_H=True, simplex_W=False, tol=0.0,
                   max_iter=CROP_ITERS + 2, device=device)
    eng.load_state(W0, H0)
    eng.iterate(CROP_ITERS, final_loss=True)
    torch.cuda.synchronize()
    ours = eng.history()["loss"]
    parity = dict(rel=float(abs(ours[-1] - r["losses"][-1]) / abs(r["losses"][-1])),
                  worst_rel_over_trajectory=float(np.max(np.abs(ours[1:] - r["losses"]) / np.abs(r["losses"]))),
                  max_abs_dH=float(np.abs(eng.get_H() - r["H"]).max()),
                  sample=f"{CROP_ITERS} iterations on the first {CROP_ROWS} image rows of the benchmark's X, same W0 / H0; x_store {eng.x_store}")
    crop = dict(value=its_crop * CROP_ROWS / NX, unit="it/s",
                sample=f"{CROP_ITERS} iterations on the first {CROP_ROWS} of {NX} image rows of the benchmark's own X ({p_crop} px x {N_CH} ch, "
                       f"{dt:.1f} s), scaled by {CROP_ROWS}/{NX}")
    base = dict(value=crop["value"], unit="it/s", cores=int(threads), kind="port",
                sample="numpy fp64 oracle (reference op sequence: dense identity G, three n x k x p products, global-stop bisection), "
                       + crop["sample"] + "; no full-size iteration: " + (skip_full or "not attempted"),
                crop=crop, threadpools=pools)
    return base, parity


def cpu_full_size_iteration(X_dev, iters=3):
    """the reference-faithful loop AT THE FULL SIZE - `iters` iterations (H update, W update, the loss of the new state:
    what `value` counts per step; three, as 8(d) asks, ~16 s on the GPU box's 16 granted cores), timed by the oracle around its loop; the
    initial loss and the final re-evaluation the fit also makes are outside that clock.  Needs ~30 GB of host memory for the dense (n, p) fp64 temporaries."""
    from oracle import mu_oracle as oc
    from espm_amd import synth
    X = X_dev.cpu().numpy().T.astype(np.float64)                        # (n, p) C-order, like the reference's input
    W0, H0 = synth.random_init(N_CH, K, NX * NY, seed=0, scale=COUNTS / N_CH)
    from espm_amd._cpu_budget import cpu_budget, limited_thread_pools
    with limited_thread_pools(cpu_budget()):   # (the CPUs the container may use, as in the crop leg)
        r = oc.fit(X, K, W=W0, H=H0, lambda_L=LAMBDA_L, simplex_H=True, simplex_W=False, shape_2d=(NX, NY), tol=0, no_stop_criterion=True,
                   max_iter=iters, time_iterations=True)
    return dict(value=iters / r["seconds"], unit="it/s", seconds_per_iteration=r["seconds"] / iters, iterations=iters, loss=float(r["losses"][-1]))


def self_launch(n):
    """Starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <the same arguments>` as a child and returns
    its exit code (rank 0's JSON line goes to the inherited stdout).  The parent never initialises a GPU (counting the devices
    does not, on this stack).  Fewer GPUs than ranks (a rehearsal on a one-GPU box): the ranks share the devices that exist
    and the process group is gloo - RCCL wants a device per rank - unless ESPM_BENCH_BACKEND says otherwise."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL and the exchange's hipIpc mailboxes need it on this host driver
    if torch.cuda.device_count() < n:
        env.setdefault("ESPM_BENCH_BACKEND", "gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline / parity leg")
    ap.add_argument("--no-cpu-full", action="store_true", help="CPU baseline from the 64-row crop only (skip the ~25 s full-size iteration)")
    ap.add_argument("--no-extras", action="store_true", help="skip the dense-store and steady-state legs (profiling runs)")
    ap.add_argument("--lambda-l", type=float, default=LAMBDA_L)
    ap.add_argument("--x-store", default="auto", choices=["auto", "ell", "u8", "bf16", "f32"])
    ap.add_argument("--no-autotune", action="store_true", help="keep the default launch plan instead of timing the candidates at set-up")
    ap.add_argument("--no-fused", action="store_true", help="two launches per iteration pair instead of the fused kernel (A/B)")
    args = ap.parse_args()

    # host thread pools no larger than the CPUs the container grants (espm_amd/_cpu_budget.py: pools sized by the 256 visible cores
    # exhaust a 16-CPU quota, and a throttled host cannot feed the device)
    from espm_amd._cpu_budget import cpu_budget
    torch.set_num_threads(max(1, min(torch.get_num_threads(), cpu_budget() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by itself: this process - which has not touched a GPU - starts the N ranks the way the
        # driver does (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) and relays their output and exit code
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with python -m torch.distributed.run --nproc-per-node {args.gpus} "
                         f"bench.py --gpus {args.gpus}, or plain `python bench.py --gpus {args.gpus}` (starts the ranks itself)")
    # (rehearsal of the N > 1 path on a box with fewer GPUs than ranks: ESPM_BENCH_BACKEND=gloo puts the ranks on the
    #  GPUs that exist and exchanges the records through gloo; the driver's runs use RCCL, one rank per GPU)
    backend = os.environ.get("ESPM_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    group = None
    backend_fell_back = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            # RCCL must come up AND carry a collective; if it does not (on every rank alike: a missing peer mapping, an IPC
            # mode the driver refuses), the line still comes out - process group on gloo, records through the one-shot
            # exchange if its self-test passes, else through gloo - with `backend_fell_back_from` saying so
            import datetime
            try:
                dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(seconds=180))
                probe = torch.ones(1, device=device)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError(f"RCCL all_reduce returned {probe.item()} on {world} ranks")
            except Exception as e:   # noqa: BLE001
                print(f"[bench rank {rank}] RCCL did not come up ({type(e).__name__}: {e}); process group on gloo", file=sys.stderr, flush=True)
                try:
                    dist.destroy_process_group()
                except Exception:   # noqa: BLE001
                    pass
                backend, backend_fell_back = "gloo", "nccl"
                dist.init_process_group("gloo")
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    from espm_amd import _lib, synth
    from espm_amd.engine import MUEngine, _stream
    import ctypes as C

    # ---- synthetic data: this rank's block of image rows.  Host-side preparation first, the device work (sampling, list
    # building) last, so that the timed loop starts on a busy device rather than after an idle phase ----
    rows = NX // world
    row0 = rank * rows
    if rank == world - 1:
        rows = NX - row0
    prob = synth.make_problem(N_CH, rows, NY, K, N=COUNTS, seed=0, row0=row0, nx_total=NX)
    W0, H0_full = synth.random_init(N_CH, K, NX * NY, seed=0, scale=COUNTS / N_CH)
    H0 = H0_full[:, row0 * NY:(row0 + rows) * NY]
    del H0_full
    if world > 1:   # the start-up self-test of the one-shot record exchange (espm_amd/sharding.py) at full length: reported below
        os.environ.setdefault("ESPM_XCHG_SELFTEST", "1000")
    total_iters = args.warmup + args.steps
    W0d, H0d = torch.from_numpy(W0).to(device, torch.float32), torch.from_numpy(H0).to(device, torch.float32)
    X = synth.sample_torch(prob, device, seed=1000, row0=row0)            # (p_local, n) f32 counts
    X_crop = X[:CROP_ROWS * NY].cpu().numpy() if (rank == 0 and world == 1 and not args.no_cpu) else None
    # The engine is built the way SmoothNMF(max_iter=FIT_LENGTH).fit builds it - autotune="auto": MUEngine's own policy, which times
    # the launch plans at set-up for fits of engine.AUTOTUNE_MIN_ITERS iterations or more (`value` is the
    # product's figure for a fit of that length, not a bench-only configuration; `short_fit` below is the engine of a
    # 200-iteration fit, which does not time its plans)
    eng = MUEngine(X, K, layout="pm", shape_2d=(rows, NY), lambda_L=args.lambda_l, simplex_H=True, simplex_W=False,
                   tol=0.0, max_iter=max(FIT_LENGTH, total_iters + 600), group=group, device=device, x_store=args.x_store, fused=not args.no_fused,
                   autotune=False if (args.no_fused or args.no_autotune) else "auto")
    # (W0 / H0 go up before the engine is built and load_state takes device tensors: nothing crosses the host here)
    eng.load_state(W0d, H0d)

    trace = (lambda msg: print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)) if os.environ.get("ESPM_BENCH_TRACE") else (lambda msg: None)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    def timed(n):
        barrier()
        t0 = time.perf_counter()
        eng.iterate(n, final_loss=False)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    trace("engine built, state loaded")
    eng.iterate(args.warmup, final_loss=False)
    transport = None
    if world > 1:
        # the one-shot P2P exchange counts a peer that does not deliver within its bounded wait instead of hanging: if that
        # happened during the warm-up (ranks sharing a GPU in a rehearsal, a node without peer mapping), every rank switches
        # to the collective transport and the warm-up is repeated
        barrier()
        if eng.exchange.ctx is not None and eng.exchange_health() > 0:
            eng.use_collective_exchange()
            eng.load_state(W0d, H0d)
            eng.iterate(args.warmup, final_loss=False)
        transport = eng.exchange.transport
        # a sharded engine does not time launch plans at set-up (one GPU only): the run-in that keeps a single GPU's clocks up
        # before the loop is done here on the transport that works, from the same state, which is then
        # loaded again and warmed up as asked
        eng.iterate(150, final_loss=False)
        eng.load_state(W0d, H0d)
        eng.iterate(args.warmup, final_loss=False)
    def timed_legs():
        trace(f"timed legs on {eng.exchange.transport if world > 1 else 'one GPU'}")
        dt_ = timed(args.steps)
        trace("timed steps done")
        steady_ = None
        if not args.no_extras:
            n_ss = 300
            steady_ = dict(steps=n_ss, value=n_ss / timed(n_ss), unit="it/s",
                           note="300 further iterations, same bracket: the device clocks have ramped by then")
        # ---- N > 1: every rank's launch times (HIP events on the launch stream), so that a scaling curve can be read: the local
        # half-steps against the W step that holds the record exchange - a rank that waits for a slower peer shows it there ----
        mine_ = None
        if world > 1:
            hs, ws = eng.timed_iterations(40)
            torch.cuda.synchronize()
            trace("per-launch timing done")
            mine_ = dict(rank=rank, rows=rows, half_steps_us=float(np.median(hs)), w_step_with_exchange_us=float(np.median(ws)),
                         w_step_with_exchange_p90_us=float(np.percentile(ws, 90)), lost_peers=int(eng.exchange.lost_peers()),
                         exchange_selftest=eng.exchange.selftest_result)
            # First-contact insurance for a node nobody has measured on: what this rank's shard costs with NO peer at
            # all - the unsharded engine on the rank's own block, the library's loop with HIP events around its launches - so that the
            # line separates "the shard is slow" from "the rank waits": `exchange_wait_us` = the W step with the exchange in it minus
            # the same shard's W step without one; `local_iteration_us` = the projected single-rank time of the shard.
            try:
                e_loc = MUEngine(X, K, layout="pm", shape_2d=(rows, NY), lambda_L=args.lambda_l, simplex_H=True, simplex_W=False, tol=0.0,
                                 max_iter=400, device=device, x_store=args.x_store, fused=not args.no_fused, autotune=False)
                e_loc.load_state(W0d, H0d)
                e_loc.iterate(30, final_loss=False)
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                e_loc.iterate(100, final_loss=False)
                torch.cuda.synchronize()
                loc_it = (time.perf_counter() - t0_) / 100 * 1e6
                lf, lr = e_loc.iterate_timed(40)
                mine_.update(local_iteration_us=float(loc_it), local_half_steps_us=float(np.median(lf)), local_w_step_us=float(np.median(lr)),
                             exchange_wait_us=float(np.median(ws) - np.median(lr)))
                del e_loc
            except Exception as e:   # noqa: BLE001 - diagnostics must not cost the line
                mine_["local_iteration_error"] = f"{type(e).__name__}: {e}"
            # the replicated W must be the same bits on every rank (the records are summed in rank order everywhere)
            import zlib
            mine_["w_crc32"] = int(zlib.crc32(eng.get_W().tobytes()))
            sr = eng.exchange.selftest_result or {}
            mine_["exchange_transport"] = eng.exchange.transport
            mine_["exchange_fallback_reason"] = (None if eng.exchange.transport == "p2p" else
                                                 ("forced by ESPM_XCHG" if os.environ.get("ESPM_XCHG") == "collective" else
                                                  f"self-test at start-up: lost {sr.get('lost')}, corrupt {sr.get('corrupt')}" if sr.get("fell_back_from") and (sr.get("lost") or sr.get("corrupt"))
                                                  else "a bounded wait gave up during the run (ranks sharing a device, or a peer that did not deliver)" if sr.get("n") else
                                                  "the one-shot exchange could not be opened (hipIpc peer mapping)"))
        return dt_, steady_, mine_

    dt, steady, mine = timed_legs()
    if world > 1 and eng.exchange.ctx is not None:
        # a bounded wait of the one-shot exchange can also give up AFTER a clean warm-up (ranks that share a device stall
        # each other depending on who is resident when: profiles/r04c_*): the health is asked again behind the timed legs, jointly, and
        # a run that lost a peer anywhere is timed once more on the collective transport - its first figures measured 2 s waits
        barrier()
        trace("health check behind the timed legs")
        if eng.exchange_health() > 0:
            trace("a wait gave up: collective transport")
            eng.use_collective_exchange()
            trace("exchange replaced")
            transport = eng.exchange.transport
            eng.load_state(W0d, H0d)
            trace("state reloaded")
            eng.iterate(args.warmup, final_loss=False)
            dt, steady, mine = timed_legs()
    its = args.steps / dt
    per_rank = None
    if world > 1:
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, mine)

    # ---- loss sanity + per-kernel timing with HIP events on the launch stream (rank-local) --------
    eng.eval_current(advance_h=False)
    hist = eng.history()
    loss_first, loss_last = float(hist["loss"][0]), float(hist["loss"][-1])
    bad = float(hist["bad"].sum())

    st = eng.st
    reps = 20

    def time_kernel(fn):
        """Mean of `reps` isolated launches, each between its own pair of HIP events, sequenced from Python (the half-steps of the
        two-launch paths; the fused launch is timed inside the library's loop, below)."""
        fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in evs])) * 1e-3

    s = _stream()   # torch's current stream IS the launch stream of every call below
    lib = eng.lib
    fused = bool(lib.espm_mu_fused_applies(C.byref(st)))
    p_loc = eng.p
    if eng.x_store == "ell":
        e_h, e_w = eng.ell["entries_h"], eng.ell["entries_w"]
        nnz_loc = float(eng.ell["nnz"])
        nnz = torch.tensor([nnz_loc], dtype=torch.float64, device=device)
        if world > 1:
            torch.distributed.all_reduce(nnz)
        nnz_total = float(nnz.item())
        bytes_it = 2 * nnz_total + 2 * K * NX * NY * 4             # SURVEY 8(d) with X = its non-zero entries (2 B each), once; H read + written
        flops_it = 8.0 * K * nnz_total                             # four products restricted to the non-zero entries
        bytes_h = 2 * e_h + 2 * K * p_loc * 4 + p_loc * 4          # H-step alone: its lists once, H read + written, the loss constants
        bytes_w = 2 * e_w + K * p_loc * 4                          # W accumulation alone: its lists once, H read
        bytes_fused_once = 2 * nnz_loc + 2 * K * p_loc * 4         # the fused launch by 8(d)'s count: X once, H read + written
        bytes_fused_lists = 2 * (e_h + e_w) + 2 * K * p_loc * 4 + p_loc * 4   # what it streams: both list sets (splits of counts > their field included)
        nnz_frac = nnz_total / (float(N_CH) * NX * NY)
    else:
        xbytes = {"u8": 1, "bf16": 2, "f32": 4}[eng.x_store]
        bytes_h = N_CH * p_loc * xbytes + 2 * K * p_loc * 4
        bytes_w = N_CH * p_loc * xbytes + K * p_loc * 4
        bytes_it = N_CH * NX * NY * xbytes + 2 * K * NX * NY * 4
        flops_it = 8.0 * N_CH * K * NX * NY
        bytes_fused_once = bytes_fused_lists = None
        nnz_frac = None
    traffic = traffic_source = None
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            tj = json.load(f)
        if world == 1:
            entry = tj["stores"][eng.x_store]["fused" if fused else "h_step"]
            traffic = entry["hbm_bytes_per_launch"]
            traffic_source = f"profiles/hbm_traffic.json ({entry.get('round', 'r01')} binaries, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
    except (OSError, KeyError, ValueError):
        pass
    if fused:
        # The fused launch's duration, measured live, two ways that bracket it (Weak 3: rounds 1-4 reported a figure ABOVE the
        # step it is part of - their loop was sequenced from Python and the device idled inside the brackets):
        #   `launch_ms`                = the timed region's step MINUS the launch that follows the fused one (slab reduction + W update), that
        #                                launch between HIP events on the launch stream inside the library's own loop (espm_mu_iterate_timed:
        #                                espm_mu_iterate's launches, enqueued from C) - what the fused launch occupies of the timed loop; the small
        #                                launch's bracket carries the events' own cost (a few us), so this is the LOWER end;
        #   `launch_ms_event_brackets` = the fused launch between its own events in that loop: every bracket serialises the dispatch of the next
        #                                launch behind the completion of the last (3.5-5 us per bracket) - the UPPER end, and what `frac_event_brackets`
        #                                prices.  rocprofv3's median of the same kernel (profiles/*_ks_kernel_summary.csv) lies between the two.
        step_ms = dt / args.steps * 1e3
        if world == 1:
            f_us, r_us = eng.iterate_timed(100)
            t_br, t_br_median, t_w2 = float(np.mean(f_us)) * 1e-6, float(np.median(f_us)) * 1e-6, float(np.mean(r_us)) * 1e-6
            t_f = step_ms * 1e-3 - t_w2
        else:
            t_br = t_br_median = t_f = time_kernel(lambda: _lib.check(lib.espm_mu_step_hw(C.byref(st), st.cur, s)))
            t_w2 = None
        roofline = dict(bound="hbm", kernel="mu_fused_ell_kernel<5, loss> (H update + W accumulation of a 1024-pixel block per workgroup)",
                        achieved=bytes_fused_once / t_f / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=bytes_fused_once / t_f / HBM_PEAK,
                        traffic=traffic, traffic_source=traffic_source,
                        bytes_per_launch=bytes_fused_once, launch_ms=t_f * 1e3, step_ms=step_ms,
                        second_launch_ms=(t_w2 * 1e3 if t_w2 is not None else None), fits_in_step=bool(t_f * 1e3 <= step_ms),
                        launch_ms_event_brackets=t_br * 1e3, launch_ms_event_brackets_median=t_br_median * 1e3,
                        frac_event_brackets=bytes_fused_once / t_br / HBM_PEAK,
                        launch_timing=("timed region's step minus the second launch of an iteration; that launch, and the fused one for `launch_ms_event_brackets`, "
                                       "between HIP events on the launch stream in 100 iterations of the library's loop (espm_mu_iterate_timed)") if world == 1
                                      else "HIP events, 20 isolated launches",
                        bytes_definition="SURVEY 8(d): X once (sparse store: 2 B per non-zero entry, lossless) + H read + H written",
                        frac_lists_twice=bytes_fused_lists / t_f / HBM_PEAK, bytes_lists_twice=bytes_fused_lists)
    else:
        t_h_upd = time_kernel(lambda: _lib.check(lib.espm_mu_step_h(C.byref(st), st.cur, 1, s)))
        t_w = time_kernel(lambda: _lib.check(lib.espm_mu_w_accum(C.byref(st), s)))
        roofline = dict(bound="hbm", kernel=("h_step_ell_kernel<5,loss>" if eng.x_store == "ell" else "h_step_kernel<5,%s,...,loss>" % eng.x_store),
                        achieved=bytes_h / t_h_upd / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=bytes_h / t_h_upd / HBM_PEAK,
                        traffic=traffic, traffic_source=traffic_source, bytes_per_launch=bytes_h, launch_ms=t_h_upd * 1e3,
                        bytes_definition="this launch's own streams: X once in its stored form + H read + H written",
                        w_accum=dict(achieved=bytes_w / t_w / 1e9, frac=bytes_w / t_w / HBM_PEAK, launch_ms=t_w * 1e3))
    roofline["iteration"] = dict(algorithmic_GB=bytes_it / 1e9, hbm_frac=bytes_it * its / HBM_PEAK,
                                 traffic_GB=(traffic + 11.1e6) / 1e9 if (traffic and fused) else None,   # + 11.1 MB of the slab reduction launch (profiles/hbm_traffic.json)
                                 valu_f32_frac=flops_it * its / VALU_F32_PEAK, bf16_mfma_frac=flops_it * its / BF16_PEAK)

    # ---- the same iteration on the dense stores (N = 1 only): the sparse rate is a property of the dose ----
    dense = None
    if world == 1 and not args.no_extras and eng.x_store == "ell":
        dense = {}
        for name in ("u8", "bf16"):
            e2 = MUEngine(X, K, layout="pm", shape_2d=(rows, NY), lambda_L=args.lambda_l, simplex_H=True, simplex_W=False,
                          tol=0.0, max_iter=260, device=device, x_store=name)
            e2.load_state(W0d, H0d)
            e2.iterate(20, final_loss=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e2.iterate(100, final_loss=False)
            torch.cuda.synchronize()
            dense[name + "_its"] = 100 / (time.perf_counter() - t0)
            del e2
        dense["note"] = "dense 8-bit / bf16 stores, 100 iterations after 20 warm-up, same image and state"

    # ---- the engine SmoothNMF.fit builds for a SHORT fit (max_iter = 200 < engine.AUTOTUNE_MIN_ITERS: no launch-plan timing, so the
    # timed steps sit in the clock ramp that follows the ingest) ----
    product_default = None
    if world == 1 and not args.no_extras and eng.x_store == "ell" and not (args.no_fused or args.no_autotune):
        e3 = MUEngine(X, K, layout="pm", shape_2d=(rows, NY), lambda_L=args.lambda_l, simplex_H=True, simplex_W=False,
                      tol=0.0, max_iter=max(200, total_iters + 10), device=device, x_store=args.x_store, autotune="auto")
        e3.load_state(W0d, H0d)
        e3.iterate(args.warmup, final_loss=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e3.iterate(args.steps, final_loss=False)
        torch.cuda.synchronize()
        product_default = dict(value=args.steps / (time.perf_counter() - t0), unit="it/s", steps=args.steps, warmup=args.warmup,
                               note="the engine of SmoothNMF(max_iter=200).fit: autotune='auto' leaves the launch plans untimed below 5000 iterations")
        del e3

    # ---- a whole fit through the estimator: the benchmark's image as a HOST fp32 array in, host arrays out, 200 iterations
    # (upload, the reference's passes over X, NNDSVD initialisation, store build, loop, read-back), five in a row ----
    whole_fit = None
    if world == 1 and not args.no_extras and eng.x_store == "ell" and args.x_store == "auto":
        import contextlib
        import io
        from espm_amd.estimators import SmoothNMF
        Xh = X.t().contiguous().cpu().numpy()            # (n, p) fp32, C order: what a caller of the reference hands over
        secs = []
        for rep in range(7):
            est = SmoothNMF(n_components=K, lambda_L=args.lambda_l, simplex_H=True, simplex_W=False, shape_2d=(NX, NY), max_iter=200, tol=0,
                            no_stop_criterion=True, verbose=0, random_state=0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                est.fit_transform(Xh)
            torch.cuda.synchronize()
            if os.environ.get("ESPM_FIT_TIMING"):         # (the fit's own section stamps: to stderr, the JSON line stays alone on stdout)
                print(buf.getvalue(), file=sys.stderr)
            if rep >= 2:                                  # (the first fit of a process loads kernels and libraries; the second still pays first touches of the allocator's new blocks: upload 45 ms instead of 40)
                secs.append(time.perf_counter() - t0)
        whole_fit = dict(seconds=secs, median_s=float(sorted(secs)[len(secs) // 2]), min_s=float(min(secs)), iterations=int(est.n_iter_), loss_last=float(est.losses_[-1]),
                         note="SmoothNMF(...).fit_transform(host fp32 array (2048, 262144)): five consecutive fits after two warm-up fits")
        del est, Xh

    # ---- BASELINE configuration 5 on one GPU: 1980 ch x 1024 x 1024 px, k = 8, G 1980 x 17, mu = 0.05, lambda = 1, simplex_H ----
    c5 = None
    if world == 1 and not args.no_extras and args.x_store == "auto":
        n5, nx5, ny5, k5, m5 = 1980, 1024, 1024, 8, 17
        prob5 = synth.make_problem(n5, nx5, ny5, k5, N=COUNTS, seed=0, m=m5)
        X5 = synth.sample_torch(prob5, device, seed=3000)
        W5, H5 = synth.random_init(m5, k5, nx5 * ny5, seed=0, scale=COUNTS / n5)
        e5 = MUEngine(X5, k5, layout="pm", G=prob5["G"], shape_2d=(nx5, ny5), lambda_L=1.0, mu=0.05, simplex_H=True, simplex_W=False,
                      tol=0.0, max_iter=200, device=device)
        del X5
        e5.load_state(W5, H5)
        e5.iterate(30, final_loss=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e5.iterate(100, final_loss=False)
        torch.cuda.synchronize()
        it5 = (time.perf_counter() - t0) / 100
        c5 = dict(workload="1980ch x (1024x1024)px, k=8, G 1980x17 fixed, mu=0.05, lambda_L=1, simplex_H, X stored %s" % e5.x_store,
                  us_per_iteration=it5 * 1e6, value=1.0 / it5, unit="it/s", n_gpus=1)
        if e5.x_store == "ell" and bool(e5.lib.espm_mu_fused_applies(C.byref(e5.st))):
            f5_us, r5_us = e5.iterate_timed(50)   # (HIP events around every launch of 50 iterations of the library's loop; as for the headline:)
            t5 = it5 - float(np.mean(r5_us)) * 1e-6   # the iteration minus the launches that follow the fused one
            c5["rest_of_iteration_us"] = float(np.mean(r5_us))
            c5["fused_launch_event_brackets_us"] = float(np.mean(f5_us))
            b5 = 2 * float(e5.ell["nnz"]) + 2 * k5 * nx5 * ny5 * 4
            c5["roofline"] = dict(bound="hbm", kernel="mu_fused_ell_kernel<8, loss>", launch_ms=t5 * 1e3, bytes_per_launch=b5,
                                  achieved=b5 / t5 / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=b5 / t5 / HBM_PEAK,
                                  frac_lists_twice=(2 * (e5.ell["entries_h"] + e5.ell["entries_w"]) + 2 * k5 * nx5 * ny5 * 4 + nx5 * ny5 * 4) / t5 / HBM_PEAK,
                                  nnz_frac=float(e5.ell["nnz"]) / (float(n5) * nx5 * ny5))
        del e5

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
                       "process_group": (dict(backend=backend, fell_back_from=backend_fell_back, ranks_per_device=max(1, world // max(torch.cuda.device_count(), 1)))
                                         if world > 1 else None),
                       "record_exchange": (dict(eng.exchange.selftest_result or {}, transport=transport) if world > 1 else None),
                       "autotune": bool(eng.plan_timings is not None), "engine_policy": f"autotune='auto', as SmoothNMF(max_iter={FIT_LENGTH}).fit builds it",
                       "per_rank": per_rank,
                       "loss_every_iteration": True, "launches_per_iteration": 2 if fused else 3,
                       "launch_plan": getattr(eng, "plan", None), "launch_plan_timings_us": eng.plan_timings,
                       "nnz_frac": nnz_frac, "counts_per_pixel": COUNTS},
            "loss_first": loss_first, "loss_last": loss_last, "nonfinite": bad,
            "roofline": roofline,
        }
        if steady:
            out["steady_state"] = steady
        if dense:
            out["dense_store"] = dense
        if product_default:
            out["short_fit"] = product_default
        if c5:
            out["c5"] = c5
        if whole_fit:
            out["whole_fit"] = whole_fit
        if X_crop is not None:
            # a full-size faithful iteration where the host can hold it (checked, not assumed)
            full, skip = None, None
            try:
                import psutil
                avail = psutil.virtual_memory().available / 2 ** 30
                if avail < 45:
                    skip = f"{avail:.0f} GiB of host memory available, ~30 GiB of dense fp64 temporaries + copies needed"
            except Exception as e:   # noqa: BLE001
                skip = f"psutil: {e}"
            if skip is None and not args.no_cpu_full:
                try:
                    full = cpu_full_size_iteration(X)
                except MemoryError:
                    skip = "MemoryError"
            elif args.no_cpu_full:
                skip = "--no-cpu-full"
            out["cpu_baseline"], parity = cpu_baseline_and_parity(X_crop, device, skip_full=skip)
            if full:
                cb = out["cpu_baseline"]
                cb["value"], cb["full_size"] = full["value"], full
                cb["sample"] = (f"numpy fp64 oracle (reference op sequence: dense identity G, three n x k x p products, global-stop bisection), {full['iterations']} iterations "
                                f"at the full size ({N_CH} ch x {NX * NY} px, {full['seconds_per_iteration']:.1f} s each); beside it `crop`: {cb['crop']['sample']}")
            out["loss_parity_rel"] = parity["rel"]
            out["loss_parity"] = parity
        print(json.dumps(out), flush=True)
    del X
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

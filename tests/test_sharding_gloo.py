"""world_size > 1 on CPU (gloo): the pixel-row sharding protocol of espm_amd.sharding.

Each rank owns a block of image rows and runs the per-iteration protocol with the SAME ShardExchange
object the GPU engine uses (record layout, one all-gather per iteration, fixed-order combine,
neighbour halo offsets); the local arithmetic that the HIP kernels do on the GPU box is done here with
the numpy oracle.  The sharded result must equal the single-process fit.
"""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from espm_amd import sharding
from oracle import mu_oracle as oc

N, NX, NY, K, ITERS = 24, 9, 5, 3, 6
LAM, MU, EPS = 1.5, 0.3, 1e-14


def _problem():
    rng = np.random.default_rng(11)
    p = NX * NY
    H = rng.random((K, p)) ** 2 + 0.05
    H /= H.sum(axis=0, keepdims=True)
    W = rng.random((N, K)) * 3 + 0.01
    X = rng.poisson(W @ H * 4).astype(np.float64)
    W0 = rng.random((N, K)) * 2 + 0.1
    H0 = rng.random((K, p)) + 0.05
    H0 /= H0.sum(axis=0, keepdims=True)
    return X, W0, H0


def _stencil_with_halo(H, nx, ny, top, bot):
    """(H L) on a row block: rows above/below come from the neighbours' records (None = image edge)."""
    k = H.shape[0]
    img = H.reshape(k, nx, ny)
    out = np.zeros_like(img)
    out[:, 1:, :] += img[:, 1:, :] - img[:, :-1, :]
    out[:, :-1, :] += img[:, :-1, :] - img[:, 1:, :]
    if top is not None:
        out[:, 0, :] += img[:, 0, :] - top
    if bot is not None:
        out[:, -1, :] += img[:, -1, :] - bot
    out[:, :, 1:] += img[:, :, 1:] - img[:, :, :-1]
    out[:, :, :-1] += img[:, :, :-1] - img[:, :, 1:]
    return out.reshape(k, nx * ny)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, W, H0 = _problem()
        row0, rows = sharding.split_rows(NX, world, rank)
        sl = slice(row0 * NY, (row0 + rows) * NY)
        Xl, H = X[:, sl], H0[:, sl].copy()
        n_pad = (N + 7) // 8 * 8
        ex = sharding.ShardExchange(dist.group.WORLD, K, n_pad, NY, True, "cpu")
        lay = ex.layout

        def pack(A, Hnew):
            a, hs, top, bot = ex.record_views(ex.send)
            a.zero_()
            a.view(K, n_pad)[:, :N] = torch.from_numpy(A.T.astype(np.float32))      # (k, n_pad) like a_slab
            hs.zero_()
            hs[:K] = torch.from_numpy(Hnew.sum(axis=1))
            hs[8:8 + K] = torch.from_numpy(Hnew.max(axis=1))
            img = Hnew.reshape(K, rows, NY)
            top.copy_(torch.from_numpy(img[:, 0].astype(np.float32)))
            bot.copy_(torch.from_numpy(img[:, -1].astype(np.float32)))

        def combine():
            A = np.zeros((K, n_pad), np.float32)
            hs = np.zeros(16)
            for r in range(world):                                                   # fixed rank order
                a, s, _, _ = ex.record_views(ex.recv, r)
                A = A + a.view(K, n_pad).numpy()
                hs[:8] += s[:8].numpy()
                hs[8:] = np.maximum(hs[8:], s[8:].numpy())
            return A[:, :N].T.astype(np.float64), hs

        def halos():
            t, b = ex.halo_offsets()
            row = K * NY * 4
            get = lambda o: ex.recv[o:o + row].view(torch.float32).view(K, NY).numpy().astype(np.float64)  # noqa: E731
            return (get(t) if t is not None else None), (get(b) if b is not None else None)

        # initial state: global statistics of H0 and the halo rows
        pack(np.zeros((N, K)), H)
        ex.gather()
        _, hs = combine()
        top, bot = halos()
        G = np.eye(N)
        for _ in range(ITERS):
            GW = G @ W
            Y = GW @ H
            num = GW.T @ (Xl / Y)
            den = GW.sum(axis=0)[:, None] + MU / (H + 1.0)
            maxH = hs[8:8 + K][:, None]                                              # GLOBAL (updates.py:139)
            HL = _stencil_with_halo(H, rows, NY, top, bot)
            num = num + LAM * 8 * maxH
            den = den + LAM * 8 * maxH + LAM * HL
            num = H * num
            delta, e = oc.dichotomy_simplex_exact(num, den, EPS)
            Hn = np.fmax(num / (delta + e), EPS)
            A = (Xl / (GW @ Hn)) @ Hn.T                                              # local R H^T
            pack(A, Hn)
            ex.gather()                                                              # ONE collective / iteration
            Ag, hs = combine()
            W = np.maximum(W * (G.T @ Ag) / (G.sum(axis=0)[:, None] @ hs[:K][None, :]), EPS)
            top, bot = halos()
            H = Hn
        out[rank] = (W, H, lay.nbytes)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3, 8])   # (8: a real process group of the node's size - every rank owns ONE image row, the last two)
def test_sharded_protocol_equals_single_process(world):
    X, W0, H0 = _problem()
    ref = oc.fit(X, K, W=W0.copy(), H=H0.copy(), lambda_L=LAM, mu=MU, simplex_H=True, simplex_W=False,
                 shape_2d=(NX, NY), tol=0, no_stop_criterion=True, max_iter=ITERS, exact_root=True)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = dict(out)
    Ws = [res[r][0] for r in range(world)]
    for Wr in Ws[1:]:
        np.testing.assert_array_equal(Wr, Ws[0])           # replicated W is bit-identical on all ranks
    H = np.concatenate([res[r][1] for r in range(world)], axis=1)
    # records carry fp32 (like the device path): agreement to fp32 rounding
    np.testing.assert_allclose(Ws[0], ref["W"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(H, ref["H"], rtol=2e-5, atol=1e-7)


def test_record_layout_matches_the_library():
    import __graft_entry__ as ge
    ge.build()
    from espm_amd import _lib
    for n, k, ny, p in ((2048, 5, 512, 512 * 64), (1980, 8, 1024, 1024 * 128), (100, 3, 20, 400), (7, 1, 0, 1),
                        (2048, 12, 512, 512 * 64), (300, 16, 30, 900),    # (9..16 components: the wide build's statistics block)
                        (2048, 17, 512, 512 * 64), (300, 32, 30, 900)):   # (17..32: the third build's)
        st = _lib.MUState()
        st.n, st.p, st.k, st.ny, st.x_dtype = n, p, k, ny, 1
        lib = _lib.variant(k).lib
        assert lib.espm_mu_query(C.byref(st)) == 0
        lay = sharding.record_layout(k, st.n_pad, ny)
        assert lay.nbytes == lib.espm_mu_shard_record_bytes(C.byref(st))
        assert lay.off_hstat % 8 == 0 and lay.off_top % 4 == 0 and lay.nbytes % 16 == 0


def test_split_rows():
    assert [sharding.split_rows(512, 8, r) for r in (0, 7)] == [(0, 64), (448, 64)]
    assert [sharding.split_rows(10, 3, r) for r in range(3)] == [(0, 3), (3, 3), (6, 4)]
    with pytest.raises(ValueError):
        sharding.split_rows(2, 4, 0)


def _shard_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from espm_amd.estimators.base import _Shard
        rng = np.random.default_rng(5)            # the same arrays on every rank, rank 0's made distinguishable below
        H = rng.random((K, NX * NY))
        W = rng.random((N, K)) + rank             # what a rank's own initialisation would give
        sh = _Shard(dist.group.WORLD, (NX, NY), NX * NY)
        row0, rows = sharding.split_rows(NX, world, rank)
        assert (sh.sl.start, sh.sl.stop) == (row0 * NY, (row0 + rows) * NY) and sh.shape_2d == (rows, NY) and sum(sh.counts) == NX * NY
        Wb, Hb = sh.broadcast([W, H], "cpu")                     # rank 0's arrays everywhere
        mine = torch.from_numpy(sh.cols(H)).contiguous()         # this rank's columns ...
        full = sh.gather_cols(mine)                              # ... assembled on every rank
        flat = _Shard(dist.group.WORLD, None, NX * NY)           # no image grid: a contiguous split of the pixels
        out[rank] = (Wb, Hb, full, sh.cols(None) is None, (flat.sl.start, flat.sl.stop), flat.shape_2d, flat.counts)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_estimator_shard_helper_under_gloo(world):
    """The host side of ``est.shard(group)`` (espm_amd/estimators/base.py::_Shard) without a GPU: every rank's block of image
    rows, rank 0's initial arrays on every rank, the ranks' blocks of H assembled in rank order."""
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_shard_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = dict(out)
    rng = np.random.default_rng(5)
    H = rng.random((K, NX * NY))
    W0 = rng.random((N, K))
    stops = []
    for r in range(world):
        Wb, Hb, full, none_ok, flat_sl, flat_shape, flat_counts = res[r]
        np.testing.assert_array_equal(Wb, W0)                    # rank 0's W (the others had + rank)
        np.testing.assert_array_equal(Hb, H)
        np.testing.assert_array_equal(full, H)
        assert none_ok and flat_shape is None and sum(flat_counts) == NX * NY
        stops.append(flat_sl)
    assert stops[0][0] == 0 and stops[-1][1] == NX * NY and all(a[1] == b[0] for a, b in zip(stops[:-1], stops[1:]))


# ---- shard-local ingest and initialisation on CPU tensors (round 4): the scans' combination and the sharded randomized SVD ----
def _ingest_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from espm_amd import init_device
        from espm_amd.estimators.base import _Shard
        n, nx, ny, k = 48, 12, 16, 3
        rng = np.random.default_rng(5)
        Wt, Ht = rng.random((n, k)) + 0.05, rng.random((k, nx * ny)) + 0.05
        X = rng.poisson(Wt @ Ht * 3).astype(np.float32)
        X[7, :] = 0                                     # an empty channel: only the all-reduced channel sums can know
        sh = _Shard(dist.group.WORLD, (nx, ny), nx * ny)
        Xl = torch.from_numpy(np.ascontiguousarray(X[:, sh.sl]))
        # the upload's scans of this rank's block (espm_amd/estimators/base.py: _upload_with_scans), formed here on the host tensor
        xd = Xl.double()
        scans = dict(row_sum=xd.sum(dim=1), col_sum=xd.sum(dim=0), bad=torch.zeros(3, dtype=torch.int64), s1=xd.sum(),
                     s2=(xd * torch.log(xd.clamp_min(1e-14))).sum(),
                     facts=torch.stack(((Xl != Xl.round()).sum().double(), (Xl != 0).sum().double(), Xl.max().double())))
        comb = sh.combine_scans(scans, "cm")
        Xd = torch.from_numpy(X).double()
        res = dict(ch_sum_ok=bool(torch.allclose(comb["row_sum"], Xd.sum(dim=1))), s1=float(comb["s1"]), s1_ref=float(Xd.sum()),
                   s2=float(comb["s2"]), s2_ref=float((Xd * torch.log(Xd.clamp_min(1e-14))).sum()), nnz=float(comb["facts"][1]),
                   nnz_ref=float((Xd != 0).sum()), xmax=float(comb["facts"][2]), xmax_ref=float(Xd.max()),
                   px_local=bool(torch.equal(comb["col_sum"], scans["col_sum"])), empty_ch=int((comb["row_sum"] == 0).sum()))
        # the sharded randomized SVD on the blocks against the one-process routine on the whole matrix (same random stream)
        Xf = torch.from_numpy(X)
        U1, s1, V1 = init_device.randomized_svd_device(Xf, k, 0)
        U2, s2, V2 = init_device.randomized_svd_sharded(Xl, k, 0, sh)
        res.update(ds=float(np.abs(s2 - s1).max() / s1.max()), du=float(np.abs(U2 - U1).max()), dv=float(np.abs(V2 - V1).max()),
                   vshape=tuple(V2.shape), U=U2, V=V2)
        out[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_shard_local_scans_and_sharded_randomized_svd(world):
    """What a sharded fit of a large X does before its loop, on CPU tensors under gloo: every rank scans ITS block of pixels, the sums
    that are the image's are all-reduced (channel sums, s1, s2, counts; the largest entry as a maximum), the pixel sums stay local; the
    NNDSVD's randomized SVD runs on the blocks with its contractions over the pixels all-reduced and returns the one-process routine's
    factors (to rounding: the tall factor is normalised by a Cholesky QR instead of an LU), the same on every rank."""
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_ingest_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = {r: dict(out[r]) for r in range(world)}
    for r in range(world):
        v = res[r]
        assert v["ch_sum_ok"] and v["px_local"] and v["empty_ch"] == 1
        for a, b in (("s1", "s1_ref"), ("s2", "s2_ref"), ("nnz", "nnz_ref"), ("xmax", "xmax_ref")):
            np.testing.assert_allclose(v[a], v[b], rtol=1e-12)
        assert v["vshape"] == (3, 12 * 16)
        assert v["ds"] < 1e-5 and v["du"] < 5e-4 and v["dv"] < 5e-4, (v["ds"], v["du"], v["dv"])
        np.testing.assert_array_equal(v["U"], res[0]["U"])      # replicated factors: the same bits on every rank
        np.testing.assert_array_equal(v["V"], res[0]["V"])


def _agree_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from espm_amd.estimators.base import _Shard
        sh = _Shard(dist.group.WORLD, (NX, NY), NX * NY)
        sh.agree(True, "a step everybody passes")
        try:
            sh.agree(rank != 1, "the upload", MemoryError("rank 1 ran out of memory") if rank == 1 else None)
            out[rank] = "no exception"
        except MemoryError as e:
            out[rank] = ("own", str(e))
        except RuntimeError as e:
            out[rank] = ("peer", str(e))
        t = torch.ones(1)
        dist.all_reduce(t)                   # everybody is still there for the next collective
        out[f"after{rank}"] = float(t)
    finally:
        dist.destroy_process_group()


def test_ranks_agree_before_anybody_raises():
    """a rank that fails between two collectives (its upload) must not leave its peers blocked in the next one: the ranks
    exchange a flag first; the failing rank raises its own exception, every other rank one that names the step."""
    world = 3
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_agree_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = dict(out)
    assert res[1] == ("own", "rank 1 ran out of memory")
    for r in (0, 2):
        assert res[r][0] == "peer" and "the upload failed on another rank" in res[r][1]
    assert all(res[f"after{r}"] == world for r in range(world))

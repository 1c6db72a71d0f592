"""Headline-size checks (2048 channels x 512 x 512 pixels, k = 5, SmoothNMF simplex_H + Laplacian): the
kernel variants the benchmark runs (the sparse count store it selects by default, and the dense u8 / bf16
stores with 256- and 128-pixel H tiles) are exercised at
BASELINE.json's full size through size-independent properties and through the oracle on a pixel subset
(an H update of a pixel depends only on its own X column, W, its 4 neighbours and the global row maxima)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mu_oracle as oc  # noqa: E402

N, NX, NY, K = 2048, 512, 512, 5


@pytest.fixture(scope="module")
def big():
    from espm_amd import synth
    from espm_amd.engine import MUEngine
    prob = synth.make_problem(N, NX, NY, K, N=500.0, seed=0)
    X = synth.sample_torch(prob, "cuda", seed=1000)                      # (p, n) counts
    W0, H0 = synth.random_init(N, K, NX * NY, seed=0, scale=500.0 / N)
    kw = dict(layout="pm", shape_2d=(NX, NY), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=8)
    engs = {}
    for tile, store in ((256, "u8"), (128, "u8"), ("bf16", "bf16"), ("ell", "auto")):
        eng = MUEngine(X, K, tile_px=tile if isinstance(tile, int) else None, x_store=store, **kw)
        eng.load_state(W0, H0)
        engs[tile] = eng
    rows = slice(200 * NY, 203 * NY)                                      # three image rows for the oracle
    Xsub = X[rows.start - NY:rows.stop + NY].T.double().cpu().numpy()    # with one halo row each side
    del X
    return dict(engs=engs, W0=W0, H0=H0, Xsub=Xsub, rows=rows)


def test_storage_is_lossless(big):
    """Counts <= 255 are stored as 8-bit integers (auto), bf16 on request: both are exact for this data."""
    assert big["engs"][256].x_store == "u8" and big["engs"][128].x_store == "u8" and big["engs"]["bf16"].x_store == "bf16"
    assert big["engs"]["ell"].x_store == "ell"                              # auto: 20 % of the counts are non-zero
    assert big["engs"][256].st.tile_px == 256 and big["engs"][128].st.tile_px == 128
    a = big["engs"][256].x_pm[:4096].float()
    b = big["engs"]["bf16"].x_pm[:4096].float()
    assert torch.equal(a, b)


@pytest.mark.parametrize("which", [256, "ell"])
def test_first_h_step_matches_oracle_on_a_pixel_subset(big, which):
    eng = big["engs"][which]
    eng.load_state(big["W0"], big["H0"])
    H1 = eng.step_h_only()                                                # full-size H update, tile 256
    eng.load_state(big["W0"], big["H0"])
    assert np.array_equal(H1, eng.step_h_only())                          # bitwise reproducible run to run
    W0, H0, rows = big["W0"], big["H0"], big["rows"]
    ext = slice(rows.start - NY, rows.stop + NY)
    # oracle on 5 image rows (3 + halo); the global row maxima enter through sigma * max_j H
    Hsub = H0[:, ext]
    L = oc.laplacian_matrix(5, NY)
    GW = W0
    Y = GW @ Hsub
    num = GW.T @ (big["Xsub"] / Y) + 1.0 * 8 * H0.max(axis=1, keepdims=True)
    den = GW.sum(axis=0)[:, None] + 1.0 * 8 * H0.max(axis=1, keepdims=True) + 1.0 * (Hsub @ L)
    num = Hsub * num
    delta, e = oc.dichotomy_simplex_exact(num, den)
    ref = np.fmax(num / (delta + e), 1e-14)[:, NY:-NY]                    # drop the halo rows (wrong stencil there)
    np.testing.assert_allclose(H1[:, rows], ref, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(H1.sum(axis=0), 1.0, atol=5e-6)            # simplex on every one of the 262144 pixels
    assert (H1 >= 1e-14).all() and np.isfinite(H1).all()


def test_kernel_variants_agree_and_loss_decreases(big):
    out = {}
    for tile, eng in big["engs"].items():
        eng.load_state(big["W0"], big["H0"])
        eng.iterate(6, final_loss=True)
        torch.cuda.synchronize()
        out[tile] = (eng.get_W(), eng.get_H(), eng.history())
    (Wa, Ha, ha), (Wb, Hb, hb) = out[256], out[128]
    Wc, Hc, hc = out["bf16"]
    We, He, he = out["ell"]
    np.testing.assert_allclose(Wa, We, rtol=2e-5, atol=1e-8)            # sparse count store: other summation order
    np.testing.assert_allclose(Ha, He, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(ha["loss"], he["loss"], rtol=1e-6)
    np.testing.assert_allclose(ha["rel_W"], he["rel_W"], rtol=1e-4)
    np.testing.assert_allclose(ha["rel_H"], he["rel_H"], rtol=1e-3)
    assert he["bad"].sum() == 0
    np.testing.assert_allclose(Wa, Wc, rtol=1e-6, atol=1e-10)          # u8 and bf16 stores hold the same numbers
    np.testing.assert_allclose(Ha, Hc, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(ha["loss"], hc["loss"], rtol=1e-9)
    np.testing.assert_allclose(Wa, Wb, rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(Ha, Hb, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(ha["loss"], hb["loss"], rtol=1e-6)
    assert np.all(np.diff(ha["loss"]) < 0)                                # monotone decrease of the full objective
    assert ha["bad"].sum() == 0
    np.testing.assert_allclose(Ha.sum(axis=0), 1.0, atol=5e-6)
    # mass balance of the KL multiplicative W update: sum(GW H) == sum(X) after every W step
    eng = big["engs"][256]
    sum_y = float(eng.hist[6, 3].item())
    assert abs(sum_y - eng.sum_x) / eng.sum_x < 1e-5


def test_w_step_is_linear_in_the_pixel_blocks(big):
    """R H^T accumulated over all pixels == sum of the accumulations over two halves (slab reduction)."""
    import ctypes as C
    from espm_amd import _lib
    from espm_amd.engine import _stream
    eng = big["engs"][256]
    eng.load_state(big["W0"], big["H0"])
    eng.step_h_only()
    st = eng.st
    s = _stream()
    _lib.check(_lib.lib.espm_mu_w_accum(C.byref(st), s))
    _lib.check(_lib.lib.espm_mu_w_reduce(C.byref(st), s))
    full = eng.a.clone()
    slabs = eng.a_slab.double().sum(dim=0)
    np.testing.assert_allclose(full.cpu().numpy(), slabs.cpu().numpy(), rtol=2e-5, atol=1e-6)
    half = eng.a_slab[: st.nblk_w // 2].double().sum(dim=0) + eng.a_slab[st.nblk_w // 2:].double().sum(dim=0)
    np.testing.assert_allclose(slabs.cpu().numpy(), half.cpu().numpy(), rtol=1e-12)




@pytest.mark.parametrize("algo", ["bmd", "l2_surrogate", "projected_gradient"])
def test_other_solvers_on_full_tiles(algo):
    """The other H rules on an image large enough for 512-pixel tiles, paired list groups and unit rows (512 x 512 pixels, 256
    channels), against the oracle."""
    from espm_amd import synth
    from espm_amd.estimators import SmoothNMF
    n, nx, ny, k = 256, 512, 512, 5
    prob = synth.make_problem(n, nx, ny, k, N=60.0, seed=4)
    X = synth.sample_numpy(prob, seed=4)
    X[X.sum(axis=1) == 0, 0] = 1.0
    X[0, X.sum(axis=0) == 0] = 1.0
    W0, H0 = synth.random_init(n, k, nx * ny, seed=4, scale=60.0 / n)
    kw = dict(simplex_H=True, simplex_W=False, lambda_L=1.0, mu=0)
    extra = {}
    if algo == "projected_gradient":
        L = oc.laplacian_matrix(nx, ny)
        extra["gamma"] = [float(np.abs(oc.gradH(X, np.eye(n), W0, H0, lambda_L=1.0, L=L)).max() / 0.05),
                          float(np.abs(oc.gradW(X, np.eye(n), W0, H0)).max() / (0.2 * W0.mean()))]
    ref = oc.fit(X, k, W=W0.copy(), H=H0.copy(), shape_2d=(nx, ny), algo=algo, tol=0, no_stop_criterion=True, max_iter=3, **kw, **extra)
    est = SmoothNMF(n_components=k, shape_2d=(nx, ny), algo=algo, tol=0, no_stop_criterion=True, max_iter=3, verbose=0, **kw, **extra)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        est.fit_transform(X, W=W0.copy(), H=H0.copy())
    eng = est._engine
    assert eng.x_store == "ell" and eng.st.tile_px == 512 and eng.ell["unit_rows_h"] > 0
    np.testing.assert_allclose(est.losses_, ref["losses"], rtol=1e-4)
    np.testing.assert_allclose(est.H_, ref["H"], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(est.W_, ref["W"], rtol=2e-3, atol=2e-3 * np.abs(ref["W"]).mean())


def test_pixels_without_counts_at_full_size():
    """A low dose (5 counts per pixel) leaves ~1700 of the 262144 pixels without a single count, and some channels too.  The
    sparse store (lists walked in pairs at this size, the fill's numerator from its own pass) against the fp32 store, which
    holds the reference's log_shift fill as data (base.py:519-528): three iterations, every column of H - the ones the fill
    alone decides included - W and the loss."""
    from espm_amd import synth
    from espm_amd.engine import MUEngine
    prob = synth.make_problem(N, NX, NY, K, N=5.0, seed=0)
    X = synth.sample_torch(prob, "cuda", seed=1000)
    empty = (X.sum(dim=1) == 0).cpu().numpy()
    assert 500 < empty.sum() < 5000
    W0, H0 = synth.random_init(N, K, NX * NY, seed=0, scale=5.0 / N)
    kw = dict(layout="pm", shape_2d=(NX, NY), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=4)
    out = {}
    for store in ("auto", "f32"):
        eng = MUEngine(X, K, x_store=store, **kw)
        assert eng.x_store == ("ell" if store == "auto" else "f32")
        if store == "auto":
            assert int(eng.st.ell_fill_n) == int(empty.sum()) and eng.st.tile_px == 512
        eng.load_state(W0, H0)
        eng.iterate(3, final_loss=True)
        torch.cuda.synchronize()
        h = eng.history()
        assert h["bad"].sum() == 0
        out[store] = (eng.get_W(), eng.get_H(), h["loss"])
        del eng
    (Ws, Hs, ls), (Wd, Hd, ld) = out["auto"], out["f32"]
    np.testing.assert_allclose(ls, ld, rtol=2e-6)
    np.testing.assert_allclose(Hs[:, ~empty], Hd[:, ~empty], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(Hs[:, empty], Hd[:, empty], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(Hs[:, empty].sum(axis=0), 1.0, atol=1e-5)
    np.testing.assert_allclose(Ws, Wd, rtol=2e-4, atol=2e-4 * np.abs(Wd).mean())


def test_iteration_loop_is_reproducible_across_chunkings(big):
    """espm_mu_iterate lets the W update's tail ride in the next H-step's launch and flushes it at the end of a call: the same
    8 iterations in one call, and in uneven chunks, give bit-identical W, H and history (loss, rel_W, rel_H)."""
    eng = big["engs"]["ell"]
    out = []
    for chunks in ((8,), (1, 2, 5), (3, 1, 1, 3)):
        eng.load_state(big["W0"], big["H0"])
        for c in chunks:
            eng.iterate(c, final_loss=False)
        eng.eval_current(advance_h=False)
        torch.cuda.synchronize()
        out.append((eng.get_W(), eng.get_H(), eng.history()))
    for W, H, h in out[1:]:
        assert np.array_equal(W, out[0][0]) and np.array_equal(H, out[0][1])
        for key in ("loss", "rel_W", "rel_H", "bad"):
            assert np.array_equal(h[key], out[0][2][key]), key
    assert out[0][2]["bad"].sum() == 0 and np.isfinite(out[0][2]["rel_W"][1:]).all()


def test_streamed_lists_are_the_default_here_and_change_no_bit(big):
    """The headline's lists (0.5 GB) exceed the last-level cache: the engine sets espm_mu_state.ell_stream and the one-launch
    iteration reads them with non-temporal loads (include/espm_mu.h; mu_fused_stream.hip).  A hint about the cache only: the same
    8 iterations with the switch off are the same bits - W, H, the loss history - and the library refuses other values."""
    from espm_amd import _lib
    eng = big["engs"]["ell"]
    assert eng.x_bytes > _lib.ELL_STREAM_BYTES and eng.st.ell_stream == 1 and eng.st.ell_pb == _lib.ELL_PB
    out = []
    for flag in (1, 0, 1):
        eng.st.ell_stream = flag
        eng.load_state(big["W0"], big["H0"])
        eng.iterate(8, final_loss=True)
        torch.cuda.synchronize()
        out.append((eng.get_W(), eng.get_H(), eng.history()["loss"].copy()))
    for W, H, loss in out[1:]:
        assert np.array_equal(W, out[0][0]) and np.array_equal(H, out[0][1]) and np.array_equal(loss, out[0][2])
    eng.st.ell_stream = 2
    with pytest.raises(ValueError, match="ell_stream"):
        eng.iterate(1, final_loss=False)
    eng.st.ell_stream = 1
    eng.load_state(big["W0"], big["H0"])


def test_launch_plan_autotune_restores_the_state(big):
    """MUEngine(autotune=True) times the launch plans at the first load_state (fused with dynamic / fixed units, two
    launches) on the ingested image and keeps the fastest; the loaded state must come back bit for bit, and the fit that
    follows must equal the fit of an engine that never timed anything."""
    from espm_amd.engine import MUEngine
    eng = big["engs"]["ell"]
    X = None
    # an engine of its own on the same lists would need X again: reuse the fixture's engine, toggling the switch by hand
    eng.load_state(big["W0"], big["H0"])
    eng.iterate(4, final_loss=True)
    torch.cuda.synchronize()
    ref = (eng.get_W(), eng.get_H(), eng.history()["loss"].copy())
    W0d = torch.from_numpy(big["W0"]).to("cuda", torch.float32)
    H0d = torch.from_numpy(big["H0"]).to("cuda", torch.float32)
    eng.load_state(W0d, H0d)                                   # device tensors: nothing crosses the host
    before = [t.clone() for t in (eng.w[0], eng.h[0], eng.gw_s, eng.colsum_gw, eng.hstat[0], eng.hist)]
    plan = eng.autotune_plan(iters=8, warm=2)
    assert plan in MUEngine.PLANS.values() and set(eng.plan_timings) == set(MUEngine.PLANS.values())
    assert all(5.0 < v < 2000.0 for v in eng.plan_timings.values()), eng.plan_timings
    after = (eng.w[0], eng.h[0], eng.gw_s, eng.colsum_gw, eng.hstat[0], eng.hist)
    assert all(torch.equal(a, b) for a, b in zip(before, after)) and (eng.st.cur, eng.st.it) == (0, 0)
    eng.iterate(4, final_loss=True)
    torch.cuda.synchronize()
    if eng.st.no_fused == 0:                                    # the default plan won: the same fit, bit for bit
        assert np.array_equal(eng.get_W(), ref[0]) and np.array_equal(eng.get_H(), ref[1])
    np.testing.assert_allclose(eng.history()["loss"], ref[2], rtol=2e-7)
    eng.st.no_fused = 0
    del X


@pytest.mark.parametrize("store,tile", [("u8", 256), ("bf16", 128)])
def test_matrix_core_kernels_of_the_wide_build_at_full_size(store, tile):
    """16 components on a dense store at the headline image: both contractions of the H-step and of the W accumulation on
    the matrix cores (bf16 hi/lo splits, mu_h_mfma_kernel.hpp / mu_w_mfma_kernel.hpp) against the vector kernels of the same
    build from the same state - fp32-grade agreement after three iterations, identical losses to 1e-6, bit-reproducible
    from run to run (the kernels read matrix-core results through inline asm: a missing wait state shows here)."""
    from espm_amd import synth
    from espm_amd.engine import MUEngine
    k = 16
    prob = synth.make_problem(N, NX, NY, k, N=500.0, seed=3)
    X = synth.sample_torch(prob, "cuda", seed=1003)
    W0, H0 = synth.random_init(N, k, NX * NY, seed=3, scale=500.0 / N)
    kw = dict(layout="pm", shape_2d=(NX, NY), lambda_L=1.0, mu=0.05, simplex_H=True, simplex_W=False, tol=0.0, max_iter=4,
              x_store=store, tile_px=tile)
    out = {}
    for name, fused in (("mfma", True), ("mfma_again", True), ("valu", False)):
        eng = MUEngine(X, k, fused=fused, **kw)
        assert eng.x_store == store and eng.V.KP == 16 and eng.st.tile_px == tile
        eng.load_state(W0, H0)
        eng.iterate(3, final_loss=True)
        torch.cuda.synchronize()
        h = eng.history()
        assert h["bad"].sum() == 0
        out[name] = (eng.h[eng.st.cur][:, :eng.p].clone(), eng.w[eng.st.cur].clone(), h["loss"])
        del eng
    assert torch.equal(out["mfma"][0], out["mfma_again"][0]) and torch.equal(out["mfma"][1], out["mfma_again"][1])
    np.testing.assert_allclose(out["mfma"][2], out["valu"][2], rtol=1e-6)
    dh = (out["mfma"][0] - out["valu"][0]).abs().max().item()
    dw = ((out["mfma"][1] - out["valu"][1]).abs() / (out["valu"][1].abs() + 1e-6 * out["valu"][1].abs().mean())).max().item()
    assert dh < 2e-5 and dw < 2e-4, (dh, dw)


@pytest.mark.parametrize("k", [9, 12, 16, 17, 24, 32])
@pytest.mark.parametrize("store", ["u8", "bf16"])
def test_matrix_core_w_accumulation_is_bit_reproducible_over_200_launches(k, store):
    """The gate on the 32-slot matrix instruction: `w_accum_mfma_kernel` of the wide build runs both its
    contractions on gfx950's v_mfma_f32_16x16x32_bf16 (ESPM_MFMA_K32_MASK = 12), the form that made the H-step kernel differ from
    run to run with two waves per SIMD (an ordering hazard that was not pinned down).  Two waves per SIMD is
    how this kernel runs too; what protects it is that no launch has shown a differing bit.  Here: 200 launches from the same
    state at the headline size, every slab of every launch compared with the first one on the device - one differing bit in
    40 000 slabs fails the build.  (tools/ubench/mfma_k32_hazard.hip is the stand-alone reproducer.)"""
    import ctypes
    from espm_amd import synth
    from espm_amd.engine import MUEngine, _stream
    prob = synth.make_problem(N, NX, NY, k, N=500.0, seed=3)
    X = synth.sample_torch(prob, "cuda", seed=1003)
    W0, H0 = synth.random_init(N, k, NX * NY, seed=3, scale=500.0 / N)
    eng = MUEngine(X, k, layout="pm", shape_2d=(NX, NY), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=4, x_store=store)
    del X
    assert eng.x_store == store and eng.V.KP == (16 if k <= 16 else 32) and eng.st.no_fused == 0      # matrix-core kernels allowed (17..32 components: the third build, two halves of 16)
    eng.load_state(W0, H0)
    eng.eval_current(advance_h=True)          # H' and its transposed copy, the inputs of the accumulation
    eng._flush_finalize()
    st, s = eng.st, _stream()
    eng._check(eng.lib.espm_mu_w_accum(ctypes.byref(st), s))
    ref = eng.a_slab.clone()
    assert bool(torch.isfinite(ref).all()) and float(ref.abs().max()) > 0
    differing = torch.zeros((), dtype=torch.int64, device=ref.device)
    for _ in range(200):
        eng.a_slab.fill_(-1.0)                # (a launch that skipped a slab would show)
        eng._check(eng.lib.espm_mu_w_accum(ctypes.byref(st), s))
        differing += (eng.a_slab != ref).any().to(torch.int64)
    assert int(differing.item()) == 0, f"{int(differing.item())} of 200 launches differ from the first one"


@pytest.mark.parametrize("k", [13, 16, 17, 24, 32])
@pytest.mark.parametrize("store", ["u8", "bf16"])
def test_matrix_core_h_step_is_bit_reproducible_over_200_launches(k, store):
    """The same gate on the H side: `h_step_mfma_kernel` of the wide build (13..16 components, the 16-slot matrix
    instruction - the form that ships: the 32-slot form differed from run to run in THIS kernel and was never pinned down, DESIGN.md
    section 4).  200 launches of the H update from one state at the headline size: the new H and every workgroup's record (loss
    pieces, row sums, maxima) compared with the first launch's on the device - one differing bit fails the build."""
    import ctypes
    from espm_amd import synth
    from espm_amd.engine import MUEngine, _stream
    prob = synth.make_problem(N, NX, NY, k, N=500.0, seed=4)
    X = synth.sample_torch(prob, "cuda", seed=1004)
    W0, H0 = synth.random_init(N, k, NX * NY, seed=4, scale=500.0 / N)
    eng = MUEngine(X, k, layout="pm", shape_2d=(NX, NY), lambda_L=1.0, simplex_H=True, simplex_W=False, tol=0.0, max_iter=4, x_store=store)
    del X
    assert eng.x_store == store and eng.V.KP == (16 if k <= 16 else 32) and eng.st.no_fused == 0      # matrix-core kernels allowed (17..32 components: the third build, two halves of 16)
    eng.load_state(W0, H0)
    eng._flush_finalize()
    st, s = eng.st, _stream()
    cur = st.cur
    prev = eng.h[1 - cur].clone()             # (the H update reads the other buffer - the previous H, for rel_H - before it overwrites it)
    eng.hpart.fill_(-1.0)                     # (every launch, the first included, starts from the same records: fields a launch does not write stay -1)
    eng._check(eng.lib.espm_mu_step_h(ctypes.byref(st), cur, 1, s))
    ref_h, ref_rec = eng.h[1 - cur].clone(), eng.hpart.clone()
    assert bool(torch.isfinite(ref_h).all()) and float(ref_rec.abs().max()) > 0
    differing = torch.zeros((), dtype=torch.int64, device=ref_h.device)
    for _ in range(200):
        eng.h[1 - cur].copy_(prev)
        eng.hpart.fill_(-1.0)
        eng._check(eng.lib.espm_mu_step_h(ctypes.byref(st), cur, 1, s))
        differing += ((eng.h[1 - cur] != ref_h).any() | (eng.hpart != ref_rec).any()).to(torch.int64)
    assert int(differing.item()) == 0, f"{int(differing.item())} of 200 launches differ from the first one"

"""The sparse count store (include/espm_mu.h, x_dtype = ESPM_X_ELL) built by espm_amd.ell decodes back to X.

CPU only: the builder is torch tensor plumbing and runs on host tensors too; the kernels that read the lists
are exercised by the gpu tests."""
import numpy as np
import pytest
import torch

from espm_amd import _lib, ell


from ell_decode import decode


@pytest.mark.parametrize("n,p,rate,big,tile_px", [(100, 400, 0.3, False, 64), (1980, 1300, 0.2, True, 512), (70, 2049, 1.5, True, 128)])
def test_lists_decode_to_x(n, p, rate, big, tile_px):
    rng = np.random.default_rng(n + p)
    X = rng.poisson(rate * rng.uniform(0.1, 2.0, size=(1, n)), size=(p, n)).astype(np.float32)
    if big:  # counts beyond the count field of an entry are split over several entries
        X[rng.integers(0, p, 40), rng.integers(0, n, 40)] = rng.integers(32, 256, 40)
    cbits = max(1, int(np.ceil(np.log2(n))))
    p_pad = (p + 511) // 512 * 512
    store = ell.build(torch.from_numpy(X), p_pad, cbits, tile_px, chunk=512)
    assert store["nnz"] == int((X != 0).sum())
    Xh, Xw, rows_h, rows_w = decode(store, p, n, p_pad, cbits, tile_px)
    assert np.array_equal(Xh[:p], X.astype(np.int64)) and not Xh[p:].any()
    assert np.array_equal(Xw[:p], X.astype(np.int64)) and not Xw[p:].any()
    # loss correction of the split counts: sum x log2 x - sum over the entries of x_i log2 x_i, per pixel
    xmax = (1 << (16 - cbits)) - 1
    lg = lambda v: v * np.log2(np.maximum(v, 1.0))
    Xd = X.astype(np.float64)
    nfull = np.maximum(np.ceil(Xd / xmax), 1.0) - 1.0
    ref = (lg(Xd) - nfull * lg(np.float64(xmax)) - lg(Xd - nfull * xmax)).sum(axis=1)
    np.testing.assert_allclose(store["klc"].numpy()[:p], ref, rtol=1e-6, atol=1e-6)
    assert (ref > 0).any() == bool((X > xmax).any())
    # channels of every block in order of decreasing list length, every channel exactly once
    perm = store["chan_perm"].numpy()
    pb = 2 * tile_px                                  # a block of the W lists is two H tiles (espm_mu_state.ell_pb)
    xmax_w = (1 << (16 - (pb.bit_length() - 1))) - 1
    assert store["nblk_w"] == (p + pb - 1) // pb
    for b in range(store["nblk_w"]):
        row = perm[b][perm[b] >= 0]
        assert sorted(row.tolist()) == list(range(n))
        ent_c = np.ceil(X[b * pb:(b + 1) * pb] / xmax_w).sum(axis=0)
        assert (np.diff(ent_c[row]) <= 0).all()
    # pixels of every window in order of decreasing list length; lists padded to the longest of their 64 slots only
    ent = np.ceil(X / xmax).sum(axis=1)
    ent = np.concatenate([ent, np.zeros(p_pad - p)])
    pix = store["pix_perm"].numpy().reshape(-1, tile_px)
    assert (np.sort(pix, axis=1) == np.arange(tile_px)).all()
    ent_slot = np.take_along_axis(ent.reshape(-1, tile_px), pix, axis=1)
    assert (np.diff(ent_slot, axis=1) <= 0).all()
    # unit rows: as many entries as the poorest of the 64 lists has ones, in whole batches of ELL_UNIT_ROWS rows;
    # general rows: what is left of the longest list
    ub = _lib.ELL_UNIT_ROWS
    ones = np.concatenate([(X == 1).sum(axis=1), np.zeros(p_pad - p, dtype=np.int64)])
    ones_slot = np.take_along_axis(ones.reshape(-1, tile_px), pix, axis=1).reshape(-1, 64)
    unit = ones_slot.min(axis=1) // (2 * ub) * ub
    assert np.array_equal(rows_h[:, 0], unit)
    assert np.array_equal(rows_h[:, 1], (ent_slot.reshape(-1, 64).max(axis=1) - 2 * unit + 1) // 2)
    assert store["unit_rows_h"] == unit.sum() and store["rows_h"] == rows_h.sum()
    assert store["unit_rows_w"] == rows_w[:, 0].sum() and store["rows_w"] == rows_w.sum()
    assert (rows_w[:, 0] % ub == 0).all()
    if rate < 1.0 and p >= 1024:
        assert unit.sum() > 0 and rows_w[:, 0].sum() > 0   # the sparse cases do exercise the unit rows

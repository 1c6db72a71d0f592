"""Device-side engine of the multiplicative-update loop.

``MUEngine`` owns the HBM-resident data of one fit on one GPU (its share of the image when the
pixel rows are sharded over ranks) and sequences the kernels of ``libespm_mu`` on the current
torch stream.  PyTorch is used for device memory, streams and ``torch.distributed`` only; all
arithmetic of the update rules runs in the HIP library (there is no CPU path).

Data layout in HBM (all caller-visible arrays are plain torch tensors):
  ell   sparse count store (x_store "ell"): 16-bit lists of the non-zero entries, by pixel for the H-step and by
        (1024-pixel block, channel) for the W-step (espm_amd/ell.py, include/espm_mu.h); or the dense stores
  x_cm  (p_pad/x_tile, n, x_tile) u8|bf16|f32   X, channel-major inside pixel blocks, streamed by the H-step
  x_pm  (p, n_pad)   u8|bf16|f32   pixel-major X, streamed by the W-step
  h[2]  (k, p_pad)   f32        ping-pong H;  h_t (p, 8) transposed copy of the newest H
  w[2]  (M, k)       f32        ping-pong W (M = m, or n when G is the identity)
  gw_s  (n_pad, 8)   f32        G @ W / xscale (wave-uniform rows, read through the scalar cache)
  a_slab (nblk_w, k, n_pad), a (k, n_pad)   partial / reduced  R H^T
  hist  (max_iter + 2, 8) f64   per-state loss pieces and relative changes
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import time

import numpy as np
import torch

from . import _lib
from ._lib import MUState

# the sparse count store is chosen (x_store='auto') when at most this fraction of X is non-zero
# launch plans are timed at set-up (MUEngine.autotune_plan) for fits at least this long when autotune="auto"
AUTOTUNE_MIN_ITERS = 5000
ELL_MAX_DENSITY = 0.5   # measured crossover with the dense 8-bit store at k = 5: 42 % non-zero 312 vs 394 us, 58 % 410 vs 399 us


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(device=None):
    if not torch.cuda.is_available():
        raise RuntimeError("espm_amd needs an AMD GPU (MI355X / gfx950): no HIP device is visible and there is "
                           "no CPU fallback for the multiplicative-update path")
    return torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")


class MUEngine:
    """One SmoothNMF problem resident on one GPU.

    Parameters mirror ``multiplicative_step_h`` / ``multiplicative_step_w``
    (espm/estimators/updates.py:6-16, :83) and the estimator attributes that feed them
    (espm/estimators/smooth_nmf.py:325-339, :405-414).

    X : (n, p) array (``layout="cm"``) or (p, n) (``layout="pm"``, hyperspy's layout); numpy or a
        torch tensor (host or device), float32/float64.  When ``group`` is given, X is this rank's
        block of image rows and ``shape_2d`` its local (rows, ny).
    """

    def __init__(self, X, n_components, *, layout="cm", G=None, shape_2d=None, lambda_L=0.0, mu=0,
                 epsilon_reg=1.0, simplex_H=False, simplex_W=True, log_shift=1e-14, dicotomy_tol=1e-5,
                 tol=1e-4, sigmaL=8.0, fixed_H=None, fixed_W=None, simplex_rows=None, xscale=1.0,
                 x_store="auto", max_iter=200, device=None, group=None, compute_loss=True,
                 fix_zero_lines=True, gw_floor=1e-30, x_tile=None, tile_px=None, h_variant=None, bregman=False, h_rule=0, pg_gamma_w=0.0,
                 filled_channels=None, filled_pixels=None, frobenius=False, fused=True, force_sharded=False, autotune=False, x_facts=None):
        # x_facts: what the caller already knows about X exactly as handed over (espm_amd/estimators/base.py: the scans that ride
        # behind the upload) - {"nonneg": True, "sum_x": float, "is_count": integers <= 255, "nnz": int}: the passes over X that
        # would establish the same here (9 ms at the headline size) are skipped.  One GPU, no lines to fill.
        self.device = require_gpu(device)
        self.group = group
        self.world = torch.distributed.get_world_size(group) if group is not None else 1
        self.rank = torch.distributed.get_rank(group) if group is not None else 0
        # the sharded code path (records, exchange, combine) also for a group of ONE rank: what a rank of an N-GPU run does
        # per iteration, measurable on one GPU (tools/analysis/shard_iter.py)
        self.sharded = self.world > 1 or (bool(force_sharded) and group is not None)
        dev = self.device
        k = int(n_components)
        self.k = k
        self.V = _lib.variant(k)          # the build with the kernels for k components (1..8, 9..16, or 17..32 on the dense stores)
        self.lib, self._check = self.V.lib, self.V.check

        _t_dbg = [time.perf_counter()] if os.environ.get("ESPM_ENGINE_TIMING") else None

        def _tick(name):   # ESPM_ENGINE_TIMING=1: device-synchronised time of the set-up steps (tools/analysis)
            if _t_dbg is not None:
                torch.cuda.synchronize()
                now = time.perf_counter()
                print(f"[engine set-up] {name}: {1e3 * (now - _t_dbg[0]):.1f} ms", file=sys.stderr, flush=True)
                _t_dbg[0] = now

        # ---- X to the device, zero lines, storage type ------------------------------------------
        Xd = X if isinstance(X, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(X))
        if Xd.dtype not in (torch.float32, torch.float64):
            Xd = Xd.to(torch.float64)
        Xd = Xd.to(dev)
        if Xd.dim() != 2:
            raise ValueError("X must be 2-D")
        if layout == "pm":
            p, n = Xd.shape
            ch_axis, px_axis = 1, 0
        elif layout == "cm":
            n, p = Xd.shape
            ch_axis, px_axis = 0, 1
        else:
            raise ValueError("layout must be 'cm' or 'pm'")
        self.n, self.p = int(n), int(p)
        self.out_dtype = np.float64 if Xd.dtype == torch.float64 else np.float32
        if x_facts is not None and (group is not None or fix_zero_lines):
            x_facts = None
        if not (x_facts and x_facts.get("nonneg")) and bool((Xd < 0).any()):
            raise ValueError("Negative values in data")  # espm/estimators/base.py:528
        # channels that hold nothing but the log_shift fill (all ranks see the same mask): the sparse store may leave them empty
        empty_ch = None
        if filled_channels is not None:
            empty_ch = torch.as_tensor(filled_channels, dtype=torch.bool).to(dev)
            if empty_ch.shape != (self.n,):
                raise ValueError("filled_channels must be a boolean mask over the n channels")
        empty_px = None   # (and the pixels without counts, local to the rank)
        if filled_pixels is not None:
            empty_px = torch.as_tensor(filled_pixels, dtype=torch.bool).to(dev)
            if empty_px.shape != (self.p,):
                raise ValueError("filled_pixels must be a boolean mask over the p pixels")
        if fix_zero_lines:
            # all-zero channels / pixels become log_shift, base.py:519-528 (channel sums are global)
            ch_sum = Xd.sum(dim=px_axis, dtype=torch.float64)
            if group is not None:
                torch.distributed.all_reduce(ch_sum, group=group)
            px_sum = Xd.sum(dim=ch_axis, dtype=torch.float64)
            zc, zp = ch_sum == 0, px_sum == 0
            if bool(zc.any()) or bool(zp.any()):
                Xd = Xd.clone()
                empty_ch, empty_px = zc, zp
                if layout == "cm":
                    Xd[:, zp] = log_shift
                    Xd[zc, :] = log_shift
                else:
                    Xd[zp, :] = log_shift
                    Xd[:, zc] = log_shift
        if x_facts and "sum_x" in x_facts:
            self.sum_x = float(x_facts["sum_x"])
        else:
            self.sum_x = Xd.sum(dtype=torch.float64)
            if group is not None:
                torch.distributed.all_reduce(self.sum_x, group=group)
            self.sum_x = float(self.sum_x)
        _tick("sign check, empty lines, sum of X")
        self.bregman = bool(bregman)
        if self.bregman:
            # Bregman variant (updates.py:40-48, :120-125): sums of X over the channels (per pixel) and over the pixels
            # (per channel, global); built for G = identity only - the reference's own W step needs a square G
            if G is not None:
                raise NotImplementedError("the Bregman variant (algo='bmd' / use_bregman) is built for G = identity only: "
                                          "the reference's W step fails for a non-square G (updates.py:43)")
            sr_ch = Xd.sum(dim=px_axis, dtype=torch.float64)
            if group is not None:
                torch.distributed.all_reduce(sr_ch, group=group)
            self._breg_ch = sr_ch.to(torch.float32).contiguous()
            self._breg_px_local = Xd.sum(dim=ch_axis, dtype=torch.float64).to(torch.float32)
        # A fit with the Frobenius data term (l2=True with algo="l2_surrogate", smooth_nmf.py:223-237, :404-413, base.py:197-198):
        # quadratic-surrogate H step, Frobenius W step and loss; on the dense fp32 store
        self.frobenius = bool(frobenius)
        if self.frobenius:
            if float(xscale) != 1.0 or int(h_rule) != 1 or simplex_W:
                raise NotImplementedError("the Frobenius fit is built for xscale = 1 (hand over the scaled X), "
                                          "h_rule = 1 and no simplex over W (updates.py:31-36 has none)")
            x_store = "f32"
        if h_variant:
            raise NotImplementedError("h_variant=1 (Y = GW H on the matrix cores) was retired: slower than the vector kernels "
                                      "at k <= 8 and sensitive to a transcendental-operand hazard (DESIGN.md)")
        refill = False

        def set_empty(v):
            for mask, axis in ((empty_px, px_axis), (empty_ch, ch_axis)):   # (the reference's order, base.py:524-525)
                if mask is not None:
                    if axis == 0:
                        Xd[mask, :] = v
                    else:
                        Xd[:, mask] = v
        if x_store in ("auto", "ell"):
            # u8 / ell: integer counts <= 255.  All-zero channels / pixels were filled with 1e-14 above (base.py:519-528),
            # which is not an integer: such data keep the bf16 store and the reference's exact semantics.
            # ell (non-zero entries only) when at most ELL_MAX_DENSITY of the entries are non-zero and the GW table
            # fits in LDS; the decision is taken jointly by all ranks.
            # Channels that are empty in the whole image (common in measured spectra: the bins below the detector's
            # threshold and above the beam energy) do not cost the sparse store: their fill of log_shift = 1e-14 counts
            # per bin moves W, H and the loss by O(1e-14) and is left out of the lists; without
            # the sparse store the fill stays, as in the reference.  Pixels without a single count (holes, vacuum, low
            # dose) keep empty lists too: what their fill contributes to W and to the loss is of the same order, but under
            # simplex_H it alone decides their column of H - the H-step adds its numerator (include/espm_mu.h, ell_fill_*).
            unfilled = (not self.bregman and ((empty_ch is not None and bool(empty_ch.any()))
                                              or (empty_px is not None and bool(empty_px.any()))))
            if unfilled:
                set_empty(0)
            known = x_facts if (x_facts and not unfilled and "is_count" in x_facts and "nnz" in x_facts) else None
            is_count = known["is_count"] if known else ((Xd == Xd.round()).all() & (Xd.max() <= 255))
            if bool(is_count):
                code = 2
            else:   # (the bf16 round trip is two more passes over X and two temporaries of its size: only when it decides)
                code = 1 if bool((Xd.to(torch.bfloat16).to(Xd.dtype) - Xd).abs().max() <= 1e-16) else 0
            self.x_store_note = None
            if code == 2:
                from . import ell as _ell
                n_pad8 = (self.n + 7) // 8 * 8
                fits = k <= _lib.WIDE_MAX_K and self.n <= 16384 and _ell.lds_bytes_h(n_pad8, k) <= _lib.ELL_LDS_MAX
                nnz_x = int(known["nnz"]) if known else int(torch.count_nonzero(Xd))
                self._nnz_known = nnz_x          # (the sparse store's build does not count again: a pass over the 8-bit copy was 2.6 of its 12 ms)
                sparse = float(nnz_x) <= ELL_MAX_DENSITY * Xd.numel()
                if fits and (x_store == "ell" or sparse):
                    code = 3
                elif sparse and k > _lib.WIDE_MAX_K:
                    self.x_store_note = (f"sparse count data, but the sparse store is built for up to {_lib.WIDE_MAX_K} components (k={k}: a table row of "
                                         "32 floats leaves a workgroup's LDS no room): the dense 8-bit store is used, both contractions on the matrix cores")
                elif sparse and not fits:
                    # not silently (missing 3): sparse count data that the sparse store would take - about 3 x the dense store's
                    # rate at 20 % non-zero entries - but whose G W table does not fit a workgroup's LDS next to a tile's numerators
                    # (rows of 12 / 16 floats from 9 / 13 components on: 16 components stop at 2048 channels, 12 at 2896)
                    import warnings
                    self.x_store_note = (f"sparse count data, but the sparse store's table for n={self.n}, k={k} needs "
                                         f"{_ell.lds_bytes_h(n_pad8, k)} bytes of LDS (limit {_lib.ELL_LDS_MAX}): the dense 8-bit store is used, "
                                         "about 3 x slower per iteration at this density")
                    warnings.warn("espm_amd: " + self.x_store_note, RuntimeWarning, stacklevel=3)
            flag = torch.tensor([code], device=dev, dtype=torch.int32)
            _tick("storage type (integer counts, bf16-exact, density)")
            if group is not None:
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=group)
            if unfilled and int(flag.item()) != 3:
                set_empty(log_shift)
                flag.fill_(0)      # (the fill is neither an integer nor a bf16 value)
            if x_store == "ell" and int(flag.item()) != 3:
                raise ValueError("x_store='ell' needs integer counts <= 255, n <= 16384 and a GW table that fits in LDS (12- or 16-float rows from 9 components on)")
            refill = unfilled and int(flag.item()) == 3 and (filled_channels is not None or filled_pixels is not None)  # (the caller's tensor: put the fill back)
            x_store = ("f32", "bf16", "u8", "ell")[int(flag.item())]
        if int(h_rule) != 0 and x_store in ("u8", "bf16"):
            x_store = "f32"   # the alternate H rules are built for the sparse and the fp32 store
        if x_store not in ("u8", "bf16", "f32", "ell"):
            raise ValueError("x_store must be 'auto', 'ell', 'u8', 'bf16' or 'f32'")
        self.x_store = x_store
        self.x_store_note = getattr(self, "x_store_note", None)   # (why 'auto' did not take the sparse store for sparse count data, if so)

        st = MUState()
        self.st = st
        st.n, st.p, st.k = self.n, self.p, k
        st.x_dtype = {"u8": _lib.X_U8, "bf16": _lib.X_BF16, "f32": _lib.X_F32, "ell": _lib.X_ELL}[x_store]
        if shape_2d is not None:
            nx, ny = int(shape_2d[0]), int(shape_2d[1])
            if nx * ny != p:
                raise ValueError(f"shape_2d {shape_2d} does not match the {p} pixels of X")
            st.nx, st.ny, st.grid_mode = nx, ny, 1
        else:
            st.nx, st.ny, st.grid_mode = 0, 0, 0
        self._check(self.lib.espm_mu_query(C.byref(st)))
        # tests: the sparse store's full geometry (512-pixel H tiles, hence the fused kernel) on images far smaller than the
        # ones espm_mu_query gives it to - ESPM_FORCE_ELL_TILE=512 (or tile_px=512)
        force_tile = tile_px if tile_px is not None else (int(os.environ["ESPM_FORCE_ELL_TILE"]) if os.environ.get("ESPM_FORCE_ELL_TILE") else None)
        if x_store == "ell" and force_tile in (64, 128, 256, 512):
            st.tile_px = st.x_tile = int(force_tile)
            st.ell_pb = 2 * st.tile_px        # (a block of the W accumulation is two H tiles, include/espm_mu.h)
            st.nblk_w = (p + st.ell_pb - 1) // st.ell_pb
        p_total = torch.tensor([p], dtype=torch.int64, device=dev)
        if group is not None:
            torch.distributed.all_reduce(p_total, group=group)
        st.p_total = int(p_total.item())
        self.p_total = st.p_total

        self.ell = None
        self.fill_px = self.fill_num = None
        if x_store == "ell":
            from . import ell as _ell
            self.x_cm = self.x_pm = None
            if os.environ.get("ESPM_ELL_BUILDER", "hip") == "torch":  # the tensor-op builder (tests cross-check the two)
                self.ell = _ell.build(Xd if layout == "pm" else Xd.t(), st.p_pad, st.ell_cbits, st.tile_px)
            else:
                self.ell = self._build_ell(Xd.contiguous(), layout)
                _tick("sparse store build")
            assert self.ell["n_cg"] == st.n_cg and self.ell["nblk_w"] == st.nblk_w
            # pixels without counts: marked in the per-pixel loss constants, their fill's numerator has its own small pass
            if empty_px is not None and bool(empty_px.any()):
                idx = torch.nonzero(empty_px).flatten()
                if idx.numel() >= 1 << 24:
                    raise NotImplementedError("more than 2^24 pixels without counts")
                self.fill_px = idx.to(torch.int32).contiguous()
                self.fill_num = torch.zeros((k, idx.numel()), dtype=torch.float32, device=dev)
                self.ell["klc"][idx] = -(torch.arange(idx.numel(), device=dev, dtype=torch.float32) + 1.0)
            self.x_bytes = 4 * (self.ell["ell_h"].numel() + self.ell["ell_w"].numel())
        else:
            xt = {"u8": torch.uint8, "bf16": torch.bfloat16, "f32": torch.float32}[x_store]
            if tile_px is not None:  # override the H-step tile chosen by espm_mu_query (tests, tuning)
                st.tile_px = int(tile_px)
                st.x_tile = int(tile_px)
            if x_tile is not None:
                st.x_tile = int(x_tile)
            self.x_cm = torch.empty((st.p_pad // st.x_tile, st.n_cm, st.x_tile), dtype=xt, device=dev)
            self.x_pm = torch.empty((self.p, st.n_pad), dtype=xt, device=dev)
            Xd = Xd.contiguous()
            self._check(self.lib.espm_mu_pack_x(_ptr(Xd), _lib.SRC_F64 if Xd.dtype == torch.float64 else _lib.SRC_F32,
                                     _lib.LAYOUT_PM if layout == "pm" else _lib.LAYOUT_CM, Xd.shape[1], self.n, self.p,
                                     _ptr(self.x_cm), _ptr(self.x_pm), st.x_dtype, st.n_pad, st.p_pad, st.x_tile, st.n_cm, _stream()))
            self.x_bytes = self.x_cm.numel() * self.x_cm.element_size() + self.x_pm.numel() * self.x_pm.element_size()
        if refill:
            set_empty(log_shift)
        torch.cuda.current_stream().synchronize()
        del Xd

        # ---- G ------------------------------------------------------------------------------------
        if G is not None:
            Gh = np.ascontiguousarray(np.asarray(G, dtype=np.float32))
            if Gh.ndim != 2 or Gh.shape[0] != self.n:
                raise ValueError(f"G must be (n={self.n}, m), got {Gh.shape}")
            self.m = Gh.shape[1]
            self.g = torch.from_numpy(Gh).to(dev)
            self.g_t = torch.zeros((self.m, st.n_pad), dtype=torch.float32, device=dev)   # coalesced reads in the W finish
            self.g_t[:, :self.n] = self.g.t()
            self.colsum_g = torch.from_numpy(np.asarray(G, dtype=np.float64).sum(axis=0).astype(np.float32)).to(dev)
            st.m = self.m
        else:
            self.m, self.g, self.colsum_g, self.g_t = 0, None, None, None
            st.m = 0
        self.M = self.m if self.m > 0 else self.n

        # ---- hyper-parameters -----------------------------------------------------------------------
        st.simplex_h, st.simplex_w = int(bool(simplex_H)), int(bool(simplex_W))
        st.compute_loss = int(bool(compute_loss))
        st.lambda_l, st.sigma_l = float(lambda_L), float(sigmaL)
        st.h_rule = int(h_rule)   # 1: quadratic surrogate of the Laplacian term (multiplicative_step_hq), 2: projected gradient
        st.pg_gamma_w = float(pg_gamma_w)
        st.eps_reg, st.log_shift = float(epsilon_reg), float(log_shift)
        st.dicotomy_tol, st.rel_tol = float(dicotomy_tol), float(tol)
        self.rel_tol = float(tol)
        st.xscale, st.gw_floor = float(xscale), float(gw_floor)
        self.lambda_L, self.xscale = float(lambda_L), float(xscale)
        mu_arr = np.asarray(mu, dtype=np.float64)
        if mu_arr.ndim == 0 and float(mu_arr) == 0.0:
            self.mu = None
        else:
            self.mu = torch.from_numpy(np.broadcast_to(mu_arr, (k,)).astype(np.float32).copy()).to(dev)

        # ---- state and workspaces --------------------------------------------------------------------
        f32 = dict(dtype=torch.float32, device=dev)
        f64 = dict(dtype=torch.float64, device=dev)
        self.w = [torch.zeros((self.M, k), **f32) for _ in range(2)]
        self.h = [torch.ones((k, st.p_pad), **f32) for _ in range(2)]  # pad columns stay positive
        self.h_t = torch.zeros((self.p, self.V.KP), **f32)
        self.gw_s = torch.zeros((st.n_pad, self.V.KP), **f32)
        self.colsum_gw = torch.zeros(self.V.KP, **f64)
        nblk_h = (self.p + st.tile_px - 1) // st.tile_px
        self.hpart = torch.zeros((nblk_h, self.V.HP_STRIDE), **f64)
        self.hstat = [torch.zeros(self.V.HS_STRIDE, **f64) for _ in range(2)]
        _tick("rest up to the state buffers")
        self.a_slab = torch.zeros((st.nblk_w, k, st.n_pad), **f32)
        self.a = torch.zeros((k, st.n_pad), **f32)
        self.w_scratch = torch.zeros((2, self.M, k), **f32)
        self.hist_len = int(max_iter) + 3   # states 0..max_iter, a spare, and a scratch slot (the last) for loss-only evaluations
        self.hist = torch.zeros((self.hist_len, _lib.HI_STRIDE), **f64)
        self.frob = torch.zeros(self.hist_len, **f64) if self.frobenius else None
        self.fixed_h = self._pad_h(fixed_H) if fixed_H is not None else None
        self.fixed_w = (torch.from_numpy(np.ascontiguousarray(np.asarray(fixed_W, dtype=np.float32))).to(dev)
                        if fixed_W is not None else None)
        if simplex_rows is not None:
            mask = np.zeros(self.M, dtype=np.int32)
            mask[np.asarray(simplex_rows)] = 1
            self.simplex_rows = torch.from_numpy(mask).to(dev)
            if log_shift > 0 and simplex_W and mask.sum() * log_shift >= 1:
                raise ValueError("No solution exists!")
        else:
            self.simplex_rows = None

        if self.ell is None:
            st.x_cm, st.x_pm = self.x_cm.data_ptr(), self.x_pm.data_ptr()
        else:
            st.x_cm = st.x_pm = None
            st.ell_h, st.ell_h_off, st.ell_klc = (self.ell[key].data_ptr() for key in ("ell_h", "ell_h_off", "klc"))
            if self.fill_px is not None:
                st.ell_fill_px, st.ell_fill_num, st.ell_fill_n = self.fill_px.data_ptr(), self.fill_num.data_ptr(), int(self.fill_px.numel())
            st.ell_w, st.ell_w_off, st.chan_perm = (self.ell[key].data_ptr() for key in ("ell_w", "ell_w_off", "chan_perm"))
            st.pix_perm = self.ell["pix_perm"].data_ptr()
        st.g = self.g.data_ptr() if self.g is not None else None
        st.g_t = self.g_t.data_ptr() if self.g_t is not None else None
        st.colsum_g = self.colsum_g.data_ptr() if self.colsum_g is not None else None
        st.w[0], st.w[1] = self.w[0].data_ptr(), self.w[1].data_ptr()
        st.h[0], st.h[1] = self.h[0].data_ptr(), self.h[1].data_ptr()
        st.gw_s, st.colsum_gw, st.h_t = self.gw_s.data_ptr(), self.colsum_gw.data_ptr(), self.h_t.data_ptr()
        st.gw_a, st.gw_p = None, None
        st.mu = self.mu.data_ptr() if self.mu is not None else None
        st.fixed_h = self.fixed_h.data_ptr() if self.fixed_h is not None else None
        st.fixed_w = self.fixed_w.data_ptr() if self.fixed_w is not None else None
        st.simplex_rows = self.simplex_rows.data_ptr() if self.simplex_rows is not None else None
        st.halo_top = st.halo_bot = None
        if self.bregman:
            self.breg_px = torch.zeros(st.p_pad, **f32)
            self.breg_px[:self.p] = self._breg_px_local
            del self._breg_px_local
            st.breg_sr_px, st.breg_sr_ch = self.breg_px.data_ptr(), self._breg_ch.data_ptr()
        else:
            st.breg_sr_px = st.breg_sr_ch = None
        st.hpart = self.hpart.data_ptr()
        st.hstat[0], st.hstat[1] = self.hstat[0].data_ptr(), self.hstat[1].data_ptr()
        st.a_slab, st.a, st.w_scratch = self.a_slab.data_ptr(), self.a.data_ptr(), self.w_scratch.data_ptr()
        st.hist, st.hist_len = self.hist.data_ptr(), self.hist_len
        self.pg_q = torch.zeros((self.hist_len, 2), **f64) if float(pg_gamma_w) > 0 or int(h_rule) == 2 else None
        st.pg_q = self.pg_q.data_ptr() if self.pg_q is not None else None
        st.cur, st.it = 0, 0
        # both half-steps of an iteration in one launch where the library's fused kernel applies (sparse store at 512-pixel
        # tiles, default H rule: include/espm_mu.h, no_fused); `fused=False` keeps the two launches (A/B, tests)
        if os.environ.get("ESPM_FUSED"):   # tests: "always" runs the fused launch on small blocks too, "0" never
            fused = {"0": False, "1": True}.get(os.environ["ESPM_FUSED"], os.environ["ESPM_FUSED"])
        st.no_fused = {"static": 2, "always": 3}.get(fused, 0 if fused else 1)   # ("always": also blocks below ESPM_FUSED_MIN_PB pixels)
        # lists larger than the last-level cache are read with loads that do not allocate there (include/espm_mu.h: ell_stream;
        # ESPM_ELL_STREAM_MB: another threshold, for A/B - 0 streams always, a huge one never)
        limit = float(os.environ["ESPM_ELL_STREAM_MB"]) * 2 ** 20 if os.environ.get("ESPM_ELL_STREAM_MB") else _lib.ELL_STREAM_BYTES
        st.ell_stream = int(x_store == "ell" and self.x_bytes > limit)
        self._accum_done = False
        # autotune: at the first load_state the launch plans that apply to this problem are timed on the ingested image and
        # the fastest is kept (see autotune_plan)
        # "auto": the policy of the product (SmoothNMF.fit hands it down, bench.py too): timing the plans costs ~35 ms of device
        # time and a plan is worth a few per cent of an iteration - only fits of AUTOTUNE_MIN_ITERS iterations or more pay it back
        self._autotune = (int(max_iter) >= AUTOTUNE_MIN_ITERS) if autotune == "auto" else bool(autotune)
        self.plan_timings = None

        # ---- sharding -----------------------------------------------------------------------------------
        if self.sharded:
            from .sharding import ShardExchange
            self.exchange = ShardExchange(group, k, st.n_pad, st.ny, bool(st.grid_mode and self.lambda_L != 0.0), dev, lib=self.lib,
                                          stream_fn=_stream)
            if self.exchange.layout.nbytes != int(self.lib.espm_mu_shard_record_bytes(C.byref(st))):
                raise RuntimeError("record layout of espm_amd.sharding and libespm_mu disagree")

    # ------------------------------------------------------------------------------------------------
    def _build_ell(self, Xd, layout):
        """Sparse count store through the C ABI: dense 8-bit pixel-major copy (espm_mu_pack_x) -> espm_mu_ell_count /
        _plan / _fill.  Same result as espm_amd.ell.build (tests/test_gpu_updates.py::test_ell_builders_agree)."""
        st, dev = self.st, self.device
        i32 = dict(dtype=torch.int32, device=dev)
        x8 = torch.empty((self.p, st.n_pad), dtype=torch.uint8, device=dev)
        # (the channel-major copy beside it: the channel lists' fill reads it with 16-byte loads, include/espm_mu.h; ESPM_ELL_BUILD_CM=0: without, A/B)
        x8c = torch.empty((st.p_pad // _lib.PPAD, st.n_cm, _lib.PPAD), dtype=torch.uint8, device=dev) if os.environ.get("ESPM_ELL_BUILD_CM") != "0" else None
        self._check(self.lib.espm_mu_pack_x(_ptr(Xd), _lib.SRC_F64 if Xd.dtype == torch.float64 else _lib.SRC_F32,
                                 _lib.LAYOUT_PM if layout == "pm" else _lib.LAYOUT_CM, Xd.shape[1], self.n, self.p,
                                 _ptr(x8c) if x8c is not None else None, _ptr(x8), _lib.X_U8, st.n_pad, st.p_pad, _lib.PPAD, st.n_cm, _stream()))
        cnt_px = torch.empty((2, st.p_pad), **i32)                    # entries, elements equal to 1
        cnt_bc = torch.empty((2, st.nblk_w, st.n_cg * 64), **i32)
        klc = torch.empty(st.p_pad, dtype=torch.float32, device=dev)
        # (the lists' histograms of unit elements go from the count to the fill, which then walks X once instead of twice: include/espm_mu.h;
        #  ESPM_ELL_BUILD_HIST=0: without, A/B)
        hist = os.environ.get("ESPM_ELL_BUILD_HIST") != "0"
        bkt_px = torch.empty((st.p_pad, _lib.ELL_BUCKETS), dtype=torch.uint8, device=dev) if hist else None
        bkt_bc = torch.empty((st.nblk_w, st.n_cg * 64, _lib.ELL_BUCKETS), dtype=torch.uint8, device=dev) if hist else None
        self._check(self.lib.espm_mu_ell_count_hist(C.byref(st), _ptr(x8), _ptr(cnt_px), _ptr(cnt_bc), _ptr(klc),
                                                    _ptr(bkt_px) if hist else None, _ptr(bkt_bc) if hist else None, _stream()))
        chan_perm = torch.empty((st.nblk_w, st.n_cg * 64), **i32)
        pix_perm = torch.empty(st.p_pad, **i32)
        h_off = torch.empty(2 * (st.p_pad // 64) + 1, **i32)          # per group: first unit row, first general row
        w_off = torch.empty(2 * st.nblk_w * st.n_cg + 1, **i32)
        rows = torch.zeros(2, dtype=torch.int64, device=dev)
        self._check(self.lib.espm_mu_ell_plan(C.byref(st), _ptr(cnt_px), _ptr(cnt_bc), _ptr(chan_perm), _ptr(pix_perm), _ptr(h_off),
                                   _ptr(w_off), _ptr(rows), _stream()))
        rows_h, rows_w = (int(v) for v in rows.cpu())
        if max(rows_h, rows_w) * 64 >= 2 ** 31:
            raise ValueError("sparse count store: the lists exceed 2^31 dwords")
        ell_h = torch.zeros(max(rows_h, 1) * 64, **i32)
        ell_w = torch.zeros(max(rows_w, 1) * 64, **i32)
        st.x_cm = x8c.data_ptr() if x8c is not None else None
        try:
            self._check(self.lib.espm_mu_ell_fill_hist(C.byref(st), _ptr(x8), _ptr(chan_perm), _ptr(pix_perm), _ptr(h_off), _ptr(w_off),
                                                       _ptr(ell_h), _ptr(ell_w), _ptr(bkt_px) if hist else None, _ptr(bkt_bc) if hist else None, _stream()))
        finally:
            st.x_cm = None
        nnz = getattr(self, "_nnz_known", None)
        if nnz is None:
            nnz = int(torch.count_nonzero(x8))   # (padding channels hold zeros)
        torch.cuda.current_stream().synchronize()
        return dict(ell_h=ell_h, ell_h_off=h_off, klc=klc, pix_perm=pix_perm, ell_w=ell_w, ell_w_off=w_off, chan_perm=chan_perm, n_cg=st.n_cg,
                    nblk_w=st.nblk_w, nnz=nnz, entries_h=int(cnt_px[0].sum()), entries_w=int(cnt_bc[0].sum()), rows_h=rows_h,
                    rows_w=rows_w, unit_rows_h=int((h_off[1::2] - h_off[0:-1:2]).sum()),
                    unit_rows_w=int((w_off[1::2] - w_off[0:-1:2]).sum()))

    def _pad_h(self, H):
        Hh = np.asarray(H, dtype=np.float32)
        if Hh.shape != (self.k, self.p):
            raise ValueError(f"expected an array of shape {(self.k, self.p)}, got {Hh.shape}")
        out = torch.full((self.k, self.st.p_pad), -1.0, dtype=torch.float32, device=self.device)
        out[:, :self.p] = torch.from_numpy(np.ascontiguousarray(Hh)).to(self.device)
        return out

    def load_state(self, W, H):
        """Install (W, H) as the current state: builds GW, its column sums and the statistics of H.  numpy arrays, or torch
        tensors (already on the device: nothing crosses the host then)."""
        st = self.st

        def as_f32(a, shape, name):
            if isinstance(a, torch.Tensor):
                t = a.to(device=self.device, dtype=torch.float32)
            else:
                t = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))
            if tuple(t.shape) != shape:
                raise ValueError(f"{name} must be {shape}, got {tuple(t.shape)}")
            return t
        Wt = as_f32(W, (self.M, self.k), "W")
        Ht = as_f32(H, (self.k, self.p), "H")
        st.cur, st.it = 0, 0
        self._pending_finalize = None
        self._pending_tail = False
        self._accum_done = False
        self.hist.zero_()
        self.w[0].copy_(Wt)
        self.h[0][:, :self.p].copy_(Ht)
        self._check(self.lib.espm_mu_build_gw(C.byref(st), 0, _stream()))
        self._check(self.lib.espm_mu_hstat(C.byref(st), 0, _stream()))
        if self.sharded:
            self._globalize_hstat(0)
            self._exchange_halo_only(0)
        if self._autotune:
            self._autotune = False
            self.autotune_plan()

    PLANS = {0: "fused, units handed out dynamically", 2: "fused, fixed unit assignment", 1: "H-step and W accumulation as two launches"}

    def autotune_plan(self, iters=48, warm=4):
        """Times the launch plans of the iteration that apply to this problem - the fused kernel with dynamic or fixed
        assignment of its work units, the two-launch path (include/espm_mu.h: no_fused) - on the ingested image from the
        loaded state, keeps the fastest and restores the state.  Which plan wins depends on the component count (the fused
        kernel has fewer segments per list group from k = 6 on), the dose and the image size; at the headline problem the
        dynamic fused plan does.  ~(2 + 3) x iters iterations of device time, once per fit.  One GPU only."""
        st = self.st
        if self.sharded or self.frobenius or self.ell is None:
            return None
        keep = st.no_fused
        st.no_fused = 0
        if not bool(self.lib.espm_mu_fused_applies(C.byref(st))):
            st.no_fused = keep
            return None
        iters = int(os.environ.get("ESPM_AUTOTUNE_ITERS", iters))
        iters = min(iters, self.hist_len - st.it - 2 - warm)
        if iters < 4:   # (a fit of a few iterations: nothing to gain)
            st.no_fused = keep
            return None
        tensors = [self.w[0], self.w[1], self.h[0], self.h[1], self.gw_s, self.colsum_gw, self.hstat[0], self.hstat[1], self.hist,
                   self.w_scratch, self.hpart]
        saved = [t.clone() for t in tensors]
        state = (st.cur, st.it)

        def restore():
            for t, sv in zip(tensors, saved):
                t.copy_(sv)
            st.cur, st.it = state
        # The device clocks ramp for ~20 ms after the memory-bound ingest, and the ramp is worth more than
        # the plans differ by: timed one after the other, the plan that goes first loses (on some boxes of the pool the default
        # lost to the two-launch plan that way and the loop then ran 5 % slower).  So: a run-in first, then the plans in
        # interleaved rounds - every plan gets an early, a middle and a late slice.
        rounds = 3
        per = max(iters // rounds, 2)
        st.no_fused = 0
        self._check(self.lib.espm_mu_iterate(C.byref(st), min(2 * iters, self.hist_len - st.it - 2), 0, _stream()))
        restore()
        total = {plan: 0.0 for plan in self.PLANS}
        for _ in range(rounds):
            for plan in self.PLANS:
                st.no_fused = plan
                self._check(self.lib.espm_mu_iterate(C.byref(st), min(warm, 2), 0, _stream()))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._check(self.lib.espm_mu_iterate(C.byref(st), per, 0, _stream()))
                e1.record()
                e1.synchronize()
                total[plan] += e0.elapsed_time(e1) * 1e3
                restore()
        timings = {plan: total[plan] / (rounds * per) for plan in self.PLANS}
        best = min(timings, key=timings.get)
        # the plans are timed inside the clock ramp that follows the ingest, a few per cent of noise included: another plan
        # has to beat the default (fused, dynamic units: the steady-state winner wherever it applies) by 3 % to replace it
        if best != 0 and timings[best] > 0.97 * timings[0]:
            best = 0
        st.no_fused = best
        self.plan_timings = {self.PLANS[p]: v for p, v in timings.items()}
        self.plan = self.PLANS[best]
        return self.plan
    def set_G(self, G):
        """Replace G (physics model refreshed it, espm/estimators/base.py:388-390) and rebuild G W."""
        if self.m == 0:
            raise ValueError("the engine was built with G = identity")
        self._flush_finalize()
        Gh = np.ascontiguousarray(np.asarray(G, dtype=np.float32))
        if Gh.shape != (self.n, self.m):
            raise ValueError(f"G must stay {(self.n, self.m)}, got {Gh.shape}")
        self.g.copy_(torch.from_numpy(Gh))
        self.g_t[:, :self.n] = self.g.t()
        self.colsum_g.copy_(torch.from_numpy(np.asarray(G, dtype=np.float64).sum(axis=0).astype(np.float32)))
        self._gtg = None   # G^T G of the Frobenius W step (updates.py:31-36) belongs to the old G
        self._check(self.lib.espm_mu_build_gw(C.byref(self.st), self.st.cur, _stream()))

    def exchange_health(self):
        """Sharded engines: the number of bounded waits of the one-shot exchange that gave up, maximum over the ranks (0 on a
        healthy node; a peer that never delivers is counted, not waited for - the iterates are garbage from there on)."""
        if not self.sharded:
            return 0
        lost = torch.tensor([self.exchange.lost_peers()], dtype=torch.int64, device=self.device)
        torch.distributed.all_reduce(lost, op=torch.distributed.ReduceOp.MAX, group=self.group)
        return int(lost.item())

    def settle_exchange(self, iters=3):
        """Sharded engine on the one-shot transport: ``iters`` iterations from the loaded state with the real kernels (the
        in-launch, piece-wise exchange of espm_mu_shard_exchange_finish, not only the start-up self-test's post / wait), then
        the health check - jointly, on every rank - and, if a bounded wait gave up anywhere, the collective transport on every
        rank.  Returns the transport in effect.  The state has moved: the caller loads it again.  (A transport that stalls
        only later still raises from ``history``; this catches the configurations in which it never works - ranks sharing a
        device with whole-CU workgroups, a node without peer mapping.)"""
        if not self.sharded or self.exchange.ctx is None or self.frobenius:
            return self.exchange.transport if self.sharded else None
        n = max(0, min(int(iters), self.hist_len - self.st.it - 2))
        if n:
            self.iterate(n, final_loss=False)
        torch.cuda.synchronize()
        if self.exchange_health() > 0:
            self.use_collective_exchange()
        return self.exchange.transport

    def use_collective_exchange(self):
        """Replaces the one-shot exchange by the collective transport on every rank (call it on all of them); the state has to
        be loaded again afterwards."""
        from .sharding import ShardExchange
        torch.cuda.synchronize()
        tested = self.exchange.selftest_result
        self.exchange.close()
        self.exchange = ShardExchange(self.group, self.k, self.st.n_pad, self.st.ny, bool(self.st.grid_mode and self.lambda_L != 0.0),
                                      self.device, lib=self.lib, stream_fn=_stream, mode="collective")
        self.exchange.selftest_result = dict(tested or {}, transport="collective", fell_back_from="p2p")

    # ---- sharded helpers ---------------------------------------------------------------------------
    def _globalize_hstat(self, which):
        hs = self.hstat[which]
        torch.distributed.all_reduce(hs[:self.V.HS_MAX], group=self.group)
        torch.distributed.all_reduce(hs[self.V.HS_MAX:], op=torch.distributed.ReduceOp.MAX, group=self.group)

    def _gather_rows(self, vec):
        """(world, len) tensor of every rank's ``vec``, in rank order (a host-sequenced collective: linesearch, read-back)."""
        out = torch.empty((self.world,) + tuple(vec.shape), dtype=vec.dtype, device=vec.device)
        if torch.distributed.get_backend(self.group) == "gloo":
            parts = [torch.empty_like(vec, device="cpu") for _ in range(self.world)]
            torch.distributed.all_gather(parts, vec.cpu(), group=self.group)
            for r, part in enumerate(parts):
                out[r].copy_(part)
        else:
            torch.distributed.all_gather_into_tensor(out, vec.contiguous(), group=self.group)
        return out

    def _set_halo_from_records(self):
        top, bot = self.exchange.halo_offsets()
        base = self.exchange.recv_ptr
        self.st.halo_top = base + top if top is not None else None
        self.st.halo_bot = base + bot if bot is not None else None

    def _exchange_halo_only(self, which):
        """Boundary rows of h[which] to the neighbours (initial state only)."""
        self._check(self.lib.espm_mu_shard_pack(C.byref(self.st), which, C.c_void_p(self.exchange.send_ptr), _stream()))
        self.exchange.gather()
        self._set_halo_from_records()

    # ---- one iteration, granular (stop criteria / sharded) ---------------------------------------------
    def eval_current(self, advance_h=True):
        """H-step from the current state: fills history slot ``it`` with the loss pieces of the current
        state and (advance_h) leaves the new H in the other buffer.

        With advance_h the reduction of the H-step's records is deferred: ``finish_iteration`` folds it into its
        slab-reduction launch; any other consumer (``history``, ``step_*_only``, ...) flushes it first."""
        st = self.st
        self._flush_finalize(tail=False)
        if self.frobenius:
            self._frobenius_of_current()
        # the tail of the W update that produced this state (column sums of G W', rel_W) rides in this launch (tail_mode,
        # include/espm_mu.h) instead of costing one of its own
        carry = getattr(self, "_pending_tail", False)
        st.tail_mode = _lib.TAIL_RIDE if carry else 0
        self._pending_tail = False
        try:
            if advance_h:
                # (the W accumulation rides in the same launch where the fused kernel applies; finish_iteration then skips it)
                self._accum_done = (not self.frobenius) and bool(self.lib.espm_mu_fused_applies(C.byref(st)))
                if self._accum_done:
                    self._check(self.lib.espm_mu_step_hw(C.byref(st), st.cur, _stream()))
                else:
                    self._check(self.lib.espm_mu_step_h(C.byref(st), st.cur, 1, _stream()))
                self._pending_finalize = (st.cur, st.it)
            else:
                self._check(self.lib.espm_mu_loss_only(C.byref(st), st.cur, st.it, _stream()))
        finally:
            st.tail_mode = 0

    def _frobenius_of_current(self, rows=32768):
        """||X - G W H||_F^2 of the current state (espm/measures.py:350-384) into its history slot: residual in fp32 over
        blocks of pixels of the pixel-major X, sums in fp64."""
        cur = self.st.cur
        gw = self.w[cur] if self.m == 0 else self.g @ self.w[cur]          # (n, k)
        ht = self.h[cur][:, :self.p].t()                                      # (p, k)
        total = torch.zeros((), dtype=torch.float64, device=self.device)
        for lo in range(0, self.p, rows):
            r = self.x_pm[lo:lo + rows, :self.n] - ht[lo:lo + rows] @ gw.t()
            total += (r * r).sum(dtype=torch.float64)
        self.frob[self.st.it] = total

    def _finish_iteration_frobenius(self):
        """Frobenius W step (updates.py:31-36) with the H of ``eval_current``; flips the buffers."""
        st = self.st
        cur, slot = st.cur, st.it
        self._flush_finalize()
        work, scratch = self._l2_buffers()
        if self.m > 0 and getattr(self, "_gtg", None) is None:
            self._gtg = (self.g.double().t() @ self.g.double()).float().contiguous()
        s = _stream()
        gtg = _ptr(self._gtg) if self.m > 0 else None
        if self.sharded:
            # the statistics of this rank's new H block become global, its boundary rows travel to the neighbours, and the two
            # sums over the pixels the W step needs (X H^T, H H^T) are added over the ranks: the same W' on every rank
            self._globalize_hstat(1 - cur)
            if self.exchange.with_halo:
                self._exchange_halo_only(1 - cur)
            self._check(self.lib.espm_mu_l2_w_partials(C.byref(st), _ptr(work), _ptr(scratch), scratch.numel(), s))
            torch.distributed.all_reduce(self.a, group=self.group)
            torch.distributed.all_reduce(work[1], group=self.group)
            self._check(self.lib.espm_mu_l2_w_finish(C.byref(st), cur, gtg, _ptr(work), s))
        else:
            self._check(self.lib.espm_mu_l2_step_w(C.byref(st), cur, gtg, _ptr(work), _ptr(scratch), scratch.numel(), s))
        self._check(self.lib.espm_mu_build_gw(C.byref(st), 1 - cur, s))
        wn, wo = self.w[1 - cur].double(), self.w[cur].double()
        self.hist[slot + 1, _lib.HI_REL_W] = ((wn - wo).abs() / (wn + self.rel_tol * wn.mean())).max()   # base.py:323
        st.cur, st.it = 1 - cur, slot + 1

    def _flush_finalize(self, tail=True):
        pend = getattr(self, "_pending_finalize", None)
        if pend is not None:
            self._pending_finalize = None
            self._check(self.lib.espm_mu_h_finalize(C.byref(self.st), pend[0], pend[1], _stream()))
        if tail and getattr(self, "_pending_tail", False):   # no H-step will carry the tail of the last W update: a launch of its own
            self._pending_tail = False
            self._check(self.lib.espm_mu_w_update_tail(C.byref(self.st), 1 - self.st.cur, self.st.it - 1, _stream()))

    def finish_iteration(self):
        """W-step with the H produced by ``eval_current`` and the bookkeeping; flips the buffers."""
        st = self.st
        cur, slot = st.cur, st.it
        if slot + 1 >= self.hist_len:
            raise ValueError("history buffer exhausted: raise max_iter")
        if self.frobenius:
            return self._finish_iteration_frobenius()
        s = _stream()
        if not self._accum_done:   # (fused: the W accumulation of this H update rode in eval_current's launch, and h_t was not written)
            self._check(self.lib.espm_mu_w_accum(C.byref(st), s))
        self._accum_done = False
        ride = getattr(self, "_pending_finalize", None) == (cur, slot)   # the H-step's record reduction rides along
        if ride:
            self._pending_finalize = None
        else:
            self._flush_finalize()
        # sparse store, local W update: its tail is left to the next H-step's launch (eval_current) or to _flush_finalize
        defer = (self.ell is not None and self.pg_q is None and bool(self.lib.espm_mu_w_update_is_local(C.byref(st))))
        st.tail_mode = _lib.TAIL_DEFER if defer else 0
        if self.sharded and self.exchange.ctx is not None and ride:
            # one-shot transport, and the H-step's records are still the workspace's content (their reduction is pending): the
            # library's exchange launch(es), as in the batch loop (espm_mu_iterate_sharded) - slab reduction, record reduction,
            # granule exchange, W update.  (Once the records have been reduced, other launches may have reused the workspace -
            # the projected gradient's linesearch evaluates a loss in between - and the pieces below carry the statistics along.)
            self.exchange.seq.value += 1
            self._check(self.lib.espm_mu_shard_exchange_finish(C.byref(st), self.exchange.ctx, self.exchange.seq, cur, slot, s))
            self._set_halo_from_records()
        elif self.sharded:
            if ride:   # slab reduction + record reduction + this rank's record, one launch
                self._check(self.lib.espm_mu_w_reduce_pack(C.byref(st), cur, slot, C.c_void_p(self.exchange.send_ptr), s))
            else:
                self._check(self.lib.espm_mu_w_reduce(C.byref(st), s))
                self._check(self.lib.espm_mu_shard_pack(C.byref(st), 1 - cur, C.c_void_p(self.exchange.send_ptr), s))
            self.exchange.gather()
            # sum over the ranks + W update (one launch when W' needs nothing global, include/espm_mu.h)
            self._check(self.lib.espm_mu_shard_combine_finish(C.byref(st), C.c_void_p(self.exchange.recv_ptr), self.world, cur, slot, s))
            self._set_halo_from_records()
        else:
            self._check(self.lib.espm_mu_w_reduce_finish(C.byref(st), cur, slot, int(ride), s))
        st.tail_mode = 0
        self._pending_tail = defer
        st.cur, st.it = 1 - cur, slot + 1

    def linesearch_step(self, gamma):
        """Adapts sigma_L after an iteration (espm/estimators/smooth_nmf.py:376-381, surrogates.py:116-149): with Ht the
        H before the last update and H the current one, d = g(H, Ht) - 1/2 tr(H L H^T) (lambda_L = 1, as the reference
        calls it); gamma / 1.05 when d > 0, else gamma * 1.5.  Call after ``finish_iteration``; returns the new gamma
        (already in effect for the next H-step).  One host synchronisation."""
        st = self.st
        if st.it < 1:
            raise ValueError("linesearch needs a completed iteration")
        self._flush_finalize()
        if getattr(self, "_ls_out", None) is None:
            self._ls_out = torch.zeros(4 + self.V.KP, dtype=torch.float64, device=self.device)
        if self.sharded:
            # every rank: the terms of its block of image rows (the old H's boundary rows of the neighbours are still in the
            # records of the exchange before the last one), summed over the ranks in rank order - the same d, hence the same
            # gamma, on every rank
            if st.grid_mode and self.lambda_L == 0.0:
                raise NotImplementedError("linesearch of a sharded image with lambda_L = 0: the boundary rows of H are not exchanged")
            top, bot = self.exchange.halo_offsets()
            base = self.exchange.prev_recv_ptr
            self._check(self.lib.espm_mu_linesearch_terms_sharded(C.byref(st), 1 - st.cur, st.cur,
                                                                  C.c_void_p(base + top) if top is not None else None,
                                                                  C.c_void_p(base + bot) if bot is not None else None,
                                                                  _ptr(self._ls_out), _stream()))
            parts = self._gather_rows(self._ls_out).cpu().numpy()
            t = parts[0].copy()
            for r in range(1, self.world):
                t += parts[r]
        else:
            self._check(self.lib.espm_mu_linesearch_terms(C.byref(st), 1 - st.cur, st.cur, _ptr(self._ls_out), _stream()))
            t = self._ls_out.cpu().numpy()
        if st.h_rule == 1:     # quadratic surrogate: sigma ||Ht - H||^2 (surrogates.py:6-58)
            t3 = float(t[3])
        else:                  # sigma sum_k max_j H_kj sum_j dgkl(Ht, H) (surrogates.py:65-114)
            maxh = self.hstat[st.cur][self.V.HS_MAX:self.V.HS_MAX + self.k].cpu().numpy()   # max_j H[k, j] of the new H (global)
            t3 = float((maxh * t[4:4 + self.k]).sum())
        d = 0.5 * (2.0 * t[1] - t[0] + float(gamma) * t3) - 0.5 * t[2]
        gamma = float(gamma) / 1.05 if d > 0 else float(gamma) * 1.5
        st.sigma_l = gamma
        return gamma

    # ---- linesearch of the projected gradient (espm/estimators/smooth_nmf.py:382-401, :438-447) -------------------------
    def _loss_sum_of_slot(self, slot, local_stats=False):
        """Un-averaged loss from a history row.  Sharded image: the sums over the pixels are added over the ranks; with
        ``local_stats`` the row was evaluated from the statistics of this rank's H block alone (an H that has not been through
        the exchange yet), so its sum-of-GWH term is a per-rank piece too."""
        row = self.hist[slot]
        if self.sharded:
            cols = [_lib.HI_KLX, _lib.HI_REG, _lib.HI_LAP] + ([_lib.HI_SUMY] if local_stats else [])
            parts = self._gather_rows(row[cols].contiguous()).cpu().numpy()
            tot = parts[0].copy()
            for r in range(1, self.world):
                tot += parts[r]
            row = row.cpu().numpy().copy()
            row[cols] = tot
        else:
            row = row.cpu().numpy()
        return float(row[_lib.HI_KLX] + row[_lib.HI_SUMY] - self.xscale * self.sum_x + row[_lib.HI_REG]
                     + 0.5 * self.lambda_L * row[_lib.HI_LAP])

    def _pg_term(self, t, which):
        """The quadratic term the H step (which = 0: a sum over the pixels, added over the ranks of a sharded image) or the W
        step (1: W is replicated) of iteration t left in pg_q."""
        if which == 0 and self.sharded:
            parts = self._gather_rows(self.pg_q[t, 0:1].contiguous()).cpu().numpy()
            return float(sum(float(v[0]) for v in parts))
        return float(self.pg_q[t, which])

    def pg_linesearch_h(self, gamma_h):
        """After the H-step from state t (``eval_current(True)``), before the W-step: d = f(W, Ht) + <H - Ht, grad> +
        gamma ||H - Ht||^2 - f(W, H) with the losses not averaged; gamma_H / 1.05 when d > 0, else gamma_H * 1.5 (in effect
        from the next H-step).  Costs one loss-only pass over X (the loss of (W_t, H_{t+1})) and two host synchronisations."""
        if self.pg_q is None:
            raise NotImplementedError("the projected gradient's linesearch needs an engine built with h_rule=2")
        st = self.st
        t = st.it
        f_xt = float(self.history(upto=t, average=False)["loss"][t])   # (also reduces the H-step's records: pg_q[t][0])
        scratch = self.hist_len - 1
        self._check(self.lib.espm_mu_loss_only(C.byref(st), 1 - st.cur, scratch, _stream()))   # state (W_t, H_{t+1})
        self._pg_f_mid = self._loss_sum_of_slot(scratch, local_stats=True)   # (sharded: H_{t+1}'s statistics are still per rank)
        d = f_xt + self._pg_term(t, 0) - self._pg_f_mid
        gamma_h = float(gamma_h) / 1.05 if d > 0 else float(gamma_h) * 1.5
        st.sigma_l = gamma_h
        return gamma_h

    def pg_linesearch_w(self, gamma_w):
        """After the state the W-step produced has been evaluated: the same test with f(Wt, H), the W-step's quadratic term
        and f(W, H); the new gamma_W is in effect from the next W-step."""
        st = self.st
        t = st.it
        f_x = float(self.history(upto=t, average=False)["loss"][t])
        d = self._pg_f_mid + self._pg_term(t, 1) - f_x
        gamma_w = float(gamma_w) / 1.05 if d > 0 else float(gamma_w) * 1.5
        st.pg_gamma_w = gamma_w
        return gamma_w

    def iterate(self, n_iter, final_loss=True):
        """``n_iter`` iterations without host synchronisation (no stop criterion)."""
        st = self.st
        if st.it + n_iter + 1 > self.hist_len:
            raise ValueError("history buffer exhausted: raise max_iter")
        self._flush_finalize()
        self._accum_done = False   # (a pending fused H update is simply redone by the loop)
        if not self.sharded and not self.frobenius:
            self._check(self.lib.espm_mu_iterate(C.byref(st), int(n_iter), int(bool(final_loss)), _stream()))
        elif self.sharded and self.exchange.ctx is not None and not self.frobenius:
            # sharded, one-shot exchange: the whole batch is enqueued by the library (no host-side collective per iteration)
            self._check(self.lib.espm_mu_iterate_sharded(C.byref(st), self.exchange.ctx, C.byref(self.exchange.seq), int(n_iter),
                                                         int(bool(final_loss)), _stream()))
        else:
            for _ in range(int(n_iter)):
                self.eval_current(True)
                self.finish_iteration()
            if final_loss:
                self.eval_current(False)

    def iterate_timed(self, n_iter):
        """``n_iter`` iterations of the library's own loop (``espm_mu_iterate``: enqueued from C, the device never waits for the
        host) with HIP events on the launch stream around the launches of every iteration: returns (first_us, rest_us) - the
        iteration's first launch (the H update; fused: with the W accumulation) and what follows up to the new W.  One GPU, not
        for the Frobenius fit.  Synchronises."""
        st = self.st
        n_iter = int(n_iter)
        if st.it + n_iter + 1 > self.hist_len:
            raise ValueError("history buffer exhausted: raise max_iter")
        if self.sharded or self.frobenius:
            raise NotImplementedError("iterate_timed: the unsharded multiplicative loop (timed_iterations serves a sharded engine)")
        self._flush_finalize()
        self._accum_done = False
        first, rest = (C.c_float * n_iter)(), (C.c_float * n_iter)()
        self._check(self.lib.espm_mu_iterate_timed(C.byref(st), n_iter, first, rest, _stream()))
        return np.array(first[:], dtype=np.float64) * 1e3, np.array(rest[:], dtype=np.float64) * 1e3

    def timed_iterations(self, n_iter):
        """``n_iter`` iterations with HIP events between the launches, for diagnosis (bench.py prints the per-rank figures of a
        multi-GPU run): returns (half_steps_us, w_step_us), two float arrays of length n_iter.  ``half_steps``: the launch(es)
        that update this rank's H block and accumulate its R H'^T - local work only; ``w_step``: what follows up to the new W -
        slab reduction, the record exchange (this is where a rank WAITS for its peers) and the W update.  The same entry
        points, in the same order, as ``iterate`` (for the one-shot transport the two launches espm_mu_iterate_sharded
        enqueues per iteration), sequenced from Python; the events cost ~2 us per iteration."""
        st = self.st
        n_iter = int(n_iter)
        if st.it + n_iter + 1 > self.hist_len:
            raise ValueError("history buffer exhausted: raise max_iter")
        if self.frobenius:
            raise NotImplementedError("timed_iterations: not for the Frobenius fit")
        self._flush_finalize()
        self._accum_done = False
        s = _stream()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n_iter)]
        oneshot = self.sharded and self.exchange.ctx is not None
        if oneshot:
            defer = self.ell is not None and self.pg_q is None and bool(self.lib.espm_mu_w_update_is_local(C.byref(st)))
            pending = False
            for e0, e1, e2 in ev:
                cur, slot = st.cur, st.it
                st.tail_mode = _lib.TAIL_RIDE if pending else 0
                e0.record()
                try:
                    self._check(self.lib.espm_mu_step_hw(C.byref(st), cur, s))
                finally:
                    st.tail_mode = 0
                e1.record()
                self.exchange.seq.value += 1
                st.tail_mode = _lib.TAIL_DEFER if defer else 0
                try:
                    self._check(self.lib.espm_mu_shard_exchange_finish(C.byref(st), self.exchange.ctx, self.exchange.seq, cur, slot, s))
                finally:
                    st.tail_mode = 0
                e2.record()
                pending = defer
                self._set_halo_from_records()
                st.cur, st.it = 1 - cur, slot + 1
            self._pending_tail = pending
        else:
            for e0, e1, e2 in ev:
                e0.record()
                self.eval_current(True)
                e1.record()
                self.finish_iteration()
                e2.record()
        torch.cuda.synchronize()
        return (np.array([a.elapsed_time(b) * 1e3 for a, b, _ in ev]), np.array([b.elapsed_time(c) * 1e3 for _, b, c in ev]))

    # ---- single half steps for the module-level functions ----------------------------------------------
    def _l2_buffers(self):
        if getattr(self, "_l2_work", None) is None:
            self._l2_work = torch.zeros((2, self.V.KP, self.V.KP), dtype=torch.float32, device=self.device)
            self._l2_scratch = torch.zeros(64 * self.V.KP * self.V.KP, dtype=torch.float64, device=self.device)
        return self._l2_work, self._l2_scratch

    def step_h_only(self, l2=False):
        st = self.st
        self._flush_finalize()
        if l2:  # Frobenius branch, updates.py:109-118
            work, scratch = self._l2_buffers()
            self._check(self.lib.espm_mu_l2_step_h(C.byref(st), st.cur, _ptr(work), _ptr(scratch), scratch.numel(), _stream()))
            self._check(self.lib.espm_mu_h_finalize(C.byref(st), st.cur, st.it, _stream()))
            return self._h_numpy(1 - st.cur)
        self._check(self.lib.espm_mu_step_h(C.byref(st), st.cur, 1, _stream()))
        self._check(self.lib.espm_mu_h_finalize(C.byref(st), st.cur, st.it, _stream()))
        return self._h_numpy(1 - st.cur)

    def step_w_only(self, l2=False):
        """W update using the CURRENT H (its transposed copy is refreshed first)."""
        st = self.st
        cur = st.cur
        self._flush_finalize()
        self.h_t.zero_()
        self.h_t[:, :self.k].copy_(self.h[cur][:, :self.p].t())
        s = _stream()
        if l2:  # Frobenius branch, updates.py:31-36
            work, scratch = self._l2_buffers()
            gtg = None
            if self.m > 0:
                gtg = (self.g.double().t() @ self.g.double()).float().contiguous()   # G^T G (m, m): a constant of the fit
            self._check(self.lib.espm_mu_l2_step_w(C.byref(st), cur, _ptr(gtg) if gtg is not None else None, _ptr(work), _ptr(scratch),
                                        scratch.numel(), s))
            return self.w[1 - cur].cpu().numpy()
        self._check(self.lib.espm_mu_w_accum(C.byref(st), s))
        self._check(self.lib.espm_mu_w_reduce(C.byref(st), s))
        self._check(self.lib.espm_mu_w_finish(C.byref(st), cur, cur, -1, s))
        return self.w[1 - cur].cpu().numpy()

    # ---- read-back -----------------------------------------------------------------------------------------
    def _h_numpy(self, which):
        return self.h[which][:, :self.p].cpu().numpy()

    def get_W(self):
        return self.w[self.st.cur].cpu().numpy()

    def get_H(self):
        return self._h_numpy(self.st.cur)

    def bad_count(self):
        self._flush_finalize()
        return float(self.hist[:self.st.it + 1, _lib.HI_BAD].sum().item())

    def history(self, upto=None, average=True):
        """Loss pieces of states 0..upto (inclusive) assembled like SmoothNMF.loss
        (espm/estimators/smooth_nmf.py:457-475): returns dict of float64 arrays."""
        upto = self.st.it if upto is None else upto
        self._flush_finalize()
        hist = self.hist[:upto + 1].clone()
        if self.sharded and self.exchange.ctx is not None and self.exchange_health() > 0:
            raise _lib.LostPeerError("the record exchange between the ranks lost a peer (a bounded wait gave up): the iterates since "
                                 "then are not valid; ESPM_XCHG=collective selects the collective transport")
        if self.sharded:
            sums = hist[:, [_lib.HI_KLX, _lib.HI_REG, _lib.HI_LAP, _lib.HI_BAD]].contiguous()
            torch.distributed.all_reduce(sums, group=self.group)
            relh = hist[:, _lib.HI_REL_H].contiguous()
            torch.distributed.all_reduce(relh, op=torch.distributed.ReduceOp.MAX, group=self.group)
            hist[:, _lib.HI_KLX], hist[:, _lib.HI_REG] = sums[:, 0], sums[:, 1]
            hist[:, _lib.HI_LAP], hist[:, _lib.HI_BAD] = sums[:, 2], sums[:, 3]
            hist[:, _lib.HI_REL_H] = relh
        h = hist.cpu().numpy()
        numel = float(self.n) * float(self.p_total) if average else 1.0
        kl = (h[:, _lib.HI_KLX] + h[:, _lib.HI_SUMY] - self.xscale * self.sum_x) / numel
        if self.frobenius:  # base.py:197-198
            frob = self.frob[:upto + 1].clone()
            if self.sharded:
                torch.distributed.all_reduce(frob, group=self.group)
            kl = 0.5 * frob.cpu().numpy() / numel
        reg = h[:, _lib.HI_REG] / numel
        lap = 0.5 * self.lambda_L * h[:, _lib.HI_LAP] / numel
        return dict(loss=kl + reg + lap, kl=kl, reg=reg, lap=lap, rel_W=h[:, _lib.HI_REL_W],
                    rel_H=h[:, _lib.HI_REL_H], bad=h[:, _lib.HI_BAD])

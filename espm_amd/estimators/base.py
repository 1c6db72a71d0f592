"""sklearn-style NMF estimator base with the reference's surface (espm/estimators/base.py:20-529).

The fit loop of the reference (base.py:313-394) calls ``_iteration`` + ``loss`` on numpy arrays
once per multiplicative-update iteration.  Here the loop drives a device engine
(:class:`espm_amd.engine.MUEngine`): X, W, H stay in HBM, every iteration is a short chain of HIP
kernels, and the loss of a state is produced by the H-step that starts the NEXT iteration (it
forms the same G W H), so the reference's per-iteration bookkeeping costs no extra pass over X.
Stop rules, attribute names, printed messages and return conventions follow the reference.
"""
from __future__ import annotations

import os
import time
from abc import ABC, abstractmethod

import numpy as np
from sklearn.base import BaseEstimator, TransformerMixin
from sklearn.utils.validation import check_is_fitted, validate_data

from espm_amd.conf import log_shift
from espm_amd.estimators.updates import initialize_algorithms
from espm_amd.utils import create_laplacian_matrix, identity_laplacian, rescaled_DH


def normalization_factor(X, nc):
    """espm/estimators/base.py:16-18."""
    m = np.mean(X)
    return nc / (m * X.shape[0])


# X with at least this many entries is uploaded once and prepared on the device (see fit_transform)
_DEVICE_PREP_MIN_SIZE = 4_000_000


def _is_physical_model(G):
    return G is not None and not isinstance(G, np.ndarray) and hasattr(G, "NMF_update")


def _upload_with_scans(host, device, log_shift):
    """Uploads a C-contiguous host array in row chunks and has the device scan every chunk while the next one is on the bus: the
    passes the reference makes over X before its loop (finiteness - validate_data was told to skip it -, sign, line sums for the
    empty-line test base.py:519-528, the mean for `normalize`, const_KL_ = sum(X log X - X) in fp64, base.py:200-201) cost 16 ms
    as passes over the uploaded image; behind the 38 ms upload (a blocking copy from pageable memory per chunk, the device idle
    otherwise) they cost nothing.  Returns the device array and the scans' results as device tensors (nothing is read back here):
    row_sum / col_sum of the array as it lies in memory, `bad` = [non-finite entries, NaNs, negative entries], s1 = sum x and
    s2 = sum x log(max(x, log_shift)), both fp64; `facts` = [entries that are not integers, non-zero entries, largest entry]."""
    import threading
    import torch
    rows, cols = host.shape
    out = torch.empty(host.shape, dtype=torch.from_numpy(host[:0]).dtype, device=device)
    # 128 MB chunks, the last one halved down to 32 MB: the scans of the LAST chunk are what the fit waits for after the bus has
    # gone quiet - 3.3 ms behind 256 MB chunks, 1.4 behind 128, 0.3 behind 32, while many small chunks cost the bus a little
    # (upload 40.2 ms in 256 MB chunks, 41.2 in 32 MB ones; profiles/r05z_fit_timing_chunk*.log)
    row_bytes = max(1, cols * host.itemsize)
    step = max(1, (int(os.environ.get("ESPM_UPLOAD_CHUNK_MB", "128")) << 20) // row_bytes)
    f64 = dict(dtype=torch.float64, device=device)
    row_sum = torch.empty(rows, **f64)
    col_sum = torch.zeros(cols, **f64)
    bad = torch.zeros(3, dtype=torch.int64, device=device)
    s1, s2 = torch.zeros((), **f64), torch.zeros((), **f64)
    facts = torch.zeros(3, **f64)                                # entries that are not integers, non-zero entries, the largest entry
    chunks = [(a, min(rows, a + step)) for a in range(0, rows, step)]
    small = max(1, (32 << 20) // row_bytes)
    while len(chunks) > 1 and chunks[-1][1] - chunks[-1][0] >= 2 * small and os.environ.get("ESPM_UPLOAD_TAIL", "1") != "0":
        a, b = chunks.pop()
        mid = a + (b - a + 1) // 2
        chunks += [(a, mid), (mid, b)]       # (the first half stays as it is, the second is looked at again)
    # the copies go back to back from a thread of their own (a blocking copy from pageable memory per chunk, the GIL released
    # inside it); this thread queues the scans of a chunk as soon as it has arrived
    main = torch.cuda.current_stream(device)
    side = torch.cuda.Stream(device=device)
    side.wait_stream(main)                                        # (the allocator may hand out memory with work still queued on it)
    arrived = [threading.Event() for _ in chunks]
    done = [torch.cuda.Event() for _ in chunks]
    err = []

    def copier():
        try:
            with torch.cuda.stream(side):
                for i, (a, b) in enumerate(chunks):
                    out[a:b].copy_(torch.from_numpy(host[a:b]))
                    done[i].record(side)
                    arrived[i].set()
        except BaseException as e:   # noqa: BLE001 - re-raised below
            err.append(e)
            for ev in arrived:
                ev.set()
    th = threading.Thread(target=copier, daemon=True)
    th.start()
    for i, (a, b) in enumerate(chunks):
        arrived[i].wait()
        if err:
            break
        main.wait_event(done[i])
        x = out[a:b]
        fin = torch.isfinite(x)
        bad += torch.stack(((~fin).sum(), torch.isnan(x).sum(), (x < 0).sum()))
        xd = x.to(torch.float64)
        rs = xd.sum(dim=1)
        row_sum[a:b] = rs
        col_sum += xd.sum(dim=0)
        s1 += rs.sum()
        s2 += (xd * torch.log(xd.clamp_min(log_shift))).sum()
        # (what the engine's choice of a store asks of X - integer counts up to 255, how many non-zero: engine.py - while the data pass by)
        facts[:2] += torch.stack(((x != x.round()).sum(), (x != 0).sum())).to(torch.float64)
        facts[2] = torch.maximum(facts[2], x.max().to(torch.float64))
        del x, fin, xd, rs
    th.join()
    if err:
        raise err[0]
    return out, dict(row_sum=row_sum, col_sum=col_sum, bad=bad, s1=s1, s2=s2, facts=facts)


class _HostCopy:
    """``X_`` of a large fit in the making.  The reference keeps its own copy of the data (remove_zeros_lines copies,
    base.py:519-528; normalize scales it, base.py:264-267): at 2048 x 512^2 fp32 those host passes are 0.2-0.4 s of a fit whose
    200 iterations take 0.03 s.  They run on worker threads (numpy releases the GIL) while the device initialises and
    iterates; the fit joins them at its end.  Until then this object stands in for the array: shape and dtype are known at once,
    anything else (``__array__``, ``.T``, indexing ...) waits for the copy.

    ``finish`` hands over what the device scans found, ``start`` lets the copy run - ONE pass over row blocks on a few threads:
    copy, scale, fill - behind the upload (overlapping the upload the two halved each other's host bandwidth)."""

    THREADS = 4

    class _State:   # what the worker holds (not the stand-in itself: a stand-in nobody references any more lets its worker go)
        __slots__ = ("src", "layout", "params", "go", "out", "err")

    def __init__(self, src, layout):
        import threading
        self.shape, self.dtype, self.size, self.ndim = src.shape, src.dtype, src.size, src.ndim
        st = self._st = _HostCopy._State()
        st.src, st.layout = src, layout
        st.params = None            # (pixel mask, channel mask, fill, scale), set by finish()
        st.go, st.out, st.err = threading.Event(), None, None
        self._thread = threading.Thread(target=_HostCopy._run, args=(st,), daemon=True)
        self._thread.start()

    def __del__(self):
        st = self.__dict__.get("_st")
        if st is not None and not st.go.is_set():   # never started (a fit that raised before its loop): nothing to copy for
            st.params = None
            st.go.set()

    @staticmethod
    def _run(st):
        import threading
        try:
            st.go.wait()
            if st.params is None:   # cancelled
                st.src = None
                return
            zp, zc, fill, scale = st.params
            # the memory as it lies (for a pixel-major input: its transposed view), not a strided gather
            src = st.src if st.layout == "cm" else st.src.T
            st.src = None
            base = np.empty(src.shape, dtype=src.dtype)
            row_mask, col_mask = (zc, zp) if st.layout == "cm" else (zp, zc)   # masks over the rows / columns of `base`
            # an empty line holds fill, then everything is scaled (base.py:519-528, :264-267): fill * scale in the array's precision
            filled = None
            if zp is not None:
                filled = np.full(1, fill, dtype=src.dtype)
                if scale is not None:
                    np.multiply(filled, scale, out=filled)
            errs = []

            def block(a, b):
                try:
                    if scale is not None:
                        np.multiply(src[a:b], scale, out=base[a:b])
                    else:
                        np.copyto(base[a:b], src[a:b])
                    if filled is not None:
                        base[a:b][:, col_mask] = filled[0]
                        base[a:b][row_mask[a:b]] = filled[0]
                except BaseException as e:  # noqa: BLE001
                    errs.append(e)

            nt = max(1, min(_HostCopy.THREADS, src.shape[0]))
            edges = np.linspace(0, src.shape[0], nt + 1).astype(int)
            helpers = [threading.Thread(target=block, args=(int(edges[i]), int(edges[i + 1])), daemon=True) for i in range(1, nt)]
            for t in helpers:
                t.start()
            block(int(edges[0]), int(edges[1]))
            for t in helpers:
                t.join()
            if errs:
                raise errs[0]
            st.out = base if st.layout == "cm" else base.T
        except BaseException as e:  # noqa: BLE001 - re-raised by result()
            st.err = e

    def finish(self, zp=None, zc=None, fill=None, scale=None):
        """What the device scans found: empty pixels / channels to fill (base.py:519-528), the normalisation factor."""
        self._st.params = (zp, zc, fill, scale)

    def start(self):
        """Lets the copy run (once the upload is through and the scans of X have said what to fill; at the latest when the fit
        enters its iteration loop)."""
        self._st.go.set()

    def cancel(self):
        self._st.params = None
        self._st.go.set()

    def result(self):
        self._st.go.set()   # (asked for before the loop - e.g. a least-squares initialisation reads X_ - or never started)
        self._thread.join()
        if self._st.err is not None:
            raise self._st.err
        return self._st.out

    def __array__(self, dtype=None, copy=None):
        out = self.result()
        return out if dtype is None else out.astype(dtype, copy=False)

    def __getattr__(self, name):   # (only reached for what the stand-in does not have: .T, .sum, ...)
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.result(), name)

    def __getitem__(self, key):
        return self.result()[key]


class _Shard:
    """Pixel-row sharding of ONE fit over the ranks of a process group (espm_amd/sharding.py): every rank
    runs the same script on the same X; its engine holds a contiguous block of image rows (X and H sharded, W / G replicated)
    and the results are assembled on every rank."""

    def __init__(self, group, shape_2d, p):
        import torch.distributed as dist
        from espm_amd.sharding import split_rows
        self.group, self.world, self.rank = group, dist.get_world_size(group), dist.get_rank(group)
        self.src = dist.get_global_rank(group, 0)
        if shape_2d is not None:
            nx, ny = int(shape_2d[0]), int(shape_2d[1])
            if nx * ny != p:
                raise ValueError(f"shape_2d {shape_2d} does not match the {p} pixels of X")
            blocks = [split_rows(nx, self.world, r) for r in range(self.world)]
            self.counts = [rows * ny for _, rows in blocks]
            row0, rows = blocks[self.rank]
            self.sl, self.shape_2d = slice(row0 * ny, (row0 + rows) * ny), (rows, ny)
        else:   # no image grid (L = identity, base.py:289-291): any contiguous split of the pixels
            blocks = [split_rows(p, self.world, r) for r in range(self.world)]
            self.counts = [rows for _, rows in blocks]
            row0, rows = blocks[self.rank]
            self.sl, self.shape_2d = slice(row0, row0 + rows), None

    def combine_scans(self, scans, layout):
        """The upload's scans of this rank's block (_upload_with_scans) turned into the image's: counts and sums added over the
        ranks, the largest entry their maximum, the channel sums added (the rows of the array as it lies in the channel-major
        layout, its columns in the pixel-major one); the pixel sums stay the block's."""
        import torch
        dist = torch.distributed
        ch = "row_sum" if layout == "cm" else "col_sum"
        vec = torch.cat((scans["bad"].to(torch.float64), scans["s1"].view(1), scans["s2"].view(1), scans["facts"][:2]))
        dist.all_reduce(vec, group=self.group)
        xmax = scans["facts"][2:3].clone()
        dist.all_reduce(xmax, op=dist.ReduceOp.MAX, group=self.group)
        chs = scans[ch].clone()
        dist.all_reduce(chs, group=self.group)
        out = dict(scans)
        out["bad"] = vec[:3].round().to(torch.int64)
        out["s1"], out["s2"] = vec[3], vec[4]
        out["facts"] = torch.cat((vec[5:7], xmax))
        out[ch] = chs
        return out

    def agree(self, ok, what, err=None):
        """Every rank learns whether every rank succeeded (all-reduce MIN of a flag) before anybody raises: the failing rank re-raises
        its own exception, the others a RuntimeError naming the step - nobody is left in the next collective waiting for a rank that
        has gone."""
        import torch
        dev = f"cuda:{torch.cuda.current_device()}" if torch.distributed.get_backend(self.group) != "gloo" else "cpu"
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=self.group)
        if err is not None:
            raise err
        if int(flag.item()) == 0:
            raise RuntimeError(f"sharded fit: {what} failed on another rank")

    def cols(self, a):
        """This rank's columns of an (.., p) array (None stays None)."""
        return None if a is None else a[..., self.sl]

    def broadcast(self, arrays, device):
        """Rank 0's arrays on every rank (the initial W, H, G: bit-identical starts whatever the ranks' own init gave)."""
        import torch
        out = []
        for a in arrays:
            t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
            torch.distributed.broadcast(t, src=self.src, group=self.group)
            out.append(t.cpu().numpy())
        return out

    def gather_cols(self, local):
        """(k, p) numpy array from every rank's (k, p_local) device tensor, in rank order."""
        import torch
        width = max(self.counts)
        mine = torch.zeros((local.shape[0], width), dtype=local.dtype, device=local.device)
        mine[:, :local.shape[1]] = local
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        torch.distributed.all_gather(parts, mine, group=self.group)
        return np.concatenate([part[:, :c].cpu().numpy() for part, c in zip(parts, self.counts)], axis=1)


class NMFEstimator(ABC, TransformerMixin, BaseEstimator):
    """Abstract NMF estimator, X (n, p) ~ G (n, m) W (m, k) H (k, p); parameters and attributes as in
    espm/estimators/base.py:68-152."""

    loss_names_ = ["KL_div_loss"]
    const_KL_ = None

    def __init__(self, n_components=2, init=None, tol=1e-4, max_iter=200, random_state=None, verbose=1, debug=False,
                 l2=False, G=None, shape_2d=None, normalize=False, log_shift=log_shift, eval_print=10, true_D=None,
                 true_H=None, fixed_H=None, fixed_W=None, hspy_comp=False, no_stop_criterion=False, simplex_H=False,
                 simplex_W=True):
        self.n_components = n_components
        self.init = init
        self.tol = tol
        self.max_iter = max_iter
        self.random_state = random_state
        self.verbose = verbose
        self.log_shift = log_shift
        self.debug = debug
        self.l2 = l2
        self.G = G
        self.shape_2d = shape_2d
        self.eval_print = eval_print
        self.true_D = true_D
        self.true_H = true_H
        self.fixed_H = fixed_H
        self.fixed_W = fixed_W
        self.hspy_comp = hspy_comp
        self.normalize = normalize
        self.no_stop_criterion = no_stop_criterion
        self.simplex_H = simplex_H
        self.simplex_W = simplex_W

    def __sklearn_tags__(self):
        tags = super().__sklearn_tags__()
        tags.input_tags.positive_only = True
        return tags

    def _more_tags(self):
        return {"requires_positive_X": True}

    def __getstate__(self):
        state = super().__getstate__()
        state = dict(state)
        for key in ("_engine", "_truth_engine", "_shard_group", "_shard"):  # device buffers / ctypes pointers / process groups are not picklable
            state.pop(key, None)
        return state

    # ---- one fit over the GPUs of a node ----------------------------------------------------------------
    def shard(self, group):
        """Run the fits of this estimator sharded by image rows over the ranks of ``group`` (a ``torch.distributed`` process
        group with one GPU per rank; ``None`` switches it off).  Every rank calls ``fit`` / ``fit_transform`` with the same
        arguments - the same script under ``torchrun`` - and gets the same results: W replicated (bit-identical across the
        ranks), H and the returned arrays assembled from the ranks' blocks.  No reference analogue.
        Returns self."""
        self._shard_group = group
        return self

    # ---- hooks implemented by the concrete estimator -------------------------------------------------
    @abstractmethod
    def _iteration(self, W, H):
        pass

    def _engine_kwargs(self):
        return {}

    def _detailed(self, lkl, reg, lap):
        return [lkl]

    # ---- loss -----------------------------------------------------------------------------------------
    def _make_engine(self, X_fixed, xscale, G, filled_channels=None, filled_pixels=None, layout="cm", autotune=False, shard=None, x_facts=None,
                     x_local=False):
        """The device engine of a fit; with ``shard`` (a _Shard) X_fixed is the WHOLE image and the engine takes this rank's
        block of it - or, with ``x_local``, that block already (then ``filled_pixels`` is the block's mask too)."""
        from espm_amd.engine import MUEngine
        shape_2d, fixed_H, group = self.shape_2d, self.fixed_H, None
        if shard is not None:
            if not x_local:
                X_fixed = X_fixed[shard.sl] if layout == "pm" else X_fixed[:, shard.sl]
                filled_pixels = shard.cols(filled_pixels)
            shape_2d, group = shard.shape_2d, shard.group
            fixed_H = shard.cols(np.asarray(fixed_H)) if fixed_H is not None else None
            x_facts = None      # (they describe the whole image)

        rows = None
        if self.physics_model_ is not None and self.simplex_W:
            rows = self.physics_model_.NMF_simplex()
        # (the Bregman W update has no simplex branch, updates.py:40-48: algo="bmd" ignores simplex_W there, like the reference)
        simplex_W = self.simplex_W and getattr(self, "algo", None) != "bmd"
        if self.l2:
            # Frobenius data term (only kept by SmoothNMF with algo="l2_surrogate", smooth_nmf.py:223-237): the W step has no
            # simplex there (updates.py:31-36); the engine's Frobenius mode takes the scaled X itself
            simplex_W = False
            if xscale != 1.0:
                X_fixed, xscale, x_facts = X_fixed * xscale, 1.0, None
        # (and the projected-gradient W step of a fit is called without fixed_W, smooth_nmf.py:430-437)
        fixed_W = None if getattr(self, "algo", None) == "projected_gradient" else self.fixed_W
        return MUEngine(X_fixed, self.n_components, G=G, shape_2d=shape_2d, simplex_H=self.simplex_H,
                        simplex_W=simplex_W, log_shift=self.log_shift, tol=self.tol, fixed_H=fixed_H,
                        fixed_W=fixed_W, simplex_rows=rows, xscale=xscale, max_iter=self.max_iter,
                        fix_zero_lines=False, filled_channels=filled_channels, filled_pixels=filled_pixels, layout=layout,
                        autotune=autotune, group=group, x_facts=x_facts, **self._engine_kwargs())

    def _engine_G(self):
        G = self.G_
        if self._identity_G:
            return None
        return G

    def loss(self, W, H, average=True, X=None):
        """Loss of (W, H) (espm/estimators/base.py:167-207): generalised KL divergence of X from
        G W H (plus the regularisers of the subclass), evaluated on the device."""
        self.GWH_numel_ = self.G_.shape[0] * H.shape[1]
        if X is None:
            eng = self._get_engine()
        else:
            assert X.shape == (self.G_.shape[0], H.shape[1])
            eng = self._make_engine(np.asarray(X), 1.0, self._engine_G())
        eng.load_state(W, self._local_H(eng, H))   # (the sharded engine of a fit: this rank's block of H; the loss is summed over the ranks)
        eng.eval_current(advance_h=False)
        h = eng.history(upto=0, average=average)
        kl, loss = float(h["kl"][0]), float(h["loss"][0])
        if X is not None and not self.l2 and self.const_KL_ is not None:
            # base.py:196-203: the constant of the KL term is the one cached for the FITTED data, also when another X is
            # handed in (the reference never recomputes it): replace the engine's own constant of `X`
            Xa = np.asarray(X, dtype=np.float64)
            own = float(np.sum(Xa * np.log(np.maximum(Xa, self.log_shift))) - np.sum(Xa))
            shift = (self.const_KL_ - own) / (float(self.GWH_numel_) if average else 1.0)
            kl, loss = kl + shift, loss + shift
        self.detailed_loss_ = self._detailed(kl, float(h["reg"][0]), float(h["lap"][0]))
        return loss

    def _get_engine(self):
        eng = getattr(self, "_engine", None)
        if eng is None:
            check_is_fitted(self, "X_")
            scale = getattr(self, "norm_factor_", 1.0) if self.normalize else 1.0
            eng = self._engine = self._make_engine(self._X_fixed(), scale, self._engine_G(), shard=getattr(self, "_shard", None))
        return eng

    def _X_fixed(self):
        X = self.X_
        if self.normalize:
            X = X / self.norm_factor_
        return X

    # ---- fit --------------------------------------------------------------------------------------------
    def fit_transform(self, X, y=None, W=None, H=None):
        """Learn X ~ G W H and return G W (or H.T with ``hspy_comp``), espm/estimators/base.py:209-420.

        Six stages, each a method that reads and fills the fit's context ``f`` (a SimpleNamespace; the stages' docstrings name the reference lines
        they stand for): validation, ingest (for a large X: ONE upload whose scans replace the reference's host passes), the initial W / H / G, the
        engine, the iteration loop with the stop rules, the results."""
        import types
        f = types.SimpleNamespace()
        self._fit_validate(f, X)
        self._fit_ingest(f)
        self._fit_initial_state(f, W, H)
        eng = self._fit_engine(f)
        self._fit_loop(f, eng)
        return self._fit_results(f, eng)

    def _fit_validate(self, f, X):
        """base.py:243-259: scikit-learn's validation (the finiteness scan of a large array moves to its device copy), the hyperspy caller check, the l2 guard."""
        # base.py:243-247.  For a large array that will be uploaded anyway the finiteness scan of validate_data (a full
        # host pass: 0.18 s at 2048 x 512^2 fp32) moves to the device copy below; everything else
        # (dtype, shape, n_features_in_, feature names) is still scikit-learn's.
        # ESPM_FIT_TIMING=1: host-side time stamps of the fit's sections (no device synchronisation), printed at its end
        marks = [("enter", time.perf_counter())] if os.environ.get("ESPM_FIT_TIMING") else None
        mark = (lambda name: marks.append((name, time.perf_counter()))) if marks is not None else (lambda name: None)
        big = False
        try:
            import torch
            big = (hasattr(X, "shape") and getattr(X, "ndim", 0) == 2 and int(np.prod(X.shape)) >= _DEVICE_PREP_MIN_SIZE
                   and getattr(X, "dtype", None) in (np.float32, np.float64) and torch.cuda.is_available())
        except Exception:
            big = False
        vkw = dict(dtype=[np.float64, np.float32])
        if big:
            vkw["ensure_all_finite"] = False
        if self.hspy_comp:
            Xv = validate_data(self, X.T, **vkw)
        else:
            Xv = validate_data(self, X, **vkw)
        mark("validate_data")
        if self.hspy_comp is False:
            try:  # base.py:249-259
                # (the reference asks inspect.getouterframes(inspect.currentframe(), 2) for calframe[1][3] and calframe[1][1]: the
                #  caller's function name and file.  The frame object has both; inspect also loads two lines of source context for
                #  EVERY frame of the stack - linecache stat calls that took 76 ms in one fit of five - and its list of frames
                #  holds this frame, which holds the list: a cycle that kept every local of the fit alive until the collector ran.)
                import sys
                caller = sys._getframe(2)   # (this method <- fit_transform <- the caller the reference's check looks at)
                calframe = [None, (caller, caller.f_code.co_filename, caller.f_lineno, caller.f_code.co_name)]
                caller = None
                if calframe[1][3] == "decomposition" and "hyperspy" in calframe[1][1]:
                    print("Are you calling the function decomposition from Hyperspy?\n"
                          "If so, please set the compatibility argument 'hspy_comp' to True.\n\n"
                          "If this argument is not set correctly, the function will not work properly!!!")
            except Exception:
                pass
            finally:
                calframe = None      # (no frame reference outlives the check: tools/analysis/est_cycle_probe.py)
        mark("caller check (inspect)")
        if self.l2 and getattr(self, "algo", None) != "l2_surrogate":
            raise NotImplementedError("the Frobenius loss (l2=True) is built for SmoothNMF(algo='l2_surrogate'), the one "
                                      "combination in which the reference keeps it (smooth_nmf.py:223-237)")
        f.marks, f.mark, f.big, f.Xv = marks, mark, big, Xv

    def _fit_ingest(self, f):
        """base.py:261-268, :519-528, :200-201: sign check, zero lines, mean / normalisation, X_, const_KL_ - for a large X on ONE device copy (a sharded fit: on this rank's block, scans combined over the ranks)."""
        Xv, big, mark = f.Xv, f.big, f.mark

        # Large X: ONE upload; the passes the reference makes over X on the host before the loop (sign check, zero
        # lines base.py:519-528, mean for normalize, const_KL_ base.py:200-201, the NNDSVD's products) run on that
        # device copy, which then feeds the engine.  Small X: the host path, like the reference.
        # A (pixels, channels) array - what hyperspy's decomposition hands over, hspy_comp=True - is uploaded AS IT IS
        # (pixel-major is also the layout the engine ingests natively: no host transpose, no device transpose); Xd below
        # is the logical (n, p) view of the device copy either way.
        Xd = Xd_raw = None
        x_facts = None   # (large X on the device: what the upload's scans found out about it, for the engine)
        dev_layout = "cm"
        lazy = None    # the estimator's own host copy X_ of a large X, made on a worker thread (_HostCopy)
        # one fit over several GPUs (shard()): decided BEFORE the upload, so that a large X goes to the device as this rank's block
        # of image rows only - its scans are combined over the ranks, the initialisation's passes run on the blocks
        # (espm_amd/init_device.py), the engine takes the block as it is.  Set-up time and device memory per rank shrink with the
        # number of ranks like the iteration does (the reference: base.py:243-295, updates.py:160-223).
        grp = getattr(self, "_shard_group", None)
        shard = None
        if grp is not None:
            import torch
            if torch.distributed.get_world_size(grp) > 1:
                shard = _Shard(grp, self.shape_2d, int(Xv.shape[1]))
        x_local = False   # Xd_raw is this rank's block (sharded fit of a large X)
        if Xv.size >= _DEVICE_PREP_MIN_SIZE:
            import torch
            if torch.cuda.is_available():
                if Xv.flags.c_contiguous:
                    host = Xv
                elif Xv.T.flags.c_contiguous:
                    host, dev_layout = Xv.T, "pm"
                else:
                    host = np.ascontiguousarray(Xv)
                lazy = _HostCopy(Xv, dev_layout)
                try:
                    mark("host copy thread created")
                    if shard is not None:
                        x_local = True
                        # (pixel-major: the block is a run of rows of the array as it lies; channel-major: a strided copy of 1 / world of it)
                        host = host[shard.sl] if dev_layout == "pm" else np.ascontiguousarray(host[:, shard.sl])
                    upload_err = None
                    try:
                        Xd_raw, scans = _upload_with_scans(host, torch.device("cuda", torch.cuda.current_device()), self.log_shift)
                    except Exception as e:   # noqa: BLE001 - (out of memory, a bad block): decided jointly below
                        upload_err = e
                    if shard is not None:
                        # a rank that fails HERE must not leave its peers waiting in the scans' all-reduce: the ranks agree first
                        shard.agree(upload_err is None, "the upload of this rank's block of X", upload_err)
                    elif upload_err is not None:
                        raise upload_err
                    if shard is not None:
                        scans = shard.combine_scans(scans, dev_layout)
                    mark("upload returned")
                    Xd = Xd_raw if dev_layout == "cm" else Xd_raw.t()
                    n_bad, n_nan, n_neg = (int(v) for v in scans["bad"].cpu())
                    if big and n_bad:   # the scan validate_data was told to skip, same message
                        raise ValueError(f"Input X contains {'NaN' if n_nan else 'infinity'}.")
                    if n_neg:
                        raise ValueError("Negative values in data")
                except BaseException:
                    lazy.cancel()
                    raise
        mark("finite / sign checks read back")
        if big and Xd is None:   # (no device after all: scikit-learn's own check)
            from sklearn.utils import assert_all_finite
            assert_all_finite(Xv, input_name="X")
        self.const_KL_ = None
        xscale = 1.0
        # (channels / pixels without a single count in the image: the engine's sparse store leaves their fill out of its lists)
        if Xd is None:
            X_fixed = self.remove_zeros_lines(Xv, self.log_shift)
            mean_x = None
            empty_ch, empty_px = Xv.sum(axis=1) == 0, Xv.sum(axis=0) == 0
        else:
            try:
                # (the line sums of the array as it lies in memory, from the upload's scans: channels are its rows in the "cm" layout)
                # (sharded: the channel sums are the image's, the pixel sums this rank's block's)
                zc, zp = ((scans["row_sum"] == 0, scans["col_sum"] == 0) if dev_layout == "cm"
                          else (scans["col_sum"] == 0, scans["row_sum"] == 0))
                empty_ch, empty_px = zc, zp
                n_zero_px = zp.sum().to(torch.float64)
                if x_local:
                    torch.distributed.all_reduce(n_zero_px, group=shard.group)
                n_zero_lines, s1, n_nonint, nnz, x_max = (float(v) for v in torch.cat((torch.stack((n_zero_px + zc.sum(), scans["s1"])),
                                                                                             scans["facts"])).cpu())
                numel = float(Xv.size)
                fill = n_zero_lines > 0
                if not fill and not x_local:   # what the scans know about X as it goes to the engine (a filled X is another array: the engine looks itself)
                    x_facts = dict(nonneg=True, sum_x=s1, is_count=bool(n_nonint == 0 and x_max <= 255), nnz=int(nnz))
                if fill:
                    Xd[:, zp] = self.log_shift
                    Xd[zc, :] = self.log_shift
                    total = Xd.sum(dtype=torch.float64)
                    if x_local:
                        torch.distributed.all_reduce(total, group=shard.group)
                    mean_x = float(total) / numel
                else:
                    mean_x = s1 / numel
            except BaseException:   # (the worker must not wait for a finish() that will not come)
                lazy.cancel()
                raise
            # X_ is the estimator's own array, like the reference's (remove_zeros_lines copies): the worker thread that is
            # copying it now fills the empty lines and applies the normalisation
            X_fixed = lazy
        mark("empty lines, mean read back")
        if self.normalize:
            self.norm_factor_ = (normalization_factor(X_fixed, self.n_components) if mean_x is None
                                 else self.n_components / (mean_x * X_fixed.shape[0]))
            xscale = float(self.norm_factor_)
        if lazy is not None:
            zp_all = zp
            if fill and x_local:   # (the host copy is the whole image: it fills the empty pixels of every rank's block)
                zp_all = torch.from_numpy(shard.gather_cols(zp.to(torch.uint8)[None, :])[0].astype(bool))
            lazy.finish(zp_all.cpu().numpy() if fill else None, zc.cpu().numpy() if fill else None, self.log_shift,
                        self.norm_factor_ if self.normalize else None)
            self.X_ = lazy
            # the copy runs from here on - behind the upload, next to the initialisation, the engine set-up and the loop (~120 ms
            # for its ~45): joined at the end of the fit it has long finished.  (Rounds 2-3 started it at the loop, because
            # started earlier it "stalled" whatever phase it overlapped: that was the container's CPU quota, _cpu_budget.py.
            # ESPM_HOSTCOPY_START=loop restores that; profiles/r03ac_fit_timing_*.log.)
            if os.environ.get("ESPM_HOSTCOPY_START", "early") == "early":
                lazy.start()
        else:
            self.X_ = self.norm_factor_ * X_fixed if self.normalize else X_fixed
        X_init_dev = None
        if Xd is not None:
            X_init_dev = Xd * xscale if self.normalize else Xd
        if Xd is not None and not fill and not self.normalize:
            # const_KL_ = sum(X log X - X) (base.py:200-201): both sums came with the upload
            self._const_KL_dev = float(scans["s2"]) - s1
        elif Xd is not None:
            # (lines filled or the image rescaled: the sums belong to another array)  In fp64, in row chunks of 64 M entries: in one
            # piece its fp64 copy and the three temporaries of the expression were 17 GB next to a 2 GB image
            rows_of = X_init_dev if X_init_dev.is_contiguous() else X_init_dev.t()   # (the orientation the memory lies in)
            step = max(1, (64 << 20) // max(1, int(rows_of.shape[1])))
            total = torch.zeros((), dtype=torch.float64, device=rows_of.device)
            for a in range(0, int(rows_of.shape[0]), step):
                xs = rows_of[a:a + step].to(torch.float64)
                total += (xs * torch.log(xs.clamp_min(self.log_shift))).sum() - xs.sum()
            if x_local:
                torch.distributed.all_reduce(total, group=shard.group)
            self._const_KL_dev = float(total)
            del xs, rows_of

        mark("const_KL read back")
        f.Xd, f.Xd_raw, f.x_facts, f.dev_layout, f.lazy, f.shard, f.x_local = Xd, Xd_raw, x_facts, dev_layout, lazy, shard, x_local
        f.xscale, f.X_fixed, f.mean_x, f.empty_ch, f.empty_px, f.X_init_dev = xscale, X_fixed, mean_x, empty_ch, empty_px, X_init_dev
        f.fill = bool(fill) if Xd is not None else False

    def _fit_initial_state(self, f, W, H):
        """base.py:269-295 with updates.py:160-223: the physics model's G, W0 / H0 (NNDSVD on the device for a large X), rank 0's arrays on every rank of a sharded fit."""
        mark, shard, x_local, Xd, xscale, mean_x, X_init_dev = f.mark, f.shard, f.x_local, f.Xd, f.xscale, f.mean_x, f.X_init_dev
        f.X_init_dev = None
        if _is_physical_model(self.G):
            self.physics_model_ = self.G
            G = self.physics_model_.NMF_update()
        else:
            self.physics_model_ = None
            G = self.G
        self._identity_G = G is None
        # (Round 5, measured and withdrawn: the engine's build - it needs the uploaded X and G, not W or H - on a worker thread and a stream of its
        #  own BESIDE the initialisation: the two take 28.5 + 12.6 ms one after the other and 39.6 + 0.1 side by side - both are chains of
        #  launches that fill the device, not waits; profiles/r05g_fit_timing*.log)
        self.G_, self.W_, self.H_ = initialize_algorithms(X=self.X_, G=G, W=W, H=H, n_components=self.n_components,
                                                          init=self.init, random_state=self.random_state,
                                                          simplex_H=self.simplex_H, simplex_W=self.simplex_W,
                                                          physics_model=self.physics_model_, X_device=X_init_dev,
                                                          shard=shard if x_local else None,   # (X_device is this rank's block of pixels)
                                                          # (the mean of what the initialisation sees: known from the upload's scans)
                                                          X_mean=(mean_x * xscale if (Xd is not None and mean_x is not None) else None))
        del X_init_dev
        mark("initialize_algorithms")
        # one fit over several GPUs (shard()): this rank's block of image rows; every rank starts from rank 0's W, H, G
        if shard is not None:
            import torch
            dev = f"cuda:{torch.cuda.current_device()}"
            self.G_, self.W_, self.H_ = shard.broadcast([self.G_, self.W_, self.H_], dev)
        self._shard = shard
        say = print if shard is None or shard.rank == 0 else (lambda *a, **k: None)   # (the reference's messages: once, not per rank)
        # L_ (base.py:286-291) is only an attribute here - the kernels apply the Laplacian as a stencil - and building the
        # sparse matrix of a 512 x 512 grid costs 0.07 s: it is built on first access (property L_ below)
        self._L_cache = None
        self._L_pixels = int(self.X_.shape[1])

        out_dtype = self.X_.dtype
        f.say, f.out_dtype = say, out_dtype

    def _fit_engine(self, f):
        """The device engine of the fit with its state loaded (a sharded fit: the record exchange rehearsed), const_KL_."""
        mark, shard, x_local, Xd, Xd_raw, X_fixed, xscale, fill, Xv = f.mark, f.shard, f.x_local, f.Xd, f.Xd_raw, f.X_fixed, f.xscale, f.fill, f.Xv
        empty_ch, empty_px, dev_layout, x_facts = f.empty_ch, f.empty_px, f.dev_layout, f.x_facts
        f.Xd = f.Xd_raw = f.X_fixed = None   # (the context must not keep the device copies of X alive past the engine's build)
        no_fill = Xd is not None and not fill      # (known from the upload's scans: no read-back to ask again)
        self._engine = eng = self._make_engine(X_fixed if Xd is None else Xd_raw, xscale, None if self._identity_G else self.G_,
                                                filled_channels=None if no_fill or not bool(empty_ch.any()) else empty_ch,
                                                filled_pixels=None if no_fill or not bool(empty_px.any()) else empty_px, layout=dev_layout,
                                                x_facts=x_facts,
                                                # (timing the launch plans costs ~30 ms of device time and gains a few per cent
                                                #  per iteration: it pays for itself only in very long fits of large images)
                                                autotune="auto" if Xd is not None else False,   # (MUEngine: from AUTOTUNE_MIN_ITERS iterations on)
                                                shard=shard, x_local=x_local)
        self._ingest_layout = dev_layout   # "pm": the (pixels, channels) input went to the device without a transpose
        mark("engine built")
        del X_fixed, Xd, Xd_raw
        mark("device copies of X released")
        eng.load_state(self.W_, self.H_ if shard is None else shard.cols(self.H_))
        if shard is not None and getattr(eng, "sharded", False) and eng.exchange.ctx is not None:
            # the one-shot record exchange is rehearsed with the fit's own kernels before the loop depends on it; every rank
            # moves to the collective transport together if a wait gave up (MUEngine.settle_exchange), then the state again
            eng.settle_exchange()
            eng.load_state(self.W_, shard.cols(self.H_))
        self.GWH_numel_ = self.G_.shape[0] * self.H_.shape[1]
        self.const_KL_ = (getattr(self, "_const_KL_dev", None) if Xv.size >= _DEVICE_PREP_MIN_SIZE else None)
        if self.const_KL_ is None:
            self.const_KL_ = float(np.sum(self.X_ * np.log(np.maximum(self.X_, self.log_shift))) - np.sum(self.X_))
        self._const_KL_dev = None

        mark("state loaded, const_KL")
        return eng

    def _fit_loop(self, f, eng):
        """base.py:313-394: the iterations, the bookkeeping of every one of them and the stop rules; a sharded fit that loses a peer restarts once on the collective transport."""
        mark, shard, lazy, say, out_dtype = f.mark, f.shard, f.lazy, f.say, f.out_dtype
        algo_start = time.time()
        self.n_iter_ = 0
        self._begin_fit()
        self.losses_, self.rel_, self.detailed_losses_ = [], [], []
        # ground-truth tracking (base.py:301-311): every iteration compares G W with true_D, H with true_H and
        # evaluates the loss against the noiseless true_D @ true_H (a second, loss-only engine holds that X)
        track = self._begin_truth_tracking()
        pg_ls = bool(getattr(self, "linesearch", False)) and getattr(self, "algo", None) == "projected_gradient"
        adapt = bool(getattr(self, "linesearch", False)) and not pg_ls
        sync_each = (not self.no_stop_criterion) or self.physics_model_ is not None or track or adapt or pg_ls or bool(self.l2)
        eval_before = np.inf
        eval_init = None
        stop = False
        if lazy is not None:
            lazy.start()
        G_start = None if self.physics_model_ is None else np.array(self.G_, copy=True)
        retried = False
        try:
            while True:   # (a second pass only after a sharded fit has had to change its record exchange, below)
                try:
                    eng.eval_current(advance_h=True)  # loss of the initial state rides on the first H-step
                    while not stop:
                        # how many iterations may run before the host has to look at a loss value
                        if sync_each:
                            chunk = 1
                        else:
                            chunk = self.max_iter - self.n_iter_
                            if self.verbose > 0:
                                chunk = min(chunk, self.eval_print - self.n_iter_ % self.eval_print)
                            chunk = max(chunk, 1)
                        if pg_ls:  # smooth_nmf.py:382-401: gamma_H from the quadratic bound, before the W-step
                            self.gamma_[0] = eng.pg_linesearch_h(self.gamma_[0])
                        eng.finish_iteration()
                        if adapt:  # smooth_nmf.py:376-381: gamma_ follows the Laplacian surrogate; in effect from the next H-step
                            self.gamma_ = eng.linesearch_step(self._gamma_value())
                        if chunk > 1:
                            eng.iterate(chunk - 1, final_loss=False)
                            eng.eval_current(advance_h=(self.n_iter_ + chunk) < self.max_iter)
                        else:
                            last = self.n_iter_ + 1 >= self.max_iter
                            refresh_G = self.physics_model_ is not None and (self.n_iter_ + 1) % 3 == 0
                            eng.eval_current(advance_h=not (last or refresh_G))
                        if pg_ls:  # smooth_nmf.py:438-447: gamma_W, once the new state has been evaluated
                            self.gamma_[1] = eng.pg_linesearch_w(self.gamma_[1])
                        first = self.n_iter_ + 1
                        self.n_iter_ += chunk
                        h = eng.history(upto=self.n_iter_)
                        if eval_init is None:
                            eval_init = float(h["loss"][0])
                        for t in range(first, self.n_iter_ + 1):
                            self.losses_.append(float(h["loss"][t]))
                            self.detailed_losses_.append(self._detailed(float(h["kl"][t]), float(h["reg"][t]),
                                                                        float(h["lap"][t])))
                            self.rel_.append([float(h["rel_W"][t]), float(h["rel_H"][t])])
                        if track:
                            self._track_truth(eng)
                        eval_after = self.losses_[-1]
                        rel_W, rel_H = self.rel_[-1]

                        if self.n_iter_ >= self.max_iter:  # base.py:354-378
                            say("exits because max_iteration was reached")
                            break
                        if not self.no_stop_criterion:
                            if max(rel_H, rel_W) < self.tol:
                                say("exits because of relative change rel_A {} and rel_P {} < tol ".format(rel_H, rel_W))
                                break
                            elif abs((eval_before - eval_after) / eval_init) < self.tol:
                                say("exits because of relative change < tol: {}".format((eval_before - eval_after) / eval_init))
                                break
                            elif np.isnan(eval_after):
                                say("exit because of the presence of NaN")
                                break
                            elif (eval_before - eval_after) < 0:
                                say("exit because of negative decrease {}: {}, {}".format((eval_before - eval_after),
                                                                                             eval_before, eval_after))
                                break
                        if self.verbose > 0 and np.mod(self.n_iter_, self.eval_print) == 0:
                            say(f"It {self.n_iter_} / {self.max_iter}: loss {eval_after:3e},  "
                                  f"{self.n_iter_ / (time.time() - algo_start + log_shift):0.3f} it/s")
                        if self.physics_model_ is not None and self.n_iter_ % 3 == 0:  # base.py:388-392
                            self.G_ = self.physics_model_.NMF_update(eng.get_W().astype(out_dtype))
                            eng.set_G(self.G_)
                            eng.eval_current(advance_h=True)
                            eval_before = float(eng.history(upto=self.n_iter_)["loss"][self.n_iter_])
                        else:
                            eval_before = eval_after
                    break
                except Exception as e:   # noqa: BLE001
                    # A sharded fit on the one-shot record exchange whose bounded waits gave up AFTER the rehearsal before the loop
                    # (MUEngine.settle_exchange): history() raises on every rank at the same read-back (the health is a maximum over the ranks).
                    # Every rank then moves to the collective transport and the fit starts again from its initial W, H, G (the
                    # estimator needs the fallback bench.py has).  Anything else, or a second failure, is the caller's.
                    from espm_amd._lib import LostPeerError
                    if not isinstance(e, LostPeerError) or retried or shard is None or getattr(eng.exchange, "ctx", None) is None:
                        raise
                    retried = True
                    say("record exchange: a peer was lost on the one-shot transport; restarting the fit on the collective transport")
                    eng.use_collective_exchange()
                    if G_start is not None:
                        # the physics model's own state followed the abandoned iterates (NMF_update(W) every third iteration, base.py:388-392):
                        # it is brought back to the fit's start the way the fit got there - NMF_update() without W (base.py:269-274)
                        if self.physics_model_ is not None:
                            self.physics_model_.NMF_update()
                        self.G_ = G_start
                        eng.set_G(self.G_)
                    eng.load_state(self.W_, shard.cols(self.H_))
                    self.n_iter_, stop, eval_before, eval_init = 0, False, np.inf, None
                    self.losses_, self.rel_, self.detailed_losses_ = [], [], []
                    self._begin_fit()
                    track = self._begin_truth_tracking()
        except KeyboardInterrupt:
            pass
        finally:
            mark("loop (losses read back)")
            if lazy is not None:   # the host copy that was made meanwhile - also when the loop raised: X_ must not stay a stand-in
                self.X_ = lazy.result()   # (holding a thread and an event: not picklable)
        mark("host copy X_ joined")
        f.algo_start = algo_start

    def _fit_results(self, f, eng):
        """base.py:396-420: W_, H_ read back, rescaled_DH without a simplex, reconstruction_err_, the un-normalised W_, the return value by hspy_comp."""
        mark, marks, say, out_dtype, algo_start = f.mark, f.marks, f.say, f.out_dtype, f.algo_start
        self.W_ = eng.get_W().astype(out_dtype)
        self.H_ = self._full_H(eng).astype(out_dtype)
        mark("W, H read back")
        if marks is not None:
            print("[fit timing, ms] " + ", ".join(f"{n} {1e3 * (t - marks[i][1]):.1f}" for i, (n, t) in enumerate(marks[1:])) +
                  f" | total {1e3 * (marks[-1][1] - marks[0][1]):.1f}", flush=True)
        if not self.simplex_H and not self.simplex_W:
            self.W_, self.H_ = rescaled_DH(self.W_, self.H_)  # base.py:399-400

        algo_time = time.time() - algo_start
        say(f"Stopped after {self.n_iter_} iterations in {algo_time // 60} minutes "
              f"and {np.round(algo_time) % 60} seconds.")
        if not self.simplex_H and not self.simplex_W:
            self.reconstruction_err_ = self.loss(self.W_, self.H_)
        else:
            self.reconstruction_err_ = self.losses_[-1] if self.losses_ else self.loss(self.W_, self.H_)
            if self.losses_:
                self.detailed_loss_ = self.detailed_losses_[-1]
        if self.normalize:
            self.W_ = self.W_ / self.norm_factor_

        GW = self.G_ @ self.W_
        self.n_components_ = self.H_.shape[0]
        if self.hspy_comp:
            self.components_ = GW.T
            return self.H_.T
        self.components_ = self.H_
        return GW

    def _begin_fit(self):
        pass

    def _local_H(self, eng, H):
        """The block of H an engine holds: all of it, or this rank's image rows for the sharded engine of a fit."""
        return self._shard.cols(np.asarray(H)) if getattr(eng, "world", 1) > 1 else H

    def _full_H(self, eng):
        """H of the whole image: the engine's, or the ranks' blocks of a sharded fit assembled on every rank."""
        shard = getattr(self, "_shard", None)
        if shard is None or getattr(eng, "world", 1) == 1:
            return eng.get_H()
        return shard.gather_cols(eng.h[eng.st.cur][:, :eng.p])

    def fit(self, X, y=None, **params):
        """Learn a NMF model for the data X (espm/estimators/base.py:422-441).

        Deviation from the reference: ``n_components`` is limited to 32 (``NotImplementedError`` above that, raised before anything is
        uploaded).  The reference has no limit (base.py:126-132 stores the argument, updates.py:160-223 initialise any rank); here the
        kernels are compiled per component count - 1..8 in libespm_mu.so, 9..16 in libespm_mu_wide.so, 17..32 in libespm_mu_wide32.so -
        with the components of a pixel or channel in registers.  The sparse count store serves up to 16 components: sparse count data with
        more, or with 13-16 components and more than 2048-2144 channels (9-12: 2896-3024), take the dense 8-bit store (below 17 components
        with a RuntimeWarning: the sparse store's G W table would not fit a workgroup's LDS)."""
        self.fit_transform(X, **params)
        return self

    def inverse_transform(self, W):
        """G W H_ (espm/estimators/base.py:461-477)."""
        check_is_fitted(self)
        return self.G_ @ W @ self.H_

    @property
    def L_(self):
        """Laplacian of the pixel grid (espm/utils.py:39-76) or the identity without ``shape_2d`` (base.py:286-291), built on
        first access."""
        if getattr(self, "_L_cache", None) is None:
            if not hasattr(self, "_L_pixels"):
                raise AttributeError("L_")
            self._L_cache = (create_laplacian_matrix(*self.shape_2d) if self.shape_2d is not None
                             else identity_laplacian(self._L_pixels))
        return self._L_cache

    @L_.setter
    def L_(self, value):
        self._L_cache = value

    def get_losses(self):
        """Structured array of the loss history (espm/estimators/base.py:479-517); with true_D / true_H given also the
        spectral angles, the map errors and the loss against the noiseless truth of every iteration."""
        names = ["full_loss"] + self.loss_names_ + ["rel_W", "rel_H"]
        tracked = self.true_D is not None and self.true_H is not None
        if tracked:  # base.py:483-499 (the reference fails here when the truth was ignored for its shape; so does this)
            names += [f"ang_p{i}" for i in range(self.n_components)] + [f"mse_p{i}" for i in range(self.n_components)]
            names += ["true_KL_loss"]
        dt = np.dtype([(elt, "float64") for elt in names])
        rows = []
        for i in range(len(self.losses_)):
            row = (self.losses_[i],) + tuple(self.detailed_losses_[i]) + tuple(self.rel_[i])
            if tracked:
                row += tuple(self.angles_[i]) + tuple(self.mse_[i]) + (self.true_losses_[i],)
            rows.append(row)
        return np.array(rows, dtype=dt)

    # ---- ground-truth tracking (espm/estimators/base.py:301-311, :335-347) ------------------------------------------
    def _begin_truth_tracking(self):
        self._truth_engine = None
        if self.true_D is None or self.true_H is None:
            return False
        true_D, true_H = np.asarray(self.true_D), np.asarray(self.true_H)
        if true_D.shape[1] != self.n_components or true_H.shape[0] != self.n_components:
            print("The chosen number of components does not match the number of components of the provided truth. "
                  "The ground truth will be ignored.")
            return False
        self.angles_, self.mse_, self.true_losses_ = [], [], []
        true_DH = (true_D @ true_H).astype(np.float64)
        # loss(W, H, X = true_DH) adds the constant cached for the DATA, not the one of true_DH (base.py:196-203)
        self._truth_const = float(np.sum(true_DH * np.log(np.maximum(true_DH, self.log_shift))) - np.sum(true_DH))
        self._truth_engine = self._make_engine(true_DH, 1.0, self._engine_G())
        return True

    def _track_truth(self, eng):
        from espm_amd.measures import find_min_angle, find_min_MSE
        W, H = eng.get_W().astype(np.float64), self._full_H(eng).astype(np.float64)   # (sharded fit: every rank tracks the whole image)
        Wc, Hc = (W, H) if (self.simplex_H or self.simplex_W) else rescaled_DH(W, H)
        GW = self.G_ @ Wc
        self.angles_.append(find_min_angle(np.asarray(self.true_D).T, GW.T, unique=True))
        self.mse_.append(find_min_MSE(np.asarray(self.true_H), Hc, unique=True))
        te = self._truth_engine
        te.load_state(W, Hc)                      # base.py:344: loss(self.W_, H, X = true_DH) - W as fitted, H rescaled
        te.eval_current(advance_h=False)
        h = te.history(upto=0, average=False)
        numel = float(self.G_.shape[0] * Hc.shape[1])
        lkl = float(h["kl"][0]) if self.l2 else float(h["kl"][0]) - self._truth_const + self.const_KL_
        self.true_losses_.append((lkl + float(h["reg"][0]) + float(h["lap"][0])) / numel)

    def remove_zeros_lines(self, X, epsilon):
        """All-zero rows / columns of X become epsilon (espm/estimators/base.py:519-528)."""
        if np.all(X >= 0):
            new_X = X.copy()
            new_X[:, X.sum(axis=0) == 0] = epsilon
            new_X[X.sum(axis=1) == 0, :] = epsilon
            return new_X
        raise ValueError("Negative values in data")

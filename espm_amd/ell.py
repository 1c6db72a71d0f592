"""Builds the sparse count store of include/espm_mu.h (x_dtype = ESPM_X_ELL) from a dense count image.

Plumbing only (torch tensor ops on the device that holds X): the lists are built once per fit, outside
the iteration loop; the kernels that consume them are in espm_amd/csrc/mu_ell_kernel.hpp.

Layout (see the header for the authoritative description): 16-bit entries, two per dword, the dwords of the
64 lists of a wave interleaved ("ELL" rows of 64 dwords); value 0 pads a list to the longest of its 64.
  H-step lists: one per pixel, entry = count << cbits | channel, ascending channel; inside every window of
                ``tile_px`` pixels the lists are ordered by decreasing length (``pix_perm``: slot -> pixel offset).
  W-step lists: one per (block of 1024 pixels, channel), entry = count << 10 | pixel - block start, ascending
                pixel; the channels of a wave are 64 consecutive entries of ``chan_perm[block]`` (the block's
                channels by decreasing list length).
A count larger than its field is split over several entries.
"""
from __future__ import annotations

import torch

from . import _lib


def _split_counts(x, xmax):
    """entries needed per element: ceil(x / xmax) (0 for x = 0)."""
    return (x + (xmax - 1)) // xmax


def _expand(idx_major, idx_minor, x, reps, xmax):
    """Expands elements (sorted by major, minor) into their entries.

    Returns (major, minor, value) per entry: an element with count x and r = reps entries yields r - 1 entries of
    xmax and one of x - (r - 1) xmax."""
    if int(reps.max()) <= 1:
        return idx_major, idx_minor, x
    total = torch.repeat_interleave(reps)
    first = torch.cumsum(reps, 0) - reps
    pos = torch.arange(total.numel(), device=x.device) - first[total]
    r_e, x_e = reps[total], x[total]
    val = torch.where(pos < r_e - 1, torch.full_like(x_e, xmax), x_e - (r_e - 1) * xmax)
    return idx_major[total], idx_minor[total], val


def _store16(ell16, dword, half, value):
    """ell16 is the int16 view of the dword array: entry `half` (0 = low) of dword index `dword`."""
    v = value.to(torch.int32)
    v = torch.where(v >= 32768, v - 65536, v).to(torch.int16)
    ell16[dword * 2 + half] = v


def count_entries(Xpm, xmax_h, xmax_w, chunk=16384):
    """(entries per pixel for the H lists, entries per channel for the W lists, non-zeros) of the (p, n) image."""
    p, n = Xpm.shape
    per_px = torch.empty(p, dtype=torch.int64, device=Xpm.device)
    per_ch = torch.zeros(n, dtype=torch.int64, device=Xpm.device)
    nnz = 0
    for q0 in range(0, p, chunk):
        xi = Xpm[q0:q0 + chunk].to(torch.int32)
        per_px[q0:q0 + chunk] = _split_counts(xi, xmax_h).sum(dim=1)
        per_ch += _split_counts(xi, xmax_w).sum(dim=0)
        nnz += int((xi != 0).sum())
    return per_px, per_ch, nnz


def build(Xpm, p_pad, cbits, tile_px=512, chunk=16384):
    """Xpm: (p, n) non-negative integer-valued tensor on the device (any float dtype).

    Returns a dict of device tensors: ell_h (int32 dwords), ell_h_off (int32), klc (float32, p_pad), pix_perm (int32,
    p_pad), ell_w, ell_w_off, chan_perm (int32, nblk_w x 64 n_cg), and the python ints n_cg, nblk_w, nnz, entries_h,
    entries_w, rows_h, rows_w."""
    dev = Xpm.device
    p, n = Xpm.shape
    PB, PBITS = _lib.ELL_PB, _lib.ELL_PBITS
    xmax_h = (1 << (16 - cbits)) - 1
    xmax_w = (1 << (16 - PBITS)) - 1
    n_cg = (n + 63) // 64
    nblk_w = (p + PB - 1) // PB
    i32 = dict(dtype=torch.int32, device=dev)

    per_px, per_ch, nnz = count_entries(Xpm, xmax_h, xmax_w, chunk)

    # ---- H lists ---------------------------------------------------------------------------------------
    ngrp = p_pad // 64
    cnt_pad = torch.zeros(p_pad, dtype=torch.int64, device=dev)
    cnt_pad[:p] = per_px
    # inside every window of tile_px pixels: slots by decreasing list length (stable)
    pix_perm = torch.argsort(cnt_pad.view(-1, tile_px), dim=1, descending=True, stable=True)     # slot -> pixel offset
    win0 = (torch.arange(p_pad, device=dev) // tile_px) * tile_px
    slot_pixel = win0 + pix_perm.reshape(-1)                     # global slot -> global pixel
    slot_of = torch.empty(p_pad, dtype=torch.int64, device=dev)
    slot_of[slot_pixel] = torch.arange(p_pad, device=dev)       # global pixel -> global slot
    glen = cnt_pad[slot_pixel].view(ngrp, 64).max(dim=1).values  # entries of the longest list of each slot group
    grows = (glen + 1) // 2                                    # dword rows
    h_off = torch.zeros(ngrp + 1, dtype=torch.int64, device=dev)
    h_off[1:] = torch.cumsum(grows, 0)
    rows_h = int(h_off[-1])
    if rows_h * 64 >= 2 ** 31:
        raise ValueError("sparse count store: H lists exceed 2^31 dwords")
    ell_h = torch.zeros(max(rows_h, 1) * 64, **i32)
    ell_h16 = ell_h.view(torch.int16)
    klc = torch.zeros(p_pad, dtype=torch.float32, device=dev)
    xm = float(xmax_h)
    for q0 in range(0, p, chunk):
        blk = Xpm[q0:q0 + chunk]
        xi = blk.to(torch.int64)
        xd = blk.to(torch.float64)
        # loss correction of the split counts: x log2 x - sum over the entries x_i of x_i log2 x_i
        nfull = torch.ceil(xd / xm).clamp_min(1.0) - 1.0
        rest = xd - nfull * xm
        klc[q0:q0 + blk.shape[0]] = (xd * torch.log2(xd.clamp_min(1.0)) - nfull * xm * float(torch.log2(torch.tensor(xm)))
                                     - rest * torch.log2(rest.clamp_min(1.0))).sum(dim=1).to(torch.float32)
        nz = xi.nonzero(as_tuple=False)                       # sorted by pixel, then channel
        if nz.numel() == 0:
            continue
        q, c = nz[:, 0], nz[:, 1]
        x = xi[q, c]
        q, c, val = _expand(q, c, x, _split_counts(x, xmax_h), xmax_h)
        cnt = per_px[q0:q0 + blk.shape[0]]
        start = torch.cumsum(cnt, 0) - cnt                    # first entry of each pixel of the chunk
        j = torch.arange(q.numel(), device=dev) - start[q]    # position in the pixel's list
        slot = slot_of[q + q0]
        dword = (h_off[slot >> 6] + (j >> 1)) * 64 + (slot & 63)
        _store16(ell_h16, dword, j & 1, (val << cbits) | c)

    # ---- W lists ---------------------------------------------------------------------------------------
    # pass 1: entries per (block, channel); inside every block the channels by decreasing list length (stable)
    cnt_bc = torch.zeros((nblk_w, n), dtype=torch.int64, device=dev)
    for b in range(nblk_w):
        xi = Xpm[b * PB:(b + 1) * PB].to(torch.int32)
        cnt_bc[b] = _split_counts(xi, xmax_w).sum(dim=0).to(torch.int64)
    order = torch.argsort(cnt_bc, dim=1, descending=True, stable=True)     # (nblk_w, n): slot -> channel
    chan_perm = torch.full((nblk_w, n_cg * 64), -1, dtype=torch.int64, device=dev)
    chan_perm[:, :n] = order
    cnt_slot = torch.zeros((nblk_w, n_cg * 64), dtype=torch.int64, device=dev)
    cnt_slot[:, :n] = torch.gather(cnt_bc, 1, order)
    wrows = (cnt_slot.view(nblk_w, n_cg, 64).max(dim=2).values + 1) // 2
    w_off = torch.zeros(nblk_w * n_cg + 1, dtype=torch.int64, device=dev)
    w_off[1:] = torch.cumsum(wrows.reshape(-1), 0)
    rows_w = int(w_off[-1])
    if rows_w * 64 >= 2 ** 31:
        raise ValueError("sparse count store: W lists exceed 2^31 dwords")
    ell_w = torch.zeros(max(rows_w, 1) * 64, **i32)
    ell_w16 = ell_w.view(torch.int16)
    for b in range(nblk_w):
        xt = Xpm[b * PB:(b + 1) * PB].to(torch.int64).t().contiguous()   # (n, pixels of the block)
        nz = xt.nonzero(as_tuple=False)                       # sorted by channel, then pixel
        if nz.numel() == 0:
            continue
        c, pl = nz[:, 0], nz[:, 1]
        x = xt[c, pl]
        c, pl, val = _expand(c, pl, x, _split_counts(x, xmax_w), xmax_w)
        cnt = cnt_bc[b]                                       # entries per channel (natural order)
        start = torch.cumsum(cnt, 0) - cnt
        j = torch.arange(c.numel(), device=dev) - start[c]
        slot_of_c = torch.empty(n, dtype=torch.int64, device=dev)
        slot_of_c[order[b]] = torch.arange(n, device=dev)
        slot = slot_of_c[c]
        dword = (w_off[b * n_cg + (slot >> 6)] + (j >> 1)) * 64 + (slot & 63)
        _store16(ell_w16, dword, j & 1, (val << PBITS) | pl)

    return dict(ell_h=ell_h, ell_h_off=h_off.to(torch.int32), klc=klc, pix_perm=pix_perm.reshape(-1).to(torch.int32), ell_w=ell_w, ell_w_off=w_off.to(torch.int32),
                chan_perm=chan_perm.to(torch.int32), n_cg=n_cg, nblk_w=nblk_w, nnz=nnz,
                entries_h=int(per_px.sum()), entries_w=int(per_ch.sum()), rows_h=rows_h, rows_w=rows_w)


def lds_bytes_h(n_pad, k):
    """LDS the sparse H-step needs: the GW table plus the numerators of a 512-pixel tile."""
    wb = 0 if k <= 4 else (1 if k == 5 else (2 if k == 6 else 4))
    return n_pad * (4 + wb) * 4 + max(k * _lib.ELL_TILE * 4, 9 * (5 + 2 * k) * 8)

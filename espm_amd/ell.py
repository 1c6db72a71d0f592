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
Every list group (64 lists) starts with UNIT rows: entries with count 1 stored as index << 4, in every lane and
position, as many as the group's poorest list has ones (rounded down to a multiple of 2 ELL_UNIT_ROWS entries); the
general rows (count << bits | index) follow.  The offsets carry two words per group.
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
    """(entries per pixel of the H lists, elements equal to 1 per pixel, entries per channel of the W lists,
    non-zeros) of the (p, n) image."""
    p, n = Xpm.shape
    per_px = torch.empty(p, dtype=torch.int64, device=Xpm.device)
    ones_px = torch.empty(p, dtype=torch.int64, device=Xpm.device)
    per_ch = torch.zeros(n, dtype=torch.int64, device=Xpm.device)
    nnz = 0
    for q0 in range(0, p, chunk):
        xi = Xpm[q0:q0 + chunk].to(torch.int32)
        per_px[q0:q0 + chunk] = _split_counts(xi, xmax_h).sum(dim=1)
        ones_px[q0:q0 + chunk] = (xi == 1).sum(dim=1)
        per_ch += _split_counts(xi, xmax_w).sum(dim=0)
        nnz += int((xi != 0).sum())
    return per_px, ones_px, per_ch, nnz


def _group_rows(longest, fewest_ones):
    """(unit rows, general rows) of list groups whose longest list has `longest` entries and whose poorest list
    has `fewest_ones` elements equal to 1 (espm_mu.h: UNIT rows)."""
    ub = _lib.ELL_UNIT_ROWS
    unit = fewest_ones // (2 * ub) * ub
    return unit, (longest - 2 * unit + 1) // 2


def _offsets(unit, general):
    """[first unit row, first general row] per group and the end of the last group."""
    rows = unit + general
    first = torch.cumsum(rows, 0) - rows
    off = torch.empty(2 * rows.numel() + 1, dtype=torch.int64, device=rows.device)
    off[0:-1:2] = first
    off[1::2] = first + unit
    off[-1] = rows.sum()
    return off


def _scatter_lists(ell16, major, minor, x, n_major, unit_cap, row_unit, row_general, lane, xmax, idx_bits):
    """Writes the entries of the elements (major, minor, x) - sorted by major, then minor; x > 0 - of n_major lists.

    List i may place its first unit_cap[i] elements equal to 1 into the unit rows that start at row_unit[i] (as
    minor << 4); everything else goes to the general rows from row_general[i] (as count << idx_bits | minor, counts
    above xmax split); lane[i] is the list's lane."""
    dev = x.device
    if x.numel() == 0:
        return
    is_one = (x == 1).to(torch.int64)
    nz_per = torch.bincount(major, minlength=n_major)
    first_el = torch.cumsum(nz_per, 0) - nz_per                   # first element of each list
    ones_before = torch.cumsum(is_one, 0) - is_one
    rank_one = ones_before - ones_before[first_el.clamp_max(x.numel() - 1)][major]   # rank among the list's ones
    in_unit = (is_one == 1) & (rank_one < unit_cap[major])
    mu, ju = major[in_unit], rank_one[in_unit]
    _store16(ell16, (row_unit[mu] + (ju >> 1)) * 64 + lane[mu], ju & 1, minor[in_unit] << 4)
    rest = ~in_unit
    mg, cg, xg = major[rest], minor[rest], x[rest]
    if mg.numel() == 0:
        return
    reps = _split_counts(xg, xmax)
    mg, cg, val = _expand(mg, cg, xg, reps, xmax)
    ent_per = torch.bincount(mg, minlength=n_major)
    first_ent = torch.cumsum(ent_per, 0) - ent_per
    j = torch.arange(mg.numel(), device=dev) - first_ent[mg]
    _store16(ell16, (row_general[mg] + (j >> 1)) * 64 + lane[mg], j & 1, (val << idx_bits) | cg)


def build(Xpm, p_pad, cbits, tile_px=512, chunk=16384):
    """Xpm: (p, n) non-negative integer-valued tensor on the device (any float dtype).

    Returns a dict of device tensors: ell_h (int32 dwords), ell_h_off (int32), klc (float32, p_pad), pix_perm (int32,
    p_pad), ell_w, ell_w_off, chan_perm (int32, nblk_w x 64 n_cg), and the python ints n_cg, nblk_w, nnz, entries_h,
    entries_w, rows_h, rows_w, unit_rows_h, unit_rows_w."""
    dev = Xpm.device
    p, n = Xpm.shape
    PB = 2 * tile_px                      # pixels per block of the W accumulation (espm_mu_state.ell_pb): two H tiles
    PBITS = PB.bit_length() - 1
    xmax_h = (1 << (16 - cbits)) - 1
    xmax_w = (1 << (16 - PBITS)) - 1
    n_cg = (n + 63) // 64
    nblk_w = (p + PB - 1) // PB
    i32 = dict(dtype=torch.int32, device=dev)

    per_px, ones_px, per_ch, nnz = count_entries(Xpm, xmax_h, xmax_w, chunk)
    if n > _lib.ELL_UNIT_MAX_N:       # index << 4 must fit 16 bits
        ones_px = torch.zeros_like(ones_px)

    # ---- H lists ---------------------------------------------------------------------------------------
    ngrp = p_pad // 64
    cnt_pad = torch.zeros(p_pad, dtype=torch.int64, device=dev)
    cnt_pad[:p] = per_px
    ones_pad = torch.zeros(p_pad, dtype=torch.int64, device=dev)
    ones_pad[:p] = ones_px
    # inside every window of tile_px pixels: slots by decreasing list length (stable)
    pix_perm = torch.argsort(cnt_pad.view(-1, tile_px), dim=1, descending=True, stable=True)     # slot -> pixel offset
    win0 = (torch.arange(p_pad, device=dev) // tile_px) * tile_px
    slot_pixel = win0 + pix_perm.reshape(-1)                     # global slot -> global pixel
    slot_of = torch.empty(p_pad, dtype=torch.int64, device=dev)
    slot_of[slot_pixel] = torch.arange(p_pad, device=dev)       # global pixel -> global slot
    glen = cnt_pad[slot_pixel].view(ngrp, 64).max(dim=1).values  # entries of the longest list of each slot group
    gones = ones_pad[slot_pixel].view(ngrp, 64).min(dim=1).values
    gunit, ggen = _group_rows(glen, gones)
    h_off = _offsets(gunit, ggen)
    rows_h = int(h_off[-1])
    if rows_h * 64 >= 2 ** 31:
        raise ValueError("sparse count store: H lists exceed 2^31 dwords")
    ell_h = torch.zeros(max(rows_h, 1) * 64, **i32)
    ell_h16 = ell_h.view(torch.int16)
    klc = torch.zeros(p_pad, dtype=torch.float32, device=dev)
    xm = float(xmax_h)
    grp_of_px = slot_of >> 6                                   # pixel -> slot group
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
        g = grp_of_px[q0:q0 + blk.shape[0]]
        _scatter_lists(ell_h16, q, c, xi[q, c], blk.shape[0], 2 * gunit[g], h_off[2 * g], h_off[2 * g + 1],
                       slot_of[q0:q0 + blk.shape[0]] & 63, xmax_h, cbits)

    # ---- W lists ---------------------------------------------------------------------------------------
    # pass 1: entries and ones per (block, channel); inside every block the channels by decreasing list length (stable)
    cnt_bc = torch.zeros((nblk_w, n), dtype=torch.int64, device=dev)
    ones_bc = torch.zeros((nblk_w, n), dtype=torch.int64, device=dev)
    for b in range(nblk_w):
        xi = Xpm[b * PB:(b + 1) * PB].to(torch.int32)
        cnt_bc[b] = _split_counts(xi, xmax_w).sum(dim=0).to(torch.int64)
        ones_bc[b] = (xi == 1).sum(dim=0)
    order = torch.argsort(cnt_bc, dim=1, descending=True, stable=True)     # (nblk_w, n): slot -> channel
    chan_perm = torch.full((nblk_w, n_cg * 64), -1, dtype=torch.int64, device=dev)
    chan_perm[:, :n] = order
    cnt_slot = torch.zeros((nblk_w, n_cg * 64), dtype=torch.int64, device=dev)
    cnt_slot[:, :n] = torch.gather(cnt_bc, 1, order)
    ones_slot = torch.zeros((nblk_w, n_cg * 64), dtype=torch.int64, device=dev)   # a slot without a channel has no ones
    ones_slot[:, :n] = torch.gather(ones_bc, 1, order)
    wunit, wgen = _group_rows(cnt_slot.view(nblk_w, n_cg, 64).max(dim=2).values, ones_slot.view(nblk_w, n_cg, 64).min(dim=2).values)
    w_off = _offsets(wunit.reshape(-1), wgen.reshape(-1))
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
        slot_of_c = torch.empty(n, dtype=torch.int64, device=dev)
        slot_of_c[order[b]] = torch.arange(n, device=dev)
        g = b * n_cg + (slot_of_c >> 6)                        # channel -> list group
        _scatter_lists(ell_w16, c, pl, xt[c, pl], n, 2 * wunit.reshape(-1)[g], w_off[2 * g], w_off[2 * g + 1], slot_of_c & 63,
                       xmax_w, PBITS)

    return dict(ell_h=ell_h, ell_h_off=h_off.to(torch.int32), klc=klc, pix_perm=pix_perm.reshape(-1).to(torch.int32), ell_w=ell_w, ell_w_off=w_off.to(torch.int32),
                chan_perm=chan_perm.to(torch.int32), n_cg=n_cg, nblk_w=nblk_w, nnz=nnz,
                entries_h=int(per_px.sum()), entries_w=int(per_ch.sum()), rows_h=rows_h, rows_w=rows_w,
                unit_rows_h=int(gunit.sum()), unit_rows_w=int(wunit.sum()))


def lds_bytes_h(n_pad, k):
    """LDS the sparse H-step needs: the GW table plus the partial numerators of a 512-pixel tile (two sets for k <= 6,
    where the list groups of a window are walked in pairs)."""
    wb = 0 if k <= 4 else (1 if k == 5 else (2 if k == 6 else (4 if k <= 8 else (8 if k <= 12 else 12))))
    sets = 2 if k <= _lib.ELL_PAIR_MAX_K else 1
    kp = 8 if k <= 8 else 16
    # (the tail workgroup's scratch is a lower bound of the second term; the k doubles of the workgroup's copy of colsum(GW), when the W update's tail rides in the launch, go behind these - or, without room
    #  there, into the table once it is dead: launch_h_ell_k)
    return n_pad * (4 + wb) * 4 + max(sets * k * _lib.ELL_TILE * 4, 9 * (5 + 2 * k) * 8, (9 * (kp + 1) + 1) * 8)

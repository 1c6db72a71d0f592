"""Pixel-row sharding of one SmoothNMF fit over the GPUs of a node.

The image is cut into contiguous blocks of image rows, one per rank; X and H are sharded, W, G and
G W are replicated.  Per multiplicative-update iteration each rank contributes ONE record

    [ A = R H^T partial (k * n_pad fp32) | statistics of its new H block (16 fp64: row sums, row maxima)
      | first owned image row of the new H (k * ny fp32) | last owned image row (k * ny fp32) ]

to one all-gather (RCCL over xGMI on the GPU box; the messages are tens of KB, i.e. latency bound,
so a single collective per iteration is the design target).  Every rank then sums the A blocks in
rank order (bit-identical W on all ranks - no broadcast needed), forms the global row sums / maxima,
and reads its neighbours' boundary rows in place as the stencil halo of the next H-step.

This module holds the backend-independent part: the record layout (which must match
``espm_mu_shard_record_bytes`` / ``shard_pack_kernel`` in csrc/mu_aux.hip), the buffers, the
collective and the neighbour arithmetic.  ``MUEngine`` fills and consumes the records with HIP
kernels; the CPU tests do the same with numpy to check the protocol under gloo.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

def hs_stride(k: int) -> int:
    """ESPM_HS_STRIDE of the build that serves k components (2 KP: KP = 8 up to 8 components, else 16)."""
    return 16 if k <= 8 else 32


def split_rows(nx: int, world: int, rank: int):
    """Contiguous row block of ``rank``: (first row, number of rows); the last rank takes the rest."""
    base = nx // world
    row0 = rank * base
    rows = base if rank < world - 1 else nx - row0
    if base < 1:
        raise ValueError(f"cannot shard {nx} image rows over {world} ranks")
    return row0, rows


@dataclass(frozen=True)
class RecordLayout:
    k: int
    n_pad: int
    ny: int
    off_a: int
    off_hstat: int
    off_top: int
    off_bot: int
    nbytes: int

    @property
    def na(self):
        return self.k * self.n_pad


def record_layout(k: int, n_pad: int, ny: int) -> RecordLayout:
    """Byte layout of one rank's record (same arithmetic as espm_mu_shard_record_bytes)."""
    na = k * n_pad
    off_hstat = na * 4
    off_top = off_hstat + hs_stride(k) * 8
    row = k * max(ny, 0) * 4
    off_bot = off_top + row
    nbytes = (off_bot + row + 15) // 16 * 16
    return RecordLayout(k, n_pad, ny, 0, off_hstat, off_top, off_bot, nbytes)


class ShardExchange:
    """Send / receive buffers of the per-iteration all-gather and the neighbour bookkeeping."""

    def __init__(self, group, k, n_pad, ny, with_halo, device):
        self.group = group
        self.world = torch.distributed.get_world_size(group)
        self.rank = torch.distributed.get_rank(group)
        self.layout = record_layout(k, n_pad, ny)
        self.with_halo = bool(with_halo)
        self.send = torch.zeros(self.layout.nbytes, dtype=torch.uint8, device=device)
        self.recv = torch.zeros(self.world * self.layout.nbytes, dtype=torch.uint8, device=device)
        self._use_list = torch.distributed.get_backend(group) == "gloo"

    def gather(self):
        """All ranks' records, in rank order, into ``recv`` (stream-ordered on the current stream)."""
        if self._use_list:
            parts = list(self.recv.view(self.world, self.layout.nbytes).unbind(0))
            torch.distributed.all_gather(parts, self.send, group=self.group)
        else:
            torch.distributed.all_gather_into_tensor(self.recv, self.send, group=self.group)

    def halo_offsets(self):
        """Byte offsets into ``recv`` of (row above my block, row below my block); None at the image edge.

        The row above is the LAST owned row of rank-1, the row below the FIRST owned row of rank+1."""
        if not self.with_halo:
            return None, None
        lay = self.layout
        top = (self.rank - 1) * lay.nbytes + lay.off_bot if self.rank > 0 else None
        bot = (self.rank + 1) * lay.nbytes + lay.off_top if self.rank < self.world - 1 else None
        return top, bot

    # typed views used by the numpy protocol tests (and for debugging)
    def record_views(self, buf, r=0):
        lay = self.layout
        base = r * lay.nbytes
        a = buf[base + lay.off_a: base + lay.off_a + lay.na * 4].view(torch.float32)
        hs = buf[base + lay.off_hstat: base + lay.off_hstat + hs_stride(lay.k) * 8].view(torch.float64)
        row = lay.k * lay.ny * 4
        top = buf[base + lay.off_top: base + lay.off_top + row].view(torch.float32).view(lay.k, max(lay.ny, 0))
        bot = buf[base + lay.off_bot: base + lay.off_bot + row].view(torch.float32).view(lay.k, max(lay.ny, 0))
        return a, hs, top, bot

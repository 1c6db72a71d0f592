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

import ctypes as C
import os
from dataclasses import dataclass

import torch

def hs_stride(k: int) -> int:
    """ESPM_HS_STRIDE of the build that serves k components (2 KP: KP = 8 up to 8 components, 16 up to 16, else 32)."""
    return 16 if k <= 8 else (32 if k <= 16 else 64)


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


def _device_identity(device):
    """What tells two ranks' devices apart even when both call theirs cuda:0 (one visible device per process)."""
    try:
        dev = torch.device(device)
        if dev.type != "cuda":
            return "cpu"
        props = torch.cuda.get_device_properties(dev)
        uuid = getattr(props, "uuid", None)
        return str(uuid) if uuid is not None else f"{os.environ.get('HIP_VISIBLE_DEVICES', '')}:{os.environ.get('ROCR_VISIBLE_DEVICES', '')}:{dev.index}"
    except Exception:
        return "unknown"


class ShardExchange:
    """Send / receive side of the per-iteration record exchange and the neighbour bookkeeping.

    Two transports, same record layout, same result (world records in rank order):

    * ``p2p`` (default on GPUs): the library's one-shot exchange (``espm_xchg_*``, csrc/mu_xchg.hip) - every rank writes its
      record into every peer's mailbox over the direct links and raises a flag; no host-side collective, no host
      synchronisation, and ``espm_mu_iterate_sharded`` runs whole batches of iterations on it.  The mailboxes are mapped
      through hipIpc; the 64-byte handles travel once through the process group.  A start-up self-test (``selftest``: 64
      exchanges of checksummed patterns, ESPM_XCHG_SELFTEST) decides - jointly - whether the transport works; if not, every
      rank falls back to
    * ``collective``: ``torch.distributed.all_gather_into_tensor`` (RCCL over xGMI on the GPU box, gloo in the CPU tests).
    """

    def __init__(self, group, k, n_pad, ny, with_halo, device, lib=None, stream_fn=None, mode=None):
        self.group = group
        self.world = torch.distributed.get_world_size(group)
        self.rank = torch.distributed.get_rank(group)
        self.layout = record_layout(k, n_pad, ny)
        self.with_halo = bool(with_halo)
        self._use_list = torch.distributed.get_backend(group) == "gloo"
        self.lib, self._stream_fn = lib, stream_fn
        self.ctx = None          # espm_xchg* when the one-shot transport is up
        self.seq = C.c_uint32(0)
        self.send = None
        self.spans_devices = False    # the ranks' records cross a link (set by _open_p2p from the ranks' device identities)
        self.selftest_result = None   # what the start-up self-test of the one-shot transport measured (selftest)
        self._recv2, self._gen = None, 0   # collective transport: two receive buffers in turn (the records before the last gather stay readable)
        mode = mode or os.environ.get("ESPM_XCHG", "p2p")
        if mode == "p2p" and lib is not None and torch.device(device).type == "cuda":
            self._open_p2p(device)
        if self.ctx is None:
            self.send = torch.zeros(self.layout.nbytes, dtype=torch.uint8, device=device)
            self._recv2 = [torch.zeros(self.world * self.layout.nbytes, dtype=torch.uint8, device=device) for _ in range(2)]

    # ---- one-shot transport -------------------------------------------------------------------------------------------
    def _open_p2p(self, device):
        lib, nb = self.lib, self.layout.nbytes
        ctx = C.c_void_p()
        ok = lib.espm_xchg_create(self.world, self.rank, nb, C.byref(ctx)) == 0
        handle = (C.c_ubyte * 64)()
        ok = ok and lib.espm_xchg_handle(ctx, handle) == 0
        gathered = [None] * self.world
        torch.distributed.all_gather_object(gathered, (bool(ok), bytes(handle), _device_identity(device)), group=self.group)
        ok = all(g[0] for g in gathered)
        # ranks on more than one device: the flag behind a record is a RELEASE store at system scope unless the environment says otherwise
        # (the relaxed form's ordering over xGMI has never been validated on a multi-GPU node; the granules of the in-launch
        # exchange carry their own arrival and do not depend on this)
        self.spans_devices = len({g[2] for g in gathered}) > 1
        if ok:
            blob = b"".join(g[1] for g in gathered)
            ok = lib.espm_xchg_connect(ctx, C.create_string_buffer(blob, len(blob))) == 0
        want = 0
        by_order = {}
        if ok:   # self-test: exchanges of patterns every rank can check word by word (selftest below), timed
            self.ctx = ctx
            try:
                # under BOTH flag orders (csrc/mu_xchg.hip, the ordering contract): the one the run will use last; on a node whose
                # relaxed form lets a record arrive after its flag (`corrupt` > 0) while the release form does not, every rank moves
                # to the release form - jointly, below - instead of giving the transport up
                want = int(lib.espm_xchg_order(ctx))
                if self.spans_devices and "ESPM_XCHG_ORDER" not in os.environ:
                    want = 1
                n_test = int(os.environ.get("ESPM_XCHG_SELFTEST", "64"))
                for order in ((1 - want), want):
                    lib.espm_xchg_set_order(ctx, order)
                    by_order["release" if order else "relaxed"] = self.selftest(n_test if order == want else max(16, n_test // 4), device=device)
                self.selftest_result = dict(by_order["release" if want else "relaxed"], order="release" if want else "relaxed", orders=by_order)
                ok = self.selftest_result["lost"] == 0 and self.selftest_result["corrupt"] == 0
                if not ok and not want and by_order["release"]["lost"] == 0 and by_order["release"]["corrupt"] == 0:
                    want, ok = 1, True     # (proposed; adopted below only if every rank ends up healthy)
            except Exception:
                ok, want = False, 0
            self.ctx = None
        flags = [None] * self.world
        torch.distributed.all_gather_object(flags, (bool(ok), int(want) if ok else 0), group=self.group)
        if all(f[0] for f in flags):
            self.ctx = ctx
            if any(f[1] for f in flags):   # some rank needs (or was asked for) the release form: every rank takes it
                lib.espm_xchg_set_order(ctx, 1)
                # the headline counters are the ADOPTED order's; the relaxed run's stay under orders["relaxed"] (a healthy
                # transport used to report the relaxed run's corrupt > 0 under the label "release")
                self.selftest_result = dict(by_order.get("release") or self.selftest_result or {}, order="release", orders=by_order)
        else:   # every rank takes the collective
            if ctx:
                lib.espm_xchg_destroy(ctx)
            self.seq.value = 0
            self.selftest_result = dict(self.selftest_result or {}, transport="collective", fell_back_from="p2p")

    def selftest(self, n=1000, device=None):
        """``n`` exchanges of records filled with a pattern that depends on the sending rank AND the sequence number, every byte
        of every received record compared with what its sender must have written, each exchange timed with HIP events on the
        launch stream (post + wait: what a rank pays between having its record and having everybody's).  What this is for: the
        one-shot transport orders its write-through stores and the flag with ``s_waitcnt vmcnt(0)``, not with a system-scope
        fence (csrc/mu_xchg.hip), and the peers' mailboxes are mapped through hipIpc - neither can be validated on a box with
        one GPU, so every start on real peers is a test: a record that arrives after its flag shows up as ``corrupt`` (the
        pattern of the PREVIOUS use of the slot, two sequence numbers back), a flag that never arrives as ``lost``.
        One GPU (a group of one rank, or ranks sharing a device): the same kernels and flags, no link crossed.
        Returns dict(transport, n, p50_us, p99_us, max_us, lost, corrupt)."""
        if self.ctx is None:
            return dict(transport="collective", n=0, p50_us=None, p99_us=None, max_us=None, lost=0, corrupt=0)
        lib, nb = self.lib, self.layout.nbytes
        device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        stage = _as_tensor(lib.espm_xchg_staging(self.ctx), nb, device)
        keep = stage.clone()
        bad = torch.zeros((), dtype=torch.int64, device=device)
        ranks = torch.arange(self.world, device=device, dtype=torch.int32).view(self.world, 1)
        views = [_as_tensor(lib.espm_xchg_records(self.ctx, par), nb * self.world, device).view(self.world, nb) for par in (0, 1)]
        lost0 = self.lost_peers()
        events = []
        s = self._stream_fn()
        for i in range(int(n)):
            if i in (1, 16) or (i and i % 256 == 0):   # a transport that does not deliver costs a 2 s bounded wait per exchange: stop early
                torch.cuda.synchronize()
                if self.lost_peers() - lost0 > 0 or int(bad.item()) > 0:
                    break
            self.seq.value += 1
            q = int(self.seq.value)
            stage.fill_((self.rank * 37 + q * 11 + 1) & 255)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.espm_xchg_post(self.ctx, self.seq, s) or lib.espm_xchg_wait(self.ctx, self.seq, s)
            e1.record()
            if rc:
                raise RuntimeError(lib.espm_mu_last_error().decode())
            events.append((e0, e1))
            expect = ((ranks * 37 + (q * 11 + 1)) & 255).to(torch.uint8)
            bad += (views[q & 1] != expect).sum()
        torch.cuda.synchronize()
        stage.copy_(keep)
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in events)
        pick = (lambda f: us[min(len(us) - 1, int(f * len(us)))]) if us else (lambda f: None)
        return dict(transport="p2p", n=len(events), p50_us=pick(0.5), p99_us=pick(0.99), max_us=us[-1] if us else None,
                    lost=self.lost_peers() - lost0, corrupt=int(bad.item()))

    @property
    def transport(self):
        return "p2p" if self.ctx is not None else "collective"

    @property
    def send_ptr(self):
        """Where this rank's record is packed."""
        return int(self.lib.espm_xchg_staging(self.ctx)) if self.ctx is not None else self.send.data_ptr()

    @property
    def recv(self):
        """Collective transport: the buffer the last ``gather`` filled."""
        return self._recv2[self._gen & 1] if self._recv2 is not None else None

    @property
    def recv_ptr(self):
        """The records of all ranks, in rank order, after ``gather``."""
        return int(self.lib.espm_xchg_records(self.ctx, self.seq.value & 1)) if self.ctx is not None else self.recv.data_ptr()

    @property
    def prev_recv_ptr(self):
        """The records of the gather BEFORE the last one (both transports are double buffered): the neighbours' boundary rows of
        the H that preceded the last update are still there (linesearch of a sharded image)."""
        if self.ctx is not None:
            return int(self.lib.espm_xchg_records(self.ctx, (self.seq.value - 1) & 1))
        return self._recv2[(self._gen - 1) & 1].data_ptr()

    def gather(self):
        """All ranks' records, in rank order (stream-ordered on the current stream)."""
        if self.ctx is not None:
            s = self._stream_fn()
            self.seq.value += 1
            rc = self.lib.espm_xchg_post(self.ctx, self.seq, s) or self.lib.espm_xchg_wait(self.ctx, self.seq, s)
            if rc:
                raise RuntimeError(self.lib.espm_mu_last_error().decode())
        elif self._use_list:
            self._gen += 1
            parts = list(self.recv.view(self.world, self.layout.nbytes).unbind(0))
            torch.distributed.all_gather(parts, self.send, group=self.group)
        else:
            self._gen += 1
            torch.distributed.all_gather_into_tensor(self.recv, self.send, group=self.group)

    def lost_peers(self):
        """Waits that gave up (a peer never delivered) since the exchange was opened: 0 on a healthy node."""
        if self.ctx is None:
            return 0
        lost = C.c_uint32(0)
        self.lib.espm_xchg_timeouts(self.ctx, C.byref(lost))
        return int(lost.value)

    def close(self):
        if self.ctx is not None:
            self.lib.espm_xchg_destroy(self.ctx)
            self.ctx = None

    def halo_offsets(self):
        """Byte offsets into the gathered records of (row above my block, row below my block); None at the image edge.

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


def _as_tensor(ptr, nbytes, device):
    """A uint8 torch view of `nbytes` of device memory the library owns (the exchange's staging record / mailbox)."""
    class _Mem:
        pass
    m = _Mem()
    m.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(m, device=device)

"""Ranks as THREADS of one process, for geometries whose rank count the GPU box does not allow as processes.

A GPU box admits at most six processes on its card; BASELINE configuration 4 at eight GPUs has eight ranks.  What such a
test must exercise is the shard geometry (64-row shards: the run-time-sized fused kernel on 128-pixel blocks), the records,
`espm_mu_w_reduce_pack` and the sum over eight records in `espm_mu_shard_combine_finish` - all of it library code that does
not care who carries the records.  Here every rank is a thread with its own ``MUEngine`` on the one device, and the handful of
``torch.distributed`` calls the engine and ``espm_amd.sharding`` make on a group (world size, rank, all_reduce, all_gather)
are served, for groups of this module's type only, by a rendezvous between the threads: deposit, barrier, combine in RANK
ORDER, barrier.  All threads enqueue on the device's one default stream in the order the host executes them, and every
deposit precedes its barrier: what a rank reads was enqueued before.  The transport this stands in for is the collective one
(``all_gather_into_tensor``); the one-shot exchange needs real peers.
"""
import threading

import torch
import torch.distributed as dist


class _Shared:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class ThreadRank:
    """What a thread hands over wherever a process group is expected."""

    def __init__(self, shared, rank):
        self.shared, self.rank = shared, rank

    def _exchange(self, value):
        s = self.shared
        s.slots[self.rank] = value
        s.barrier.wait()
        vals = list(s.slots)
        s.barrier.wait()       # nobody deposits the next round's value before everybody has read this one's
        return vals


_PATCHED = {}


def _patch():
    if _PATCHED:
        return
    orig = {name: getattr(dist, name) for name in ("get_world_size", "get_rank", "get_backend", "all_reduce", "all_gather_into_tensor",
                                                   "all_gather", "all_gather_object", "barrier", "broadcast")}
    _PATCHED.update(orig)

    def get_world_size(group=None):
        return group.shared.world if isinstance(group, ThreadRank) else orig["get_world_size"](group)

    def get_rank(group=None):
        return group.rank if isinstance(group, ThreadRank) else orig["get_rank"](group)

    def get_backend(group=None):
        return "threads" if isinstance(group, ThreadRank) else orig["get_backend"](group)

    def all_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if not isinstance(group, ThreadRank):
            return orig["all_reduce"](tensor, op=op, group=group, async_op=async_op)
        vals = group._exchange(tensor.clone())
        acc = vals[0].clone()
        for v in vals[1:]:      # rank order on every rank: the same bits everywhere
            if op == dist.ReduceOp.SUM:
                acc += v
            elif op == dist.ReduceOp.MAX:
                acc = torch.maximum(acc, v)
            elif op == dist.ReduceOp.MIN:
                acc = torch.minimum(acc, v)
            else:
                raise NotImplementedError(op)
        tensor.copy_(acc)

    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        if not isinstance(group, ThreadRank):
            return orig["all_gather_into_tensor"](out, inp, group=group, async_op=async_op)
        vals = group._exchange(inp.clone())
        out.view(len(vals), -1).copy_(torch.stack([v.reshape(-1) for v in vals]))

    def all_gather(tensor_list, tensor, group=None, async_op=False):
        if not isinstance(group, ThreadRank):
            return orig["all_gather"](tensor_list, tensor, group=group, async_op=async_op)
        for dst, v in zip(tensor_list, group._exchange(tensor.clone())):
            dst.copy_(v)

    def all_gather_object(object_list, obj, group=None):
        if not isinstance(group, ThreadRank):
            return orig["all_gather_object"](object_list, obj, group=group)
        object_list[:] = group._exchange(obj)

    def barrier(group=None, **kw):
        if not isinstance(group, ThreadRank):
            return orig["barrier"](group=group, **kw)
        group._exchange(None)

    def broadcast(tensor, src=0, group=None, async_op=False):
        if not isinstance(group, ThreadRank):
            return orig["broadcast"](tensor, src=src, group=group, async_op=async_op)
        vals = group._exchange(tensor.clone() if group.rank == src else None)
        tensor.copy_(vals[src])

    for name, fn in dict(get_world_size=get_world_size, get_rank=get_rank, get_backend=get_backend, all_reduce=all_reduce,
                         all_gather_into_tensor=all_gather_into_tensor, all_gather=all_gather, all_gather_object=all_gather_object,
                         barrier=barrier, broadcast=broadcast).items():
        setattr(dist, name, fn)


def _unpatch():
    for name, fn in _PATCHED.items():
        setattr(dist, name, fn)
    _PATCHED.clear()


def run_ranks(world, fn):
    """fn(group, rank) on `world` threads; returns their results in rank order; the first exception of any rank is re-raised
    (a rank that fails breaks the barrier, so the others do not wait for it)."""
    shared = _Shared(world)
    results, errors = [None] * world, []
    dev = torch.cuda.current_device()

    def body(r):
        try:
            torch.cuda.set_device(dev)
            results[r] = fn(ThreadRank(shared, r), r)
        except BaseException as e:   # noqa: BLE001
            errors.append((r, e))
            shared.barrier.abort()

    _patch()
    try:
        threads = [threading.Thread(target=body, args=(r,), name=f"rank{r}") for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        _unpatch()
    real = [e for e in errors if not isinstance(e[1], threading.BrokenBarrierError)] or errors
    if real:
        raise RuntimeError(f"rank {real[0][0]} failed") from real[0][1]
    return results

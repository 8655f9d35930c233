"""W data-parallel ranks inside ONE process (threads), for rehearsing world sizes the GPU box's process guard does not allow
(at most 6 processes may use the card: 8 gloo ranks on one card cannot run).  Every virtual rank runs the PRODUCT code
(``DataParallel.loss_and_grads`` -> ``ElboEngine`` -> the five-piece C entry ``dsvgp_elbo_step_dp_f32``) with its own engine,
plan and workspace; only the transport differs: the collectives meet at ``threading.Barrier``s and are computed by torch on the
one device (all ranks issue on the same stream, so device order = host order).  Sums are taken in rank order, the same on every
rank: results are bitwise equal across the virtual ranks, as RCCL's are.

The library's contract is ONE host thread per context (include/dsvgp.h; the one-call step re-points the context's stream while it
queues its side-stream work), and the ranks of one process share the device's context: the ranks therefore run ONE AT A TIME --
``VirtualWorld.turn`` is held by the rank that is queueing work and handed over only while it waits inside a collective."""
import threading

import torch


class _Done:
    def wait(self):
        return None


class VirtualWorld:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.errors = []
        self.turn = threading.Lock()        # held by the one rank that is inside the library

    def wait(self):
        """barrier; the caller's turn is handed over while it waits"""
        self.turn.release()
        try:
            self.barrier.wait()
        finally:
            self.turn.acquire()

    def exchange(self, rank, value):
        """every rank deposits ``value``; returns the list of all ranks' values (valid until the next exchange)"""
        self.slots[rank] = value
        self.wait()
        vals = list(self.slots)
        self.wait()
        return vals

    def run(self, fn):
        """fn(rank, world) on `world` threads; re-raises the first exception"""
        out = [None] * self.world

        def body(r):
            self.turn.acquire()
            try:
                out[r] = fn(r, self)
            except BaseException as ex:      # noqa: BLE001  (a failed rank must not leave the others waiting)
                self.errors.append((r, ex))
                self.barrier.abort()
            finally:
                self.turn.release()

        ts = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        real = [e for e in self.errors if not isinstance(e[1], threading.BrokenBarrierError)]
        if real or self.errors:
            raise (real or self.errors)[0][1]
        return out


def make_virtual_dp(dsvgp_amd, vw, rank):
    """a ``DataParallel`` whose transport is the VirtualWorld (no torch.distributed process group)"""

    class VirtualDP(dsvgp_amd.DataParallel):
        def __init__(self):          # (the parent's constructor asks torch.distributed for rank and world)
            self.group, self.world, self.rank = None, vw.world, rank
            self.global_batch, self.algo, self.rs_ag_min_numel, self._shards = None, "allreduce", 1 << 16, {}
            self.src0, self.replicated_step = 0, False
            self.check_every, self._check_step, self.divergences = 0, 0, 0

        def all_reduce_sum(self, t):
            vals = vw.exchange(rank, t)
            total = vals[0].clone()
            for v in vals[1:]:
                total += v
            vw.wait()                   # (every rank has read the operands before anyone overwrites its own)
            t.copy_(total)
            vw.wait()

        def all_reduce_async(self, t):
            self.all_reduce_sum(t)
            return _Done()

        def all_gather_async(self, out, inp):
            vals = vw.exchange(rank, inp)
            n = inp.numel()
            flat = out.view(-1)
            for r, v in enumerate(vals):
                flat[r * n:(r + 1) * n].copy_(v.reshape(-1))
            vw.wait()
            return _Done()

    return VirtualDP()

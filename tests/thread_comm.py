"""TEST HELPER: an in-process stand-in for ``animsnapbases_amd.distributed.Comm`` that runs
``world`` "ranks" as threads of ONE process on ONE GPU.  Every rank owns its own HipEngine
(its own asb_ctx and vertex shard) on the device's null stream; collectives are emulated by a
barrier-synchronised exchange.  This exercises the REAL multi-rank HIP path (global vertex
ids, padded candidate gathers, asb_panel_assemble, record exchange, Gram all-reduces) on the
single-GPU box, where RCCL with several ranks cannot run.  The torch.distributed wiring itself
is covered by the gloo CPU tests."""
import threading

import numpy as np

from animsnapbases_amd.distributed import partition


class _Hub(object):
    def __init__(self, world):
        self.world = world
        self.slots = [None] * world
        self.bar = threading.Barrier(world, timeout=120)      # a collective that not every rank enters fails instead of hanging

    def exchange(self, rank, obj):
        self.slots[rank] = obj
        self.bar.wait()
        out = list(self.slots)
        self.bar.wait()
        return out


class ThreadComm(object):
    def __init__(self, hub, rank):
        import torch

        self.hub, self.rank, self.world = hub, rank, hub.world
        self.multi = self.world > 1
        self._torch = torch
        self._dev = torch.device("cuda", 0)
        self.on_engine_stream = False       # the emulated collectives run on torch's stream behind device-wide syncs

    def shards(self, N):
        return partition(N, self.world)

    def my_shard(self, N):
        return self.shards(N)[self.rank]

    def _sync(self):
        self._torch.cuda.synchronize()

    def allreduce_sum(self, values):
        a = np.atleast_1d(np.asarray(values, dtype=np.float64))
        return np.sum(self.hub.exchange(self.rank, a.copy()), axis=0)

    def allreduce_max(self, values):
        a = np.atleast_1d(np.asarray(values, dtype=np.float64))
        return np.max(self.hub.exchange(self.rank, a.copy()), axis=0)

    def all_gather_ints(self, values):
        a = np.atleast_1d(np.asarray(values, dtype=np.int64))
        return np.stack(self.hub.exchange(self.rank, a.copy()))

    def exchange_device(self, on_device):
        return self._dev

    def allreduce_tensor(self, t):
        self._sync()
        parts = self.hub.exchange(self.rank, t.clone())
        t.copy_(sum(parts[1:], parts[0]))
        self._sync()
        self.hub.bar.wait()

    def allreduce_min_tensor(self, t):
        self._sync()
        parts = self.hub.exchange(self.rank, t.clone())
        m = parts[0]
        for q in parts[1:]:
            m = self._torch.minimum(m, q)
        t.copy_(m)
        self._sync()
        self.hub.bar.wait()

    def all_gather_into(self, out, inp):
        self._sync()
        parts = self.hub.exchange(self.rank, inp.clone())
        out.copy_(self._torch.cat([p.reshape(-1) for p in parts]))
        self._sync()
        self.hub.bar.wait()

    def new_records(self, xlen, on_device):
        torch = self._torch
        return (torch.zeros(xlen, dtype=torch.float64, device=self._dev),
                torch.zeros(self.world * xlen, dtype=torch.float64, device=self._dev))

    def all_gather_records(self, rec, recs):
        self.all_gather_into(recs, rec)

    def new_gram_buffers(self, F, K, on_device):
        torch = self._torch
        return (torch.zeros(F * K, dtype=torch.float64, device=self._dev),
                torch.zeros(K * K, dtype=torch.float64, device=self._dev))

    def new_buffer(self, n, on_device):
        return self._torch.zeros(int(n), dtype=self._torch.float64, device=self._dev)

    def global_argmax(self, idx, val):
        allv = self.hub.exchange(self.rank, (val.copy(), idx.copy()))
        best = np.empty(idx.shape[0], dtype=np.int64)
        for k in range(idx.shape[0]):
            order = sorted(range(self.world), key=lambda r: (-allv[r][0][k], allv[r][1][k]))
            best[k] = int(allv[order[0]][1][k])
        return best

    def all_gather_rows(self, local, N, axis):
        parts = self.hub.exchange(self.rank, np.ascontiguousarray(local))
        return np.concatenate(parts, axis=axis)

    def barrier(self):
        self.hub.bar.wait()


def run_ranks(world, fn):
    """Runs fn(rank, comm) on `world` threads; re-raises the first failure."""
    hub = _Hub(world)
    errs, outs = [], [None] * world

    def body(r):
        try:
            outs[r] = fn(r, ThreadComm(hub, r))
        except BaseException as e:          # noqa: BLE001 - surfaced to the test below
            errs.append(e)
            hub.bar.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    return outs

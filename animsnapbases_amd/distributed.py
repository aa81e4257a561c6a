"""Vertex-row sharding over the GPUs of one node.

One process per GPU.  ``Comm`` wraps ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests); with no process group it degenerates to a
single rank and never imports torch.

What crosses ranks on the deflation path is tiny and latency-bound: per component one
all-gather of ``[energy, idx, 3 x F slab]`` records (<= 48 KB at F = 2000), plus scalar
all-reduces during standardisation and one all-reduce of the K residual norms at the end.
The snapshot tensor itself never moves.
"""
import os

import numpy as np


def partition(N, world):
    """Contiguous vertex ranges: the first ``N % world`` ranks get one extra vertex.
    Returns the list of (v0, n_loc)."""
    base, extra = divmod(int(N), int(world))
    out, v0 = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((v0, n))
        v0 += n
    return out


class Comm(object):
    def __init__(self, group=None, force_single=False, force_collectives=None):
        """force_collectives (default: env ASB_FORCE_COLLECTIVES=1): take the multi-rank code path -- every
        collective really issued -- even when the process group has a single rank.  That is how the RCCL wiring
        (device tensors, stream ordering against the engine's kernels) is tested on a one-GPU box."""
        self.group = group
        self.dist = None
        self.rank, self.world = 0, 1
        if force_collectives is None:
            force_collectives = os.environ.get("ASB_FORCE_COLLECTIVES", "0") == "1"
        if not force_single:
            try:
                import torch.distributed as dist
            except Exception:          # torch absent: single rank
                dist = None
            if dist is not None and dist.is_available() and dist.is_initialized():
                self.dist = dist
                self.rank = dist.get_rank(group)
                self.world = dist.get_world_size(group)
        self.multi = self.world > 1 or bool(force_collectives and self.dist is not None)
        self._torch = None
        self._dev = None
        if self.multi:
            import torch

            self._torch = torch
            backend = self.dist.get_backend(group)
            self._dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")

    # ------------------------------------------------------------ partition
    def shards(self, N):
        return partition(N, self.world)

    def my_shard(self, N):
        return self.shards(N)[self.rank]

    # ------------------------------------------------------------ collectives
    def allreduce_sum(self, values):
        """Sum of a small float64 vector over ranks (returned as ndarray)."""
        a = np.atleast_1d(np.asarray(values, dtype=np.float64)).copy()
        if not self.multi:
            return a
        t = self._torch.from_numpy(a).to(self._dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def allreduce_max(self, values):
        a = np.atleast_1d(np.asarray(values, dtype=np.float64)).copy()
        if not self.multi:
            return a
        t = self._torch.from_numpy(a).to(self._dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t.cpu().numpy()

    def all_gather_ints(self, values):
        """(world, len(values)) int64 array of every rank's small integer vector."""
        a = np.atleast_1d(np.asarray(values, dtype=np.int64)).copy()
        if not self.multi:
            return a[None]
        t = self._torch.from_numpy(a).to(self._dev)
        out = self._torch.empty(self.world * a.shape[0], dtype=self._torch.int64, device=self._dev)
        self.dist.all_gather_into_tensor(out, t, group=self.group)
        return out.cpu().numpy().reshape(self.world, a.shape[0])

    def _need_torch(self):
        if self._torch is None:
            import torch

            self._torch = torch
            self._dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        return self._torch

    def exchange_device(self, on_device):
        torch = self._need_torch()
        return self._dev if on_device else torch.device("cpu")

    def all_gather_into(self, out, inp):
        if not self.multi:
            out.copy_(inp)
            return
        self.dist.all_gather_into_tensor(out, inp, group=self.group)

    def new_records(self, xlen, on_device):
        """(rec, recs): one exchange record and the gathered (world, xlen) buffer."""
        torch = self._torch
        dev = self._dev if on_device else torch.device("cpu")
        rec = torch.zeros(xlen, dtype=torch.float64, device=dev)
        recs = torch.zeros(self.world * xlen, dtype=torch.float64, device=dev)
        return rec, recs

    def all_gather_records(self, rec, recs):
        if rec.device.type != self._dev.type:      # e.g. CPU records with an nccl group (not used by the product)
            r, rs = rec.to(self._dev), recs.to(self._dev)
            self.dist.all_gather_into_tensor(rs, r, group=self.group)
            recs.copy_(rs)
            return
        self.dist.all_gather_into_tensor(recs, rec, group=self.group)

    def new_gram_buffers(self, F, K, on_device):
        """Device buffers for the partial Gram matrices P (F x K) and M (K x K) of SPLOCS."""
        torch = self._torch
        dev = self._dev if on_device else torch.device("cpu")
        return (torch.zeros(F * K, dtype=torch.float64, device=dev), torch.zeros(K * K, dtype=torch.float64, device=dev))

    def new_buffer(self, n, on_device):
        torch = self._need_torch()
        return torch.zeros(int(n), dtype=torch.float64, device=self.exchange_device(on_device))

    def allreduce_tensor(self, t):
        if self.multi:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def global_argmax(self, idx, val):
        """Per entry k: the (val, idx) pair with the largest val over ranks, lowest idx on ties."""
        if not self.multi:
            return idx
        torch = self._torch
        loc = torch.from_numpy(np.stack([val, idx.astype(np.float64)])).to(self._dev)
        outs = [torch.empty_like(loc) for _ in range(self.world)]
        self.dist.all_gather(outs, loc, group=self.group)
        allv = np.stack([o.cpu().numpy() for o in outs])          # (world, 2, K)
        best = np.empty(idx.shape[0], dtype=np.int64)
        for k in range(idx.shape[0]):
            order = sorted(range(self.world), key=lambda r: (-allv[r, 0, k], allv[r, 1, k]))
            best[k] = int(allv[order[0], 1, k])
        return best

    def all_gather_rows(self, local, N, axis):
        """Concatenates per-rank blocks along ``axis`` (block sizes follow ``partition``)."""
        if not self.multi:
            return local
        torch = self._torch
        shards = self.shards(N)
        nmax = max(n for _, n in shards)
        loc = np.moveaxis(np.ascontiguousarray(local), axis, 0)
        pad = np.zeros((nmax,) + loc.shape[1:], dtype=loc.dtype)
        pad[:loc.shape[0]] = loc
        t = torch.from_numpy(pad).to(self._dev)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t, group=self.group)
        parts = [o.cpu().numpy()[:n] for o, (_, n) in zip(outs, shards)]
        return np.moveaxis(np.concatenate(parts, axis=0), 0, axis)

    def barrier(self):
        if self.multi:
            self.dist.barrier(group=self.group)

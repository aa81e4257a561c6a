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
        self._bufs = {}
        # the stream RCCL collectives are enqueued on (torch's current stream when the communicator was made): a consumer
        # may rely on stream order between a collective and an engine's kernels only if the engine runs on this very stream
        self.stream_handle = None
        # STAGED collectives: a gloo group with the shards on GPUs (several processes on ONE GPU: the launch rehearsal of
        # bench.py --gpus N on the one-GPU box, tests/test_gpu_launch.py).  Exchange tensors stay device tensors -- the engine
        # writes and reads them with kernels --, every collective synchronises the device, runs on a host copy and copies back.
        self._staged = False
        if self.multi:
            import torch

            self._torch = torch
            backend = self.dist.get_backend(group)
            self._staged = bool(backend == "gloo" and torch.cuda.is_available() and os.environ.get("ASB_GLOO_STAGED", "1") != "0")
            self._dev = torch.device("cuda", torch.cuda.current_device()) if (backend == "nccl" or self._staged) else torch.device("cpu")
            if backend == "nccl":
                self.stream_handle = int(torch.cuda.current_stream().cuda_stream)

    # ------------------------------------------------------------ the two primitives everything below is made of
    def _all_reduce(self, t, op):
        if self._staged and t.is_cuda:
            self._torch.cuda.synchronize()
            h = t.cpu()
            self.dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
            return
        self.dist.all_reduce(t, op=op, group=self.group)

    def _all_gather(self, out, inp):
        if self._staged and (out.is_cuda or inp.is_cuda):
            self._torch.cuda.synchronize()
            h = self._torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(h, inp.cpu(), group=self.group)
            out.copy_(h)
            return
        self.dist.all_gather_into_tensor(out, inp, group=self.group)

    def on_engine_stream(self, eng):
        """True iff collectives issued now are stream-ordered against ``eng``'s kernels (same HIP stream)."""
        if self.stream_handle is None or getattr(eng, "stream_handle", None) is None:
            return False
        try:
            cur = int(self._torch.cuda.current_stream().cuda_stream)
        except Exception:
            return False
        return cur == self.stream_handle == eng.stream_handle

    # ------------------------------------------------------------ partition
    def shards(self, N):
        return partition(N, self.world)

    def my_shard(self, N):
        return self.shards(N)[self.rank]

    # ------------------------------------------------------------ collectives
    # Small host-side values (scalars of the standardisation, the K residual norms, a panel's counters): ONE exchange
    # tensor per (kind, length) lives on the exchange device for the life of the communicator, the value is written into
    # it, reduced / gathered in place with all_reduce / all_gather_into_tensor, and read back once -- no per-call tensor
    # construction, no Python lists of per-rank tensors.
    def _scratch(self, key, n, dtype):
        buf = self._bufs.get((key, n))
        if buf is None:
            buf = self._torch.empty(n, dtype=dtype, device=self._dev)
            self._bufs[(key, n)] = buf
        return buf

    def _reduce(self, values, op):
        a = np.atleast_1d(np.asarray(values, dtype=np.float64))
        if not self.multi:
            return a.copy()
        t = self._scratch("red", a.size, self._torch.float64)
        t.copy_(self._torch.from_numpy(np.ascontiguousarray(a).reshape(-1)), non_blocking=False)
        self._all_reduce(t, op)
        return t.cpu().numpy().reshape(a.shape)

    def allreduce_sum(self, values):
        """Sum of a small float64 vector over ranks (returned as ndarray)."""
        return self._reduce(values, self.dist.ReduceOp.SUM if self.multi else None)

    def allreduce_max(self, values):
        return self._reduce(values, self.dist.ReduceOp.MAX if self.multi else None)

    def all_gather_ints(self, values):
        """(world, len(values)) int64 array of every rank's small integer vector."""
        a = np.atleast_1d(np.asarray(values, dtype=np.int64))
        if not self.multi:
            return a[None].copy()
        n = a.shape[0]
        t = self._scratch("gi_in", n, self._torch.int64)
        out = self._scratch("gi_out", self.world * n, self._torch.int64)
        t.copy_(self._torch.from_numpy(np.ascontiguousarray(a)))
        self._all_gather(out, t)
        return out.cpu().numpy().reshape(self.world, n)

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
        self._all_gather(out, inp)

    def allreduce_min_tensor(self, t):
        """In-place MIN over ranks of a (device) tensor; the caller reads it."""
        if self.multi:
            self._all_reduce(t, self.dist.ReduceOp.MIN)

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
            self._all_gather(rs, r)
            recs.copy_(rs)
            return
        self._all_gather(recs, rec)

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
            self._all_reduce(t, self.dist.ReduceOp.SUM)

    def global_argmax(self, idx, val):
        """Per entry k: the (val, idx) pair with the largest val over ranks, lowest idx on ties (NumPy's first-max rule
        across the contiguous shards)."""
        if not self.multi:
            return idx
        n = idx.shape[0]
        loc = self._scratch("am_in", 2 * n, self._torch.float64)
        out = self._scratch("am_out", 2 * n * self.world, self._torch.float64)
        loc.copy_(self._torch.from_numpy(np.concatenate([np.asarray(val, dtype=np.float64), idx.astype(np.float64)])))
        self._all_gather(out, loc)
        allv = out.view(self.world, 2, n)
        vals, ids = allv[:, 0, :], allv[:, 1, :]
        top = vals.max(dim=0).values
        cand = self._torch.where(vals == top[None, :], ids, self._torch.full_like(ids, float("inf")))
        return cand.min(dim=0).values.to(self._torch.int64).cpu().numpy()

    def all_gather_rows(self, local, N, axis):
        """Concatenates per-rank blocks along ``axis`` (block sizes follow ``partition``): one padded exchange tensor,
        one all_gather_into_tensor, one copy back."""
        if not self.multi:
            return local
        torch = self._torch
        shards = self.shards(N)
        nmax = max(n for _, n in shards)
        loc = np.moveaxis(np.ascontiguousarray(local), axis, 0)
        rest = loc.shape[1:]
        per = int(np.prod(rest, dtype=np.int64)) if rest else 1
        piece = torch.zeros(nmax * per, dtype=torch.float64, device=self._dev)
        piece[:loc.shape[0] * per].copy_(torch.from_numpy(np.ascontiguousarray(loc, dtype=np.float64).reshape(-1)))
        out = torch.empty(self.world * nmax * per, dtype=torch.float64, device=self._dev)
        self._all_gather(out, piece)
        full = out.cpu().numpy().reshape((self.world, nmax) + rest)
        parts = [full[r, :n] for r, (_, n) in enumerate(shards)]
        res = np.concatenate(parts, axis=0)
        if local.dtype != np.float64:
            res = res.astype(local.dtype)
        return np.moveaxis(res, 0, axis)

    def barrier(self):
        if self.multi:
            self.dist.barrier(group=self.group)

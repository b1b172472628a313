"""Multi-GPU sharding of the batched learner.

Books are independent given theta, so each rank owns a contiguous shard of
books (its own event streams, traces and theta replica).  The only exchange is
the shared weight vector: every ``sync_every`` steps each rank forms
delta = theta - theta_sync, the deltas are summed over ranks (RCCL all-reduce
over xGMI on the GPU box, gloo in the CPU tests) and every rank sets
theta = theta_sync + sum(delta).  This is the batched analogue of the
reference's unlocked shared-Agent threads (reference src/main.cpp:196-206) at
sync granularity.

``backend`` is anything with
    td_step(n), delta_init(), delta_tensor() -> torch tensor viewing the local
    delta (after computing it), delta_apply()
so the same driver runs the HIP engine (bench.py) and, in tests, a CPU stand-in.
"""


def shard_books(total_books, world_size, rank):
    """Contiguous shard [first, first + n) of `total_books` for `rank`."""
    base, extra = divmod(total_books, world_size)
    n = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, n


class ShardedLearner:
    def __init__(self, backend, dist=None, sync_every=64, single_rank_sync=False):
        """`single_rank_sync` keeps the exchange on even with one rank (a 1-GPU box can then run the
        whole multi-GPU code path of bench.py: delta kernels, staging, RCCL all-reduce)."""
        self.backend = backend
        self.dist = dist if (dist is not None and dist.is_initialized() and
                             (dist.get_world_size() > 1 or single_rank_sync)) else None
        self.sync_every = int(sync_every)
        self.steps = 0
        self.n_syncs = 0
        if self.dist is not None:
            backend.delta_init()

    def sync_weights(self):
        t = self.backend.delta_tensor()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        self.backend.after_all_reduce()
        self.backend.delta_apply()
        self.n_syncs += 1

    def run(self, n_steps):
        done = 0
        while done < n_steps:
            chunk = n_steps - done
            if self.dist is not None:
                chunk = min(chunk, self.sync_every - self.steps % self.sync_every)
            self.backend.td_step(chunk)
            done += chunk
            self.steps += chunk
            if self.dist is not None and self.steps % self.sync_every == 0:
                self.sync_weights()


class EngineBackend:
    """HIP engine as a ShardedLearner backend (the delta buffer lives in HBM and
    is handed to torch.distributed through __cuda_array_interface__)."""

    class _DevArray:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def __init__(self, eng, torch_mod, device, stage=True):
        self.eng, self.torch, self.device = eng, torch_mod, device
        # The collective runs on a tensor from torch's own allocator (RCCL sees only memory it
        # could register itself); the engine's buffer is copied in and out on the device, 2 x 160 MB
        # per exchange at M = 20M, i.e. ~0.1 ms every sync_every steps.  stage=False hands RCCL the
        # engine's buffer directly.
        self.use_stage, self.stage, self.view = stage, None, None

    def td_step(self, n):
        self.eng.td_step(n)

    def delta_init(self):
        self.eng.delta_init()

    def delta_tensor(self):
        ptr, n = self.eng.delta_begin()  # synchronises the engine stream
        self.view = self.torch.as_tensor(self._DevArray(ptr, n), device=self.device)
        if not self.use_stage:
            return self.view
        if self.stage is None or self.stage.numel() != n:
            self.stage = self.torch.empty(n, dtype=self.torch.float64, device=self.device)
        self.stage.copy_(self.view)
        return self.stage

    def after_all_reduce(self):
        if self.use_stage:
            self.view.copy_(self.stage)
        self.torch.cuda.synchronize()

    def delta_apply(self):
        self.eng.delta_apply()

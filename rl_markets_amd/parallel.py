"""Multi-GPU sharding of the batched learner.

Books are independent given theta, so each rank owns a contiguous shard of
books (its own event streams, traces and theta replica).  The only exchange is
the shared weight vector: every ``sync_every`` steps each rank forms
delta = theta - theta_sync, the deltas are summed over ranks and every rank
sets theta = theta_sync + sum(delta).  This is the batched analogue of the
reference's unlocked shared-Agent threads (reference src/main.cpp:196-206) at
sync granularity.

``comm`` is what carries the exchange:
  * rl_markets_amd.comm.RcclComm -- the product: one RCCL all-reduce over xGMI,
    in place on the engine's delta buffer, on the engine's stream
    (lob_theta_allreduce, include/lob_comm.h);
  * TorchComm -- torch.distributed on a tensor view of the backend's delta
    (the gloo CPU tests, where a CPU stand-in plays the engine).
``backend`` is anything with td_step(n) and delta_init() (plus, for TorchComm,
delta_tensor() / after_all_reduce() / delta_apply()).
"""


def shard_books(total_books, world_size, rank):
    """Contiguous shard [first, first + n) of `total_books` for `rank`."""
    base, extra = divmod(total_books, world_size)
    n = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, n


class TorchComm:
    """Exchange through torch.distributed (any backend) on the backend's delta tensor."""

    def __init__(self, dist):
        self.dist = dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()

    def sync_weights(self, backend):
        t = backend.delta_tensor()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        backend.after_all_reduce()
        backend.delta_apply()

    def barrier(self):
        self.dist.barrier()


class ShardedLearner:
    def __init__(self, backend, comm=None, sync_every=64):
        """`comm` None = one shard, no exchange.  A one-rank communicator keeps the exchange on (a
        1-GPU box can then run the whole multi-GPU code path: delta kernels, RCCL all-reduce)."""
        self.backend = backend
        self.comm = comm
        self.sync_every = int(sync_every)
        self.steps = 0
        self.n_syncs = 0
        if self.comm is not None:
            backend.delta_init()

    def sync_weights(self):
        self.comm.sync_weights(self.backend)
        self.n_syncs += 1

    def run(self, n_steps):
        done = 0
        while done < n_steps:
            chunk = n_steps - done
            if self.comm is not None:
                chunk = min(chunk, self.sync_every - self.steps % self.sync_every)
            self.backend.td_step(chunk)
            done += chunk
            self.steps += chunk
            if self.comm is not None and self.steps % self.sync_every == 0:
                self.sync_weights()


class EngineBackend:
    """The HIP engine as a ShardedLearner backend."""

    def __init__(self, eng):
        self.eng = eng

    def td_step(self, n):
        self.eng.td_step(n)

    def delta_init(self):
        self.eng.delta_init()

"""Multi-GPU sharding of the batched learner.

Books are independent given theta, so each rank owns a contiguous shard of
books (its own event streams, traces and theta replica).  The only exchange is
the shared weight vector: every ``sync_every`` steps each rank forms
delta = theta - theta_sync, the deltas are summed over ranks and every rank
sets theta = theta_sync + sum(delta).  This is the batched analogue of the
reference's unlocked shared-Agent threads (reference src/main.cpp:196-206) at
sync granularity.

``comm`` is what carries the exchange: rl_markets_amd.comm.RcclComm -- RCCL over
xGMI on the engine's own buffers and stream (lob_theta_allreduce,
include/lob_comm.h: the ranks' written-weights maps all-gathered, the packed
deltas of their union all-reduced; dense all-reduce of the whole vector where
there is no such map) -- or anything else with sync_weights(backend) and
barrier() (the CPU tests drive the same schedule over torch.distributed + gloo
with a stand-in they keep under tests/).  ``backend`` is anything with
td_step(n) and delta_init().
"""


def shard_books(total_books, world_size, rank):
    """Contiguous shard [first, first + n) of `total_books` for `rank`."""
    base, extra = divmod(total_books, world_size)
    n = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, n


class ShardedLearner:
    def __init__(self, backend, comm=None, sync_every=64):
        """`comm` None = one shard, no exchange.  A one-rank communicator keeps the exchange on (a
        1-GPU box can then run the whole multi-GPU code path: delta kernels, RCCL all-reduce)."""
        self.backend = backend
        self.comm = comm
        self.sync_every = int(sync_every)
        self.steps = 0
        self.n_syncs = 0
        # half steps (the exchange inside the sync step): asked of the backend explicitly -- a LOB_ESTATE from td_step_begin has
        # other causes (no reset, a half step left open) and must surface as the error it is
        q = getattr(backend, "td_split_supported", None)
        self._split_ok = hasattr(backend, "td_step_begin") and (q() if q is not None else True)
        if self.comm is not None:
            backend.delta_init()

    def sync_weights(self):
        self.comm.sync_weights(self.backend)
        self.n_syncs += 1

    def run(self, n_steps):
        """n_steps steps of every book of this shard; every `sync_every`-th step carries the exchange INSIDE it, between
        its first half (action selection + performAction) and its second (traces, TD errors, update): there no hit list is
        live -- the step's action has consumed the previous step's lists, its learn kernel builds the next under the
        exchanged weights and maps -- so the exchange voids nothing (lob_td_step_begin / lob_td_step_end)."""
        done = 0
        split = self.comm is not None and hasattr(self.backend, "td_step_begin")
        while done < n_steps:
            chunk = n_steps - done
            sync_now = False
            if self.comm is not None:
                until = self.sync_every - self.steps % self.sync_every
                chunk = min(chunk, until)
                sync_now = chunk == until
            if sync_now and split and self._split_ok:
                if chunk > 1:
                    self.backend.td_step(chunk - 1)
                self.backend.td_step_begin()
                self.sync_weights()
                self.backend.td_step_end()
            else:
                # (no half steps with this backend / engine configuration: the whole step followed by the exchange, as before the
                # split existed -- correct too, the exchange then voids the cached action-selection data of one step)
                self.backend.td_step(chunk)
                if sync_now:
                    self.sync_weights()
            done += chunk
            self.steps += chunk


class EngineBackend:
    """The HIP engine as a ShardedLearner backend."""

    def __init__(self, eng):
        self.eng = eng

    def td_step(self, n):
        self.eng.td_step(n)

    def td_step_begin(self):
        self.eng.td_step_begin()

    def td_step_end(self):
        self.eng.td_step_end()

    def td_split_supported(self):
        return self.eng.td_split_supported()

    def delta_init(self):
        self.eng.delta_init()

"""TEST INFRASTRUCTURE: the multi-GPU weight exchange schedule over torch.distributed (gloo) with a CPU stand-in playing
the engine -- what rl_markets_amd.comm.RcclComm / liblob_comm.so do with RCCL on the GPUs (include/lob_comm.h
lob_theta_allreduce), restated in numpy so that the N > 1 control flow runs without GPUs.  Two exchanges:
  TorchComm        dense: all-reduce(SUM) of theta - theta_sync, memory_size doubles;
  SparseTorchComm  the product's default for shared theta: the ranks' "weight touched" maps all-gathered, their union gives
                   one compact layout common to all ranks, the packed deltas all-reduced, the sum scattered back.
`backend` needs delta_tensor() (a torch view of theta - theta_sync), delta_apply(); the sparse one also delta_mask()."""
import numpy as np


class TorchComm:
    def __init__(self, dist):
        self.dist = dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.bytes = 0

    def sync_weights(self, backend):
        t = backend.delta_tensor()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        self.bytes += t.numel() * 8
        backend.delta_apply()

    def barrier(self):
        self.dist.barrier()


class SparseTorchComm(TorchComm):
    def sync_weights(self, backend):
        import torch
        mask = np.packbits(backend.delta_mask(), bitorder="little")            # the rank's written-weights map, one bit per weight
        mine = torch.from_numpy(mask.copy())
        gathered = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(gathered, mine)                                    # ncclAllGather of the maps
        union = np.zeros_like(mask)
        for g in gathered:
            union |= g.numpy()                                                   # sparse_union_kernel
        idx = np.flatnonzero(np.unpackbits(union, bitorder="little")[:backend.delta_mask().size])
        delta = backend.delta_tensor()
        packed = delta[torch.from_numpy(idx)].clone()                            # sparse_pack_kernel: index order = layout
        self.dist.all_reduce(packed, op=self.dist.ReduceOp.SUM)                  # ncclAllReduce of |union| doubles
        delta.zero_()
        delta[torch.from_numpy(idx)] = packed                                    # sparse_apply_kernel
        self.bytes += packed.numel() * 8 + mask.size * self.world
        backend.delta_apply()

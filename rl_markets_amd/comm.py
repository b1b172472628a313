"""ctypes handle over include/lob_comm.h (liblob_comm.so): the RCCL communicator
of one rank and the in-place all-reduce of the engine's delta-theta buffer over
xGMI.  Plumbing only -- the collective runs on the engine's HIP stream."""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liblob_comm.so")

SUM, MAX = 0, 1
ID_BYTES = 128

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    abi.load()  # liblob_comm.so depends on liblob_engine.so (same directory, $ORIGIN rpath)
    if not os.path.exists(LIB_PATH):
        raise abi.EngineLibraryMissing("%s not found: run __graft_entry__.build()" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, P = C.c_void_p, C.POINTER
    sigs = {
        "lob_comm_get_id": (C.c_int, [vp]),
        "lob_comm_create": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, P(vp)]),
        "lob_comm_create_file": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P(vp)]),
        "lob_comm_destroy": (None, [vp]),
        "lob_comm_rank": (C.c_int32, [vp]),
        "lob_comm_world": (C.c_int32, [vp]),
        "lob_comm_allreduce_f64": (C.c_int, [vp, vp, C.c_int64, vp]),
        "lob_comm_reduce_host_f64": (C.c_int, [vp, vp, C.c_int32, C.c_int32]),
        "lob_comm_barrier": (C.c_int, [vp]),
        "lob_theta_allreduce": (C.c_int, [vp, vp]),
        "lob_comm_allgather_u32": (C.c_int, [vp, vp, vp, C.c_int64, vp]),
        "lob_comm_pin_host_thread": (C.c_int, [C.c_int32, P(C.c_int32)]),
        "lob_comm_exchange_stats": (C.c_int, [vp, P(C.c_double)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib._declared = sorted(sigs)
    _lib = lib
    return lib


class CommError(RuntimeError):
    pass


class RcclComm:
    """One rank of the weight-exchange communicator (one process per GPU)."""

    def __init__(self, rendezvous_path, rank, world, device, timeout_s=300):
        self.lib = load()
        self.rank, self.world = int(rank), int(world)
        h = C.c_void_p()
        self._check(self.lib.lob_comm_create_file(rendezvous_path.encode(), self.rank, self.world, int(device), int(timeout_s), C.byref(h)))
        self.h = h

    def _check(self, rc):
        if rc != abi.LOB_OK:
            raise CommError("lob_comm error %d: %s" % (rc, abi.load().lob_last_error().decode()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.lob_comm_destroy(self.h)
            self.h = None

    # ---- rl_markets_amd.parallel.ShardedLearner interface ----
    def sync_weights(self, backend):
        """delta = theta - theta_sync -> RCCL all-reduce(SUM) in place -> theta = theta_sync + sum."""
        self._check(self.lib.lob_theta_allreduce(backend.eng.h, self.h))

    def reduce(self, values, op=SUM):
        """Small host-side reduction over ranks (timings, counters): list of floats in, list out."""
        n = len(values)
        buf = (C.c_double * n)(*[float(v) for v in values])
        self._check(self.lib.lob_comm_reduce_host_f64(self.h, buf, n, op))
        return list(buf)

    def barrier(self):
        self._check(self.lib.lob_comm_barrier(self.h))

    def exchange_stats(self):
        """Cost of the exchanges since the last call (HIP events on the engine's stream)."""
        out = (C.c_double * 7)()
        self._check(self.lib.lob_comm_exchange_stats(self.h, out))
        n = max(out[0], 1.0)
        return {"exchanges": int(out[0]), "pack_ms": out[1] / n, "collectives_ms": out[2] / n, "apply_ms": out[3] / n,
                "bytes_per_exchange": out[4] / n, "sparse": int(out[5]), "rccl_world": int(out[6])}


def pin_host_thread(device):
    """Bind this process's main thread to the cores next to GPU `device`; returns how many (0: left alone)."""
    n = C.c_int32(0)
    load().lob_comm_pin_host_thread(int(device), C.byref(n))
    return n.value

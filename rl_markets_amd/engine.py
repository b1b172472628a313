"""Thin Python handle over the C ABI (include/lob_engine.h).

Only plumbing lives here (buffer allocation, error mapping); every
computation happens in liblob_engine.so on the GPU.  The class mirrors the
reference's call surface for the hot path:

    reset()            environment::Base::Initialise + Runner prologue
    step(actions)      environment::Base::performAction
    get_state()        environment::Base::getState
    get_reward()       environment::Base::getReward
    td_step(n)         experiment::serial::Learner::_step  x n
    eval_step(n)       experiment::serial::Backtester::_step x n
"""
import ctypes as C

import numpy as np

from . import abi


class LobError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lob_engine error %d: %s" % (code, msg))
        self.code = code


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def default_params():
    lib = abi.load()
    p = abi.Params()
    lib.lob_default_params(C.byref(p))
    return p


def default_gen_params():
    lib = abi.load()
    g = abi.GenParams()
    lib.lob_default_gen_params(C.byref(g))
    return g


def record_words(depth, max_trades):
    return abi.load().lob_record_words(depth, max_trades)


def gen_stream_host(gen, depth, max_trades, first_book, n_books):
    lib = abi.load()
    W = lib.lob_record_words(depth, max_trades)
    out = np.zeros((n_books, gen.n_events, W), dtype=np.uint32)
    rc = lib.lob_gen_stream_host(C.byref(gen), depth, max_trades, C.c_uint64(first_book), n_books, _ptr(out))
    if rc:
        raise LobError(rc, lib.lob_last_error().decode())
    return out


def _take_records(lib, ptr, n, depth, max_trades):
    W = lib.lob_record_words(depth, max_trades)
    buf = (C.c_uint32 * (n * W)).from_address(ptr.value)
    out = np.frombuffer(buf, dtype=np.uint32).reshape(1, n, W).copy()
    lib.lob_free(ptr)
    return out


def convert_csv(md_path, tas_path, max_trades=2):
    """The reference's CSV pair (5-level depth + time-and-sales) -> records[1][n][W]."""
    lib = abi.load()
    ptr, n = C.c_void_p(), C.c_int32()
    rc = lib.lob_convert_csv(md_path.encode(), tas_path.encode(), max_trades, C.byref(ptr), C.byref(n))
    if rc:
        raise LobError(rc, lib.lob_last_error().decode())
    return _take_records(lib, ptr, n.value, 5, max_trades)


def convert_lobster(orderbook_path, message_path, levels_in_file, depth, max_trades=4):
    """LOBSTER orderbook + message files -> records[1][n][W] (one record per millisecond)."""
    lib = abi.load()
    ptr, n = C.c_void_p(), C.c_int32()
    rc = lib.lob_convert_lobster(orderbook_path.encode(), message_path.encode(), levels_in_file, depth, max_trades,
                                 C.byref(ptr), C.byref(n))
    if rc:
        raise LobError(rc, lib.lob_last_error().decode())
    return _take_records(lib, ptr, n.value, depth, max_trades)


class Engine:
    def __init__(self, params, n_books, device=0):
        self.lib = abi.load()
        self.params = params
        self.B = int(n_books)
        self.V = params.n_vars
        self.M = params.memory_size
        h = C.c_void_p()
        self._check(self.lib.lob_create(C.byref(params), self.B, device, C.byref(h)))
        self.h = h

    def _check(self, rc):
        if rc != abi.LOB_OK:
            raise LobError(rc, self.lib.lob_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.lob_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- streams ----
    def load_events(self, records):
        rec = np.ascontiguousarray(records, dtype=np.uint32)
        assert rec.shape[0] == self.B
        self._check(self.lib.lob_load_events(self.h, _ptr(rec), rec.shape[1]))

    def load_events_shared(self, records, phase, n_events):
        """One recorded stream [n_total][W] replayed by every book from its own phase."""
        rec = np.ascontiguousarray(records, dtype=np.uint32)
        assert rec.ndim == 2
        ph = np.ascontiguousarray(phase, dtype=np.int64)
        assert ph.shape == (self.B,)
        self._check(self.lib.lob_load_events_shared(self.h, _ptr(rec), rec.shape[0], _ptr(ph), n_events))

    def stage_events(self, records):
        """lob_stage_events: the next episode's streams, handed over while this one runs; the next reset() adopts them."""
        rec = np.ascontiguousarray(records, dtype=np.uint32)
        assert rec.shape[0] == self.B
        self._staged = rec          # (the caller's buffer must outlive the hand-over)
        self._check(self.lib.lob_stage_events(self.h, _ptr(rec), rec.shape[1]))

    def stage_wait(self):
        self._check(self.lib.lob_stage_wait(self.h))

    def gen_events(self, gen):
        self._check(self.lib.lob_gen_events_device(self.h, C.byref(gen)))

    # ---- environment ----
    def reset(self):
        self._check(self.lib.lob_reset(self.h))

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        assert a.shape == (self.B,)
        self._check(self.lib.lob_step(self.h, _ptr(a)))

    def get_state(self):
        out = np.zeros((self.B, self.V), np.float32)
        self._check(self.lib.lob_get_state(self.h, _ptr(out)))
        return out

    def get_reward(self):
        out = np.zeros(self.B, np.float64)
        self._check(self.lib.lob_get_reward(self.h, _ptr(out)))
        return out

    def get_terminal(self):
        out = np.zeros(self.B, np.uint8)
        self._check(self.lib.lob_get_terminal(self.h, _ptr(out)))
        return out

    def clear_inventory(self):
        self._check(self.lib.lob_clear_inventory(self.h))

    def get_books(self, first=0, n=None):
        n = self.B - first if n is None else n
        out = (abi.BookDump * n)()
        self._check(self.lib.lob_get_books(self.h, first, n, C.cast(out, C.c_void_p)))
        return out

    # ---- learner ----
    def td_step(self, n=1):
        self._check(self.lib.lob_td_step(self.h, n))

    def td_step_begin(self):
        """First half of one learner step (action selection + performAction); a weight exchange fits before td_step_end."""
        self._check(self.lib.lob_td_step_begin(self.h))

    def td_step_end(self):
        self._check(self.lib.lob_td_step_end(self.h))

    def model_log_enable(self, on=True):
        """lob_model_log_enable: mean |delta| rows as the reference's `model_log` logger writes them (src/rl/agent.cpp:93-100)."""
        self._check(self.lib.lob_model_log_enable(self.h, 1 if on else 0))

    def model_log_read(self, cap=8192):
        rows = np.zeros(cap, np.float64)
        n, lost = C.c_int32(0), C.c_int64(0)
        self._check(self.lib.lob_model_log_read(self.h, _ptr(rows), cap, C.byref(n), C.byref(lost)))
        return rows[:n.value].copy(), int(lost.value)

    def td_split_supported(self):
        """Whether td_step_begin / td_step_end are available with this engine configuration (lob_td_split_supported)."""
        return bool(self.lib.lob_td_split_supported(self.h))

    def eval_step(self, n=1):
        self._check(self.lib.lob_eval_step(self.h, n))

    def handle_terminal(self):
        self._check(self.lib.lob_handle_terminal(self.h))

    def set_alpha(self, a):
        self._check(self.lib.lob_set_alpha(self.h, a))

    def set_epsilon(self, e):
        self._check(self.lib.lob_set_epsilon(self.h, e))

    def set_tau(self, t):
        self._check(self.lib.lob_set_tau(self.h, t))

    def features(self, vars_):
        v = np.ascontiguousarray(vars_, dtype=np.float32).reshape(-1, self.V)
        out = np.zeros((v.shape[0], 9, 96), np.int32)
        self._check(self.lib.lob_features(self.h, _ptr(v), v.shape[0], _ptr(out)))
        return out

    def q_values(self, vars_):
        v = np.ascontiguousarray(vars_, dtype=np.float32).reshape(-1, self.V)
        out = np.zeros((v.shape[0], 9), np.float64)
        self._check(self.lib.lob_q_values(self.h, _ptr(v), v.shape[0], _ptr(out)))
        return out

    def theta(self, which=0):
        out = np.zeros(self.M, np.float64)
        self._check(self.lib.lob_theta_get(self.h, which, _ptr(out), self.M))
        return out

    def set_theta(self, values, which=0):
        v = np.ascontiguousarray(values, dtype=np.float64)
        assert v.shape == (self.M,)
        self._check(self.lib.lob_theta_set(self.h, which, _ptr(v), self.M))

    def _vec(self, fn, dtype):
        out = np.zeros(self.B, dtype)
        self._check(fn(self.h, _ptr(out)))
        return out

    def last_actions(self):
        return self._vec(self.lib.lob_get_last_actions, np.int32)

    def last_td(self):
        return self._vec(self.lib.lob_get_last_td, np.float64)

    def last_rewards(self):
        return self._vec(self.lib.lob_get_last_rewards, np.float64)

    def stepped(self):
        return self._vec(self.lib.lob_get_stepped, np.int32)

    def rng_counters(self):
        return self._vec(self.lib.lob_get_rng_counters, np.uint64)

    def learner_state(self):
        out = np.zeros((self.B, self.V), np.float32)
        self._check(self.lib.lob_get_learner_state(self.h, _ptr(out)))
        return out

    def traces(self, book):
        cap = 64 * 32  # LOB_TRACE_GENS generations of 32 tiles
        idx = np.zeros(cap, np.int32)
        e = np.zeros(cap, np.float32)
        n = C.c_int32()
        self._check(self.lib.lob_get_traces(self.h, book, _ptr(idx), _ptr(e), cap, C.byref(n)))
        return idx[:n.value].copy(), e[:n.value].copy()

    def counters(self):
        c = np.zeros(4, np.int64)
        self._check(self.lib.lob_get_counters(self.h, _ptr(c)))
        return c

    def path_stats(self):
        """lob_get_path_stats: which kernels served the books (diagnostics of the fast paths)."""
        c = np.zeros(8, np.int64)
        self._check(self.lib.lob_get_path_stats(self.h, _ptr(c)))
        return c

    def flow_stats(self):
        """lob_debug_flow (a diagnostic export, not in include/lob_engine.h): learner steps by the shape of their combined update."""
        c = np.zeros(8, np.int64)
        fn = self.lib.lob_debug_flow
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
        self._check(fn(self.h, _ptr(c)))
        return {"added_in_place": int(c[0]), "rest_on_side_stream": int(c[1]), "block_sums": int(c[2]), "every_book": int(c[3]),
                "act_inline_general": int(c[4]), "act_work_list_dense": int(c[5]), "act_work_list_other": int(c[6]),
                "dense_sums": int(c[7])}

    def deferred_generations(self):
        """lob_debug_deferred (a diagnostic export): generations without a combine slot that trace_rest_kernel left to apply_kernel so far."""
        c = np.zeros(1, np.int64)
        fn = self.lib.lob_debug_deferred
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
        self._check(fn(self.h, _ptr(c)))
        return int(c[0])

    def fastpath_stats(self):
        """lob_debug_fastpath (a diagnostic export, not in include/lob_engine.h): written weights and the live books' hit-list lengths."""
        n = 4 + 257
        c = np.zeros(n, np.int64)
        fn = self.lib.lob_debug_fastpath
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]
        self._check(fn(self.h, _ptr(c), n))
        hist = c[4:]
        with_list = int(hist.sum())
        cum = np.cumsum(hist)
        def pct(q):
            return int(np.searchsorted(cum, q * with_list)) if with_list else None
        return {"written_weights": int(c[0]), "live_books": int(c[1]), "books_without_list": int(c[2]),
                "list_len_mean": round(float(c[3]) / with_list, 2) if with_list else None,
                "list_len_p50": pct(0.5), "list_len_p99": pct(0.99), "list_len_max": int(np.nonzero(hist)[0].max()) if with_list else None,
                "hist": hist}

    # ---- multi-GPU weight exchange ----
    def delta_init(self):
        self._check(self.lib.lob_delta_init(self.h))

    def delta_begin(self):
        p = C.c_void_p()
        n = C.c_int64()
        self._check(self.lib.lob_delta_begin(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def delta_apply(self):
        self._check(self.lib.lob_delta_apply(self.h))

    # sparse exchange (include/lob_engine.h lob_delta_sparse_*): device pointers as ints
    def delta_sparse_supported(self):
        return bool(self.lib.lob_delta_sparse_supported(self.h))

    def delta_sparse_maps(self, world):
        own, gather, words = C.c_void_p(), C.c_void_p(), C.c_int64()
        self._check(self.lib.lob_delta_sparse_maps(self.h, int(world), C.byref(own), C.byref(gather), C.byref(words)))
        return own.value, gather.value, words.value

    def delta_sparse_pack(self, world):
        p, n = C.c_void_p(), C.c_int64()
        self._check(self.lib.lob_delta_sparse_pack(self.h, int(world), C.byref(p), C.byref(n)))
        return p.value, n.value

    def delta_sparse_apply(self):
        self._check(self.lib.lob_delta_sparse_apply(self.h))

    def exchange_debug(self):
        """lob_debug_exchange (a diagnostic export): sparse exchanges without / with a host synchronisation, unions that outgrew the fixed count."""
        c = np.zeros(5, np.int64)
        fn = self.lib.lob_debug_exchange
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
        self._check(fn(self.h, _ptr(c)))
        return {"without_host_sync": int(c[0]), "with_host_sync": int(c[1]), "unions_beyond_the_fixed_count": int(c[2]), "fixed_count": int(c[3]),
                "last_union": int(c[4])}

    def sync(self):
        self._check(self.lib.lob_sync(self.h))

    def kernel_timing(self, enable=True):
        """False / 0: off; True / 1: every launch; n > 1: the launches of every n-th step."""
        self._check(self.lib.lob_kernel_timing(self.h, int(enable)))

    def kernel_time_ms(self, name):
        ms = C.c_double()
        n = C.c_int64()
        self._check(self.lib.lob_kernel_time_ms(self.h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

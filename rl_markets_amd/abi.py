"""ctypes mirror of include/lob_engine.h (the C ABI of liblob_engine.so).

This module only declares types and loads the shared library; it contains no
algorithm.  The library is built in-tree by ``__graft_entry__.build()``
(hipcc, gfx950) and there is NO Python / CPU fallback: if the library is
missing, importing the engine raises.
"""
import ctypes as C
import os

LOB_N_ACTIONS = 9
LOB_N_TILINGS = 32
LOB_MAX_DEPTH = 10
LOB_MAX_TRADES = 8
LOB_MAX_BANDS = 20
LOB_MAX_VARS = 13

LOB_OK, LOB_EINVAL, LOB_ENODEV, LOB_ENOMEM, LOB_ESTATE, LOB_EDATA, LOB_EHIP = 0, -1, -2, -3, -4, -5, -6

(VAR_POS, VAR_SPD, VAR_MPM, VAR_IMB, VAR_SVL, VAR_VOL, VAR_RSI, VAR_VWAP, VAR_A_DIST, VAR_A_QUEUE,
 VAR_B_DIST, VAR_B_QUEUE, VAR_LAST_ACTION) = range(13)
VAR_NAMES = {"pos": VAR_POS, "spd": VAR_SPD, "mpm": VAR_MPM, "imb": VAR_IMB, "svl": VAR_SVL, "vol": VAR_VOL,
             "rsi": VAR_RSI, "vwap": VAR_VWAP, "a_dist": VAR_A_DIST, "a_queue": VAR_A_QUEUE,
             "b_dist": VAR_B_DIST, "b_queue": VAR_B_QUEUE, "last_action": VAR_LAST_ACTION}
(REWARD_NONE, REWARD_PNL, REWARD_PNL_DAMPED, REWARD_SPREAD, REWARD_NORMED, REWARD_LOVOL, REWARD_MM_LINEAR,
 REWARD_MM_EXP, REWARD_MM_DIV) = range(9)
REWARD_NAMES = {"none": 0, "pnl": 1, "pnl_damped": 2, "spread": 3, "normed": 4, "lovol": 5, "mm_linear": 6,
                "mm_exp": 7, "mm_div": 8}
EVT_FLAG_SAME_TIME, EVT_FLAG_TAS_DRY = 1, 2   # record word [1]
TP_MIDPRICE, TP_MICROPRICE = 0, 1
QUOTE_TARGET, QUOTE_BOOK = 0, 1
ALGO_SARSA, ALGO_QLAMBDA, ALGO_DOUBLE_Q, ALGO_R_LEARN, ALGO_ONLINE_R_LEARN, ALGO_DOUBLE_R_LEARN = 0, 1, 2, 3, 4, 5
THETA_SHARED, THETA_PRIVATE = 0, 1
POLICY_EPS_GREEDY, POLICY_BOLTZMANN = 0, 1


class _Strict(C.Structure):
    """ctypes.Structure that refuses attributes the C struct does not have (a misspelt field
    would otherwise be accepted silently and leave the real one at its default)."""

    def __setattr__(self, name, value):
        if not any(name == f[0] for f in self._fields_):
            raise AttributeError("%s has no field %r" % (type(self).__name__, name))
        super().__setattr__(name, value)


class Market(_Strict):
    _fields_ = [("open_ms", C.c_int64), ("close_ms", C.c_int64), ("n_bands", C.c_int32), ("_pad", C.c_int32),
                ("band_lb", C.c_double * LOB_MAX_BANDS), ("band_tick", C.c_double * LOB_MAX_BANDS)]


class Params(_Strict):
    _fields_ = [
        ("abi_version", C.c_int32), ("depth", C.c_int32), ("max_trades", C.c_int32), ("n_vars", C.c_int32),
        ("vars", C.c_int32 * LOB_MAX_VARS),
        ("market", Market),
        ("order_size", C.c_int32), ("reward_measure", C.c_int32), ("pos_lb", C.c_int64), ("pos_ub", C.c_int64),
        ("damping_factor", C.c_float), ("pos_weight", C.c_float), ("trd_weight", C.c_float),
        ("pnl_weight", C.c_float),
        ("lb_mpm", C.c_int32), ("lb_vlt", C.c_int32), ("lb_svl", C.c_int32), ("lb_vwap", C.c_int32),
        ("lb_rsi", C.c_int32), ("lb_spread", C.c_int32), ("lb_pnl", C.c_int32), ("lb_target", C.c_int32),
        ("target_price", C.c_int32), ("quote_mode", C.c_int32),
        ("memory_size", C.c_int64), ("n_tilings", C.c_int32), ("n_actions", C.c_int32),
        ("group_weights", C.c_double * 3), ("gamma", C.c_double), ("lambda_", C.c_double),
        ("alpha", C.c_double), ("epsilon", C.c_double),
        ("algo", C.c_int32), ("theta_mode", C.c_int32), ("seed", C.c_uint64), ("book_id_offset", C.c_uint64),
        ("policy", C.c_int32), ("random_init", C.c_int32), ("tau", C.c_double), ("beta", C.c_double),
    ]


class GenParams(_Strict):
    _fields_ = [("seed", C.c_uint64)] + [(n, C.c_int32) for n in (
        "n_events", "t0_ms", "dt_ms", "start_ticks", "min_ticks", "max_ticks", "move_prob_q16",
        "spread2_prob_q16", "trade_prob_q16", "trade2_prob_q16", "touch_prob_q16", "vol_min", "vol_max",
        "trade_min", "trade_max")]


class BookDump(C.Structure):
    _fields_ = [
        ("ask_px", C.c_double * LOB_MAX_DEPTH), ("bid_px", C.c_double * LOB_MAX_DEPTH),
        ("ask_last_px", C.c_double * LOB_MAX_DEPTH), ("bid_last_px", C.c_double * LOB_MAX_DEPTH),
        ("ask_vol", C.c_int64 * LOB_MAX_DEPTH), ("bid_vol", C.c_int64 * LOB_MAX_DEPTH),
        ("ask_last_vol", C.c_int64 * LOB_MAX_DEPTH), ("bid_last_vol", C.c_int64 * LOB_MAX_DEPTH),
        ("ask_total_volume", C.c_int64), ("bid_total_volume", C.c_int64),
        ("ask_last_total_volume", C.c_int64), ("bid_last_total_volume", C.c_int64),
        ("ask_n_transacted", C.c_int32), ("bid_n_transacted", C.c_int32),
        ("ask_has_order", C.c_int32), ("bid_has_order", C.c_int32),
        ("ask_order_px", C.c_double), ("bid_order_px", C.c_double),
        ("ask_order_rem", C.c_int64), ("bid_order_rem", C.c_int64),
        ("ask_q_head", C.c_int64), ("bid_q_head", C.c_int64), ("ask_q_tail", C.c_int64), ("bid_q_tail", C.c_int64),
        ("position", C.c_int64),
        ("ask_quote", C.c_double), ("bid_quote", C.c_double),
        ("ask_level", C.c_int32), ("bid_level", C.c_int32),
        ("pnl_step", C.c_double), ("momentum_pnl_step", C.c_double),
        ("lo_vol_step", C.c_int32), ("last_action", C.c_int32),
        ("episode_reward", C.c_double), ("episode_pnl", C.c_double), ("episode_bandh", C.c_double),
        ("spread_mean", C.c_double), ("target_price", C.c_double),
        ("time_ms", C.c_int64),
        ("cursor", C.c_int32), ("terminal", C.c_int32), ("total_ticks", C.c_int32), ("n_traces", C.c_int32),
        ("market_buys", C.c_int32), ("market_sells", C.c_int32), ("ticks_with_ask", C.c_int32), ("ticks_with_bid", C.c_int32),
        ("ticks_with_both", C.c_int32), ("ticks_with_position", C.c_int32), ("ticks_long", C.c_int32), ("ticks_short", C.c_int32),
        ("ask_transactions", C.c_int32), ("bid_transactions", C.c_int32),
    ]


_HERE = os.path.dirname(os.path.abspath(__file__))
# LOB_ENGINE_LIB: an experiment build of the same library (tools/exp_prof.py, tools/exp_variants.sh)
LIB_PATH = os.environ.get("LOB_ENGINE_LIB") or os.path.join(_HERE, "csrc", "liblob_engine.so")


class EngineLibraryMissing(RuntimeError):
    pass


_lib = None


def load():
    """Load liblob_engine.so (built by __graft_entry__.build()); raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineLibraryMissing(
            "%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    # NB: PyTorch-ROCm bundles its own libamdhip64.so.7.  Exactly one HIP runtime may live
    # in a process: a process that needs torch as well must `import torch` BEFORE this call so
    # that the library below resolves to the runtime torch loaded.  (Nothing in the product does:
    # the multi-GPU exchange is liblob_comm.so / RCCL, rl_markets_amd/comm.py.)
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    vp = C.c_void_p
    sigs = {
        "lob_abi_version": (C.c_int, []),
        "lob_last_error": (C.c_char_p, []),
        "lob_default_params": (None, [P(Params)]),
        "lob_market_preset": (C.c_int, [C.c_char_p, P(Market)]),
        "lob_to_ticks": (C.c_int, [P(Market), C.c_double, P(C.c_int32)]),
        "lob_to_price": (C.c_int, [P(Market), C.c_int32, P(C.c_double)]),
        "lob_tick_size": (C.c_int, [P(Market), C.c_double, P(C.c_double)]),
        "lob_record_words": (C.c_int32, [C.c_int32, C.c_int32]),
        "lob_default_gen_params": (None, [P(GenParams)]),
        "lob_gen_stream_host": (C.c_int, [P(GenParams), C.c_int32, C.c_int32, C.c_uint64, C.c_int32, vp]),
        "lob_validate_stream": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
        "lob_convert_csv": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int32, P(vp), P(C.c_int32)]),
        "lob_convert_lobster": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, P(vp), P(C.c_int32)]),
        "lob_free": (None, [vp]),
        "lob_create": (C.c_int, [P(Params), C.c_int32, C.c_int32, P(vp)]),
        "lob_destroy": (None, [vp]),
        "lob_load_events": (C.c_int, [vp, vp, C.c_int32]),
        "lob_load_events_shared": (C.c_int, [vp, vp, C.c_int64, vp, C.c_int32]),
        "lob_gen_events_device": (C.c_int, [vp, P(GenParams)]),
        "lob_reset": (C.c_int, [vp]),
        "lob_step": (C.c_int, [vp, vp]),
        "lob_get_state": (C.c_int, [vp, vp]),
        "lob_get_reward": (C.c_int, [vp, vp]),
        "lob_get_terminal": (C.c_int, [vp, vp]),
        "lob_clear_inventory": (C.c_int, [vp]),
        "lob_get_book": (C.c_int, [vp, C.c_int32, P(BookDump)]),
        "lob_get_books": (C.c_int, [vp, C.c_int32, C.c_int32, vp]),
        "lob_td_step": (C.c_int, [vp, C.c_int32]),
        "lob_td_step_begin": (C.c_int, [vp]),
        "lob_td_split_supported": (C.c_int, [vp]),
        "lob_model_log_enable": (C.c_int, [vp, C.c_int32]),
        "lob_model_log_read": (C.c_int, [vp, vp, C.c_int32, vp, vp]),
        "lob_td_step_end": (C.c_int, [vp]),
        "lob_eval_step": (C.c_int, [vp, C.c_int32]),
        "lob_handle_terminal": (C.c_int, [vp]),
        "lob_set_alpha": (C.c_int, [vp, C.c_double]),
        "lob_set_epsilon": (C.c_int, [vp, C.c_double]),
        "lob_set_tau": (C.c_int, [vp, C.c_double]),
        "lob_features": (C.c_int, [vp, vp, C.c_int32, vp]),
        "lob_q_values": (C.c_int, [vp, vp, C.c_int32, vp]),
        "lob_theta_get": (C.c_int, [vp, C.c_int32, vp, C.c_int64]),
        "lob_theta_set": (C.c_int, [vp, C.c_int32, vp, C.c_int64]),
        "lob_get_last_actions": (C.c_int, [vp, vp]),
        "lob_get_last_td": (C.c_int, [vp, vp]),
        "lob_get_last_rewards": (C.c_int, [vp, vp]),
        "lob_get_stepped": (C.c_int, [vp, vp]),
        "lob_get_rng_counters": (C.c_int, [vp, vp]),
        "lob_get_learner_state": (C.c_int, [vp, vp]),
        "lob_get_traces": (C.c_int, [vp, C.c_int32, vp, vp, C.c_int32, P(C.c_int32)]),
        "lob_get_counters": (C.c_int, [vp, vp]),
        "lob_get_path_stats": (C.c_int, [vp, vp]),
        "lob_delta_init": (C.c_int, [vp]),
        "lob_delta_begin": (C.c_int, [vp, P(vp), P(C.c_int64)]),
        "lob_delta_begin_async": (C.c_int, [vp, P(vp), P(C.c_int64)]),
        "lob_delta_apply": (C.c_int, [vp]),
        "lob_stage_events": (C.c_int, [vp, vp, C.c_int32]),
        "lob_stage_wait": (C.c_int, [vp]),
        "lob_delta_sparse_supported": (C.c_int, [vp]),
        "lob_delta_sparse_maps": (C.c_int, [vp, C.c_int32, P(vp), P(vp), P(C.c_int64)]),
        "lob_delta_sparse_pack": (C.c_int, [vp, C.c_int32, P(vp), P(C.c_int64)]),
        "lob_delta_sparse_apply": (C.c_int, [vp]),
        "lob_sync": (C.c_int, [vp]),
        "lob_stream": (vp, [vp]),
        "lob_kernel_time_ms": (C.c_int, [vp, C.c_char_p, P(C.c_double), P(C.c_int64)]),
        "lob_kernel_timing": (C.c_int, [vp, C.c_int32]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    lib._declared = sorted(sigs)
    _lib = lib
    return lib

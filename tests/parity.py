"""Comparison helpers shared by the GPU parity tests and __graft_entry__.smoke().
The oracle (tests/oracle_lib.py) is the checker; the engine is the thing checked."""
import ctypes as C

import numpy as np

from rl_markets_amd import abi
from tests import oracle_lib as ol

# book fields compared bit-exactly (everything except bookkeeping the oracle
# and the engine count differently)
BOOK_SKIP = {"n_traces"}


def dumps_to_np(dumps):
    return np.frombuffer(bytes(dumps), dtype=ol.BOOK_DTYPE)


def assert_books_equal(eng_books, orc_books, tag="", skip=()):
    skip = set(skip) | BOOK_SKIP
    for name in ol.BOOK_DTYPE.names:
        if name in skip:
            continue
        a, b = eng_books[name], orc_books[name]
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            i = tuple(bad[0])
            raise AssertionError("%s: book field %s differs at %s: engine=%r oracle=%r (%d mismatches)"
                                 % (tag, name, i, a[i], b[i], len(bad)))


def compare_env(eng, orc, tag=""):
    eb = dumps_to_np(eng.get_books())
    ob = orc.recs()["book"]
    assert_books_equal(eb, ob, tag)


def compare_learner_step(eng, orc, tag="", exact=True, rtol=0.0, td_floor=0.0):
    recs = orc.recs()
    eb = dumps_to_np(eng.get_books())
    assert_books_equal(eb, recs["book"], tag)
    np.testing.assert_array_equal(eb["n_traces"], recs["book"]["n_traces"], err_msg=tag + " n_traces")
    stepped = eng.stepped().astype(bool)
    np.testing.assert_array_equal(eng.rng_counters(), recs["rng_ctr"], err_msg=tag + " rng counters")
    if stepped.any():
        np.testing.assert_array_equal(eng.last_actions()[stepped], recs["action"][stepped], err_msg=tag + " actions")
        np.testing.assert_array_equal(eng.last_rewards()[stepped], recs["reward"][stepped], err_msg=tag + " rewards")
        ev = eng.learner_state()[stepped]
        ov = recs["vars"][stepped][:, :eng.V]
        np.testing.assert_array_equal(ev, ov, err_msg=tag + " state vars")
        if exact:
            np.testing.assert_array_equal(eng.last_td()[stepped], recs["td"][stepped], err_msg=tag + " td")
        else:
            # shared theta: the weights are sums of f64 atomic additions in whatever order the hardware makes them, and a TD
            # error is a difference of sums of such weights -- the absolute floor scales with their magnitude (alpha = 0.3,
            # 300 greedy books on one table: |td| ~ 40, deviations of 1-4e-9 on EVERY path incl. LOB_NO_COMBINE=1, the
            # trace-by-trace update of round 1; north star: 1e-5 relative)
            want = recs["td"][stepped]
            atol = max(1e-9 * max(1.0, float(np.abs(want).max())), td_floor)  # (td_floor: what the caller has measured the noise to be)
            np.testing.assert_allclose(eng.last_td()[stepped], want, rtol=rtol, atol=atol, err_msg=tag + " td")

"""Known-answer vectors of the reference's own unit tests, replayed against the
oracle restatement (CPU only).  Each case cites the reference test it restates."""
import numpy as np
import pytest

from tests import oracle_lib as ol


def order(qh, size, ops, price=1.0):
    a = np.array(ops, np.int64).reshape(-1, 2)
    out = np.zeros((len(a), 4), np.int64)
    ol.load().oracle_order_script(price, size, qh, ol.ptr(a), len(a), ol.ptr(out))
    return out


TX, CANCEL, BEHIND, CLEAR = 0, 1, 2, 3


# test/test_Order.cpp:74-265 "order handles cancellations"
@pytest.mark.parametrize("qh,behind,vol,want", [
    (0, 0, 50, (0, 0)), (100, 0, 50, (50, 0)), (100, 0, 100, (0, 0)), (100, 0, 150, (0, 0)),
    (0, 100, 50, (0, 50)), (100, 500, 50, (91, 459)), (100, 500, 100, (83, 417)), (100, 500, 600, (0, 0)),
    (100, 500, 1000, (0, 0)), (100, 5000, 50, (99, 4951)), (100, 5000, 100, (98, 4902)), (100, 5000, 10000, (0, 0)),
])
def test_order_cancellations(qh, behind, vol, want):
    ops = ([[BEHIND, behind]] if behind else []) + [[CANCEL, vol]]
    out = order(qh, 100, ops)
    assert (out[-1, 0], out[-1, 1]) == want


# test/test_Order.cpp:267-354 "order handles transactions": (q_head, volume) -> (executed, is_executed, q_head)
@pytest.mark.parametrize("qh,vol,executed,done,qh_after", [
    (0, 50, 50, False, 0), (0, 100, 100, True, 0), (100, 50, 0, False, 50), (100, 100, 0, False, 0),
    (100, 150, 50, False, 0), (100, 200, 100, True, 0),
])
def test_order_transactions(qh, vol, executed, done, qh_after):
    out = order(qh, 100, [[TX, vol]])
    assert 100 - out[0, 2] == executed
    assert (out[0, 2] == 0) == done
    assert out[0, 0] == qh_after


def book(depth, *script):
    s = np.array([x for part in script for x in part], np.float64)
    out = np.zeros(256, np.float64)
    n = ol.load().oracle_book_script(depth, ol.ptr(s), len(s), ol.ptr(out), 256)
    return n, out[:max(n, 0)]


def apply(side, prices, vols, trades=()):
    t = [x for pv in trades for x in pv]
    return [1, side, len(prices)] + list(prices) + list(vols) + [len(trades)] + t


def place(side, price, size):
    return [3, side, price, size]


def txn(side, trades, ref):
    return [4, side, len(trades)] + [x for pv in trades for x in pv] + [ref]


ASK, BID = 0, 1


# test/test_Book.cpp:451-489 "handles agent market orders"
def test_walk_the_book():
    base = apply(ASK, [100.0, 110.0, 120.0], [100, 200, 150])
    n, out = book(3, base, [5, ASK, 90.0, 500])
    assert list(out) == [0, 0, 0]
    for size in (400, -400):
        n, out = book(3, base, [5, ASK, 90.0, size])
        assert out[0] == 400 and out[1] == -(100 * 10.0 + 200 * 20.0 + 100 * 30.0)


# test/test_Book.cpp:491-581: bid book, one level 100 @ 100.0, order 50 @ level 0
@pytest.mark.parametrize("vol,want_vol,want_proxy,qa,rem", [
    (50, 0, 0.0, 50, 50), (100, 0, 0.0, 0, 50), (125, 25, 250.0, 0, 25), (150, 50, 500.0, -1, -1), (1000, 50, 500.0, -1, -1),
])
def test_bid_transactions_at_best(vol, want_vol, want_proxy, qa, rem):
    n, out = book(1, apply(BID, [100.0], [100]), place(BID, 100.0, 50), txn(BID, [(100.0, vol)], 110.0), [6, BID, 100.0])
    assert out[0] == 1  # placed
    assert out[1] == want_vol and out[2] == want_proxy
    assert out[4] == qa and out[6] == rem


def test_bid_two_transactions_in_a_row():  # test_Book.cpp:539-559
    n, out = book(1, apply(BID, [100.0], [100]), place(BID, 100.0, 50), txn(BID, [(100.0, 125)], 110.0),
                  txn(BID, [(100.0, 10)], 105.0), [6, BID, 100.0])
    assert list(out[1:3]) == [25, 250.0] and list(out[4:6]) == [10, 50.0]
    assert out[7] == 0 and out[9] == 15


# test_Book.cpp:583-633: ask book 100@100, 200@110, 150@120; order 100 @ 110.0
@pytest.mark.parametrize("trades,want_vol,want_proxy,qa,rem", [
    ([(110.0, 100)], 0, 0.0, 100, 100), ([(110.0, 300)], -100, 1000.0, -1, -1),
    ([(110.0, 200), (120.0, 50)], -50, 500.0, 0, 50), ([(110.0, 200), (120.0, 100)], -100, 1000.0, -1, -1),
])
def test_ask_transactions_at_level(trades, want_vol, want_proxy, qa, rem):
    n, out = book(3, apply(ASK, [100.0, 110.0, 120.0], [100, 200, 150]), place(ASK, 110.0, 100),
                  txn(ASK, trades, 100.0), [6, ASK, 110.0])
    assert out[1] == want_vol and out[2] == want_proxy
    assert out[4] == qa and out[6] == rem


# test_Book.cpp:635-695: orders priced better than the book
def test_better_priced_orders():
    n, out = book(1, apply(BID, [100.0], [100]), place(BID, 110.0, 50), txn(BID, [(100.0, 10)], 111.0), [6, BID, 110.0])
    assert list(out[1:3]) == [10, 10.0] and out[6] == 40
    n, out = book(1, apply(BID, [100.0], [100]), place(BID, 110.0, 50), txn(BID, [(100.0, 100)], 111.0), [6, BID, 110.0])
    assert list(out[1:3]) == [50, 50.0] and out[6] == -1
    n, out = book(1, apply(ASK, [100.0], [100]), place(ASK, 90.0, 50), txn(ASK, [(100.0, 10)], 85.0), [6, ASK, 90.0])
    assert list(out[1:3]) == [-10, 50.0] and out[6] == 40
    n, out = book(1, apply(ASK, [100.0], [100]), place(ASK, 90.0, 50), txn(ASK, [(100.0, 100)], 85.0), [6, ASK, 90.0])
    assert list(out[1:3]) == [-50, 250.0] and out[6] == -1


# test_Book.cpp:697-765: orders at undefined levels, multiple orders
def test_off_level_and_multiple_orders():
    n, out = book(2, apply(ASK, [100.0, 110.0], [100, 200]), place(ASK, 105.0, 50),
                  txn(ASK, [(100.0, 100), (110.0, 50)], 100.0), [6, ASK, 105.0])
    assert list(out[1:3]) == [-50, 250.0] and out[6] == -1
    n, out = book(2, apply(ASK, [100.0, 110.0], [100, 200]), place(ASK, 105.0, 50), place(ASK, 107.0, 50),
                  txn(ASK, [(100.0, 100), (110.0, 100)], 100.0), [6, ASK, 105.0], [6, ASK, 107.0])
    assert list(out[2:4]) == [-100, 5.0 * 50 + 7.0 * 50]
    assert out[7] == -1 and out[10] == -1


# test_Book.cpp:230-449 "handles agent order placement": queue ahead = displayed volume, or 0 off-level
def test_order_placement_queues():
    n, out = book(1, apply(BID, [100.0], [100]), place(BID, 100.0, 50), [6, BID, 100.0], [10, BID])
    assert list(out[1:4]) == [100, 0, 50] and out[7] == 1
    n, out = book(1, apply(BID, [100.0], [100]), place(BID, 110.0, 50), [6, BID, 110.0])
    assert list(out[1:4]) == [0, 0, 50]
    n, out = book(2, apply(ASK, [100.0, 110.0], [100, 200]), place(ASK, 110.0, 50), [6, ASK, 110.0])
    assert list(out[1:4]) == [200, 0, 50]
    n, out = book(1, [6, ASK, 100.0])
    assert list(out) == [-1, -1, -1]
    # placing twice at the same price is a silent no-op (book.cpp:250-261)
    n, out = book(1, apply(BID, [100.0], [100]), place(BID, 100.0, 50), place(BID, 100.0, 70), [6, BID, 100.0])
    assert out[0] == 1 and out[1] == 0 and out[4] == 50


# test_Book.cpp:11-208: snapshot / stash semantics, bid ordering, throws
def test_snapshot_and_stash():
    n, out = book(2, apply(BID, [100.0, 110.0], [100, 200]), [7, BID, 0], [7, BID, 1], [11, BID, 100.0])
    assert list(out) == [110.0, 200, 100.0, 100, 1]  # best bid = highest price
    n, out = book(2, apply(ASK, [100.0, 110.0], [100, 200]), [2, ASK], apply(ASK, [101.0, 111.0], [10, 20]),
                  [7, ASK, 0], [12, ASK, 0], [12, ASK, -1])
    assert list(out) == [101.0, 10, 100.0, 100, 110.0, 200]
    assert book(1, apply(ASK, [0.0], [100]))[0] == -1      # price <= 0 throws
    assert book(1, apply(ASK, [100.0], [0]))[0] == -1      # volume <= 0 throws
    assert book(1, [7, ASK, 0])[0] == -1                   # undefined price throws


# test_Book.cpp:767-853: 4-decimal price keys
def test_price_key_tolerance():
    n, out = book(1, apply(ASK, [1.1111], [100]), [11, ASK, 1.11114], [11, ASK, 1.11117])
    assert list(out) == [0, -1]
    n, out = book(1, apply(ASK, [1.1111], [100]), place(ASK, 1.11114, 10), [6, ASK, 1.1111])
    assert out[1] == 100  # queued behind the level with the "same" price


# test_Book.cpp:210-228: observed transaction value / volume
def test_observed_volume():
    n, out = book(1, apply(ASK, [100.0], [100]), txn(ASK, [(100.0, 30), (101.0, 20)], 99.0), [10, ASK])
    assert out[3] == 100.0 * 30 + 101.0 * 20 and out[4] == 50


# test/test_Accumulators.cpp:8-22,38-55 (RollingMean<float> there, <double> on the hot path)
def test_rolling_mean():
    vals = np.array([1, 2, 3, 4, 5], np.float64)
    out = np.zeros((5, 5))
    ol.load().oracle_rolling_mean(3, ol.ptr(vals), 5, ol.ptr(out))
    assert out[2, 0] == pytest.approx(2.0) and out[2, 1] == pytest.approx(1.0)
    assert out[3, 0] == pytest.approx(3.0) and out[3, 1] == pytest.approx(1.0)
    assert out[4, 0] == pytest.approx(4.0) and out[4, 1] == pytest.approx(1.0)
    assert list(out[:, 4]) == [0, 0, 1, 1, 1]
    vals = np.array([10.5, 11.5, 11.5, 11.5, 12.5])
    ol.load().oracle_rolling_mean(3, ol.ptr(vals), 5, ol.ptr(out))
    assert out[2, 0] == pytest.approx(11.1667, rel=1e-5) and out[2, 1] == pytest.approx(1 / 3)
    assert out[3, 0] == pytest.approx(11.5) and abs(out[3, 1]) < 1e-9
    assert out[4, 0] == pytest.approx(11.8333, rel=1e-5) and out[4, 1] == pytest.approx(1 / 3)

"""Writers for the data formats the converters ingest (test helpers)."""
import numpy as np


def ms_to_str(t):
    ms = t % 1000
    t //= 1000
    s = t % 60
    t //= 60
    m = t % 60
    t //= 60
    return "%02d:%02d:%02d.%03d" % (t, m, s, ms)


def write_reference_csvs(rec, depth, trades, md_path, tas_path, date=20200102):
    """One book's records -> the reference's 22-column depth CSV and 4-column time-and-sales CSV
    (include/data/basic.h:17-24,49-52).  Trades of record r are stamped 1 ms before its depth row."""
    assert depth == 5
    f32 = rec.view(np.float32)
    with open(md_path, "w") as md, open(tas_path, "w") as ts:
        md.write("date,time,ap1,ap2,ap3,ap4,ap5,av1,av2,av3,av4,av5,bp1,bp2,bp3,bp4,bp5,bv1,bv2,bv3,bv4,bv5\n")
        ts.write("date,time,price,size\n")
        o_ap, o_av, o_bp, o_bv, o_tp, o_tv = 2, 2 + depth, 2 + 2 * depth, 2 + 3 * depth, 2 + 4 * depth, 2 + 4 * depth + trades
        last_t = 0
        for r in range(rec.shape[0]):
            t = int(rec[r, 0].astype(np.int32))
            last_t = t
            for i in range(trades):
                v = int(rec[r, o_tv + i])
                if v > 0:
                    ts.write("%d,%s,%.9g,%d\n" % (date, ms_to_str(t - 1), float(f32[r, o_tp + i]), v))
            cols = [str(date), ms_to_str(t)]
            cols += ["%.9g" % float(f32[r, o_ap + l]) for l in range(depth)]
            cols += ["%d" % int(rec[r, o_av + l]) for l in range(depth)]
            cols += ["%.9g" % float(f32[r, o_bp + l]) for l in range(depth)]
            cols += ["%d" % int(rec[r, o_bv + l]) for l in range(depth)]
            md.write(",".join(cols) + "\n")
        # two sentinel trade groups after the last depth row: the T&S streamer must not run dry first
        ts.write("%d,%s,1.0,1\n" % (date, ms_to_str(last_t + 3600000)))
        ts.write("%d,%s,1.0,1\n" % (date, ms_to_str(last_t + 3600001)))


def write_lobster(rec, depth, trades, levels, ob_path, msg_path):
    """One book's records -> LOBSTER orderbook + message files (prices x 10000).  Per event: one
    execution message (type 4) per trade slot, then one submission (type 1); every message row is
    followed by the event's snapshot in the orderbook file."""
    f32 = rec.view(np.float32)
    o_ap, o_av, o_bp, o_bv, o_tp, o_tv = 2, 2 + depth, 2 + 2 * depth, 2 + 3 * depth, 2 + 4 * depth, 2 + 4 * depth + trades

    def px(x):
        return int(round(float(x) * 10000))

    with open(ob_path, "w") as ob, open(msg_path, "w") as msg:
        for r in range(rec.shape[0]):
            t = int(rec[r, 0].astype(np.int32))
            snap = []
            for l in range(levels):
                if l < depth:
                    snap += [px(f32[r, o_ap + l]), int(rec[r, o_av + l]), px(f32[r, o_bp + l]), int(rec[r, o_bv + l])]
                else:
                    snap += [9999999999, 0, -9999999999, 0]
            line = ",".join(str(x) for x in snap) + "\n"
            oid = 1000 + r
            for i in range(trades):
                v = int(rec[r, o_tv + i])
                if v > 0:
                    msg.write("%.9f,4,%d,%d,%d,-1\n" % (t / 1000.0 + 1e-4, oid, v, px(f32[r, o_tp + i])))
                    ob.write(line)
            msg.write("%.9f,1,%d,%d,%d,1\n" % (t / 1000.0 + 2e-4, oid, 100, px(f32[r, o_bp + 0])))
            ob.write(line)

"""What the host-buffer hand-over of the boundary costs (lob_load_events: the caller's records over PCIe into a temporary
device copy, then the repack kernel into the device layout): GB/s from a pageable numpy array, and what that makes of a
whole episode at 65 536 books if the stream is used ONCE (the PCIe-inclusive figure DESIGN.md quotes; never bench.py's `value`).
    python tools/exp_upload.py [books]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rl_markets_amd import abi, engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
p = engine.default_params()
p.depth, p.max_trades, p.algo, p.theta_mode = 10, 2, abi.ALGO_QLAMBDA, abi.THETA_SHARED
g = engine.default_gen_params()
g.n_events = 64 + 2048
t0 = time.perf_counter()
rec = engine.gen_stream_host(g, 10, 2, 0, B)
print("host generator: %.2f s for %d books x %d events (%.2f GB)" % (time.perf_counter() - t0, B, g.n_events, rec.nbytes / 1e9))
import ctypes as C
lib = abi.load()
t0 = time.perf_counter()
rc = lib.lob_validate_stream(rec.ctypes.data_as(C.c_void_p), 10, 2, B, g.n_events)
dt = time.perf_counter() - t0
print("lob_validate_stream (host threads): rc %d, %.3f s = %.1f GB/s" % (rc, dt, rec.nbytes / dt / 1e9))
eng = engine.Engine(p, B)
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    eng.load_events(rec)
    eng.sync()
    dt = time.perf_counter() - t0
    best = min(best, dt)
    print("lob_load_events: %.3f s = %.1f GB/s" % (dt, rec.nbytes / dt / 1e9))
rate = rec.nbytes / best
full = 65536 * g.n_events * rec.shape[2] * 4
print("65 536 books x %d events = %.1f GB of caller records: %.2f s at that rate" % (g.n_events, full / 1e9, full / rate))

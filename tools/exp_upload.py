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

# ---- a fresh stream per episode: handed over between the episodes (lob_load_events) against staged while the episode runs (lob_stage_events)
def episode(eng, chunk=32):
    n = 0
    while True:
        eng.td_step(chunk)
        n += chunk
        if eng.counters()[2] == 0:
            return n
g2 = engine.default_gen_params()
g2.n_events = g.n_events
g2.seed = 4242
rec2 = engine.gen_stream_host(g2, 10, 2, B, B)
streams = [rec, rec2]
for mode in ("between", "staged"):
    eng.load_events(streams[0])
    eng.reset()
    eng.sync()
    c0 = eng.counters()[0]
    t0 = time.perf_counter()
    for ep in range(4):
        nxt = streams[(ep + 1) % 2]
        if mode == "staged":
            eng.stage_events(nxt)
        episode(eng)
        eng.clear_inventory()
        eng.handle_terminal()
        if mode == "between":
            eng.load_events(nxt)
        eng.reset()
    eng.sync()
    dt = time.perf_counter() - t0
    steps = eng.counters()[0] - c0
    print("fresh stream every episode, %s: 4 episodes, %d env-steps in %.3f s = %.1f M env-steps/s PCIe-inclusive" % (mode, steps, dt, steps / dt / 1e6))

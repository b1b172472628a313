"""How many events one env-step consumes (CPU only, no GPU): a step of Base::performAction runs NextState until the
aggregate midprice move is no longer zero (reference src/environment/base.cpp:285-305) -- a property of the stream alone.
Prints the distribution over bench.py's synthetic streams and the expected maximum among the 64 books of a wave and among the
65 536 books of a launch: env_step_kernel waits for the longest step of the whole batch.
    python tools/exp_step_lengths.py"""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
from rl_markets_amd import engine, abi
g = engine.default_gen_params()
g.n_events = 2112
D,T = 10,2
B = 2048
rec = engine.gen_stream_host(g, D, T, 0, B)   # [B][n][W]
W = rec.shape[2]
lib = abi.load()
# mid price per event: best ask / best bid = first entries of px arrays
apx = rec[..., 2:2+D].view(np.float32)[..., 0].astype(np.float64)
# find bid px offset: layout of host records (lob_rec_*): time, flags, ask px[D], ask vol[D], bid px[D], bid vol[D]...
bpx = rec[..., 2+2*D:2+3*D].view(np.float32)[..., 0].astype(np.float64)
mid = (apx + bpx) / 2
print("sample mids", mid[0,:5], "spread", (apx-bpx)[0,:5])
mpm = np.diff(mid, axis=1)      # move at event j+1
n = mpm.shape[1]
# step length starting after event k: smallest L>=1 with |sum_{i=k..k+L-1} mpm| >= 1e-5
lens = []
for b in range(B):
    m = mpm[b]
    k = 70
    while k < n:
        s = 0.0; L = 0
        while k + L < n:
            s += m[k+L]; L += 1
            if abs(s) >= 1e-5: break
        lens.append(L); k += L
lens = np.array(lens)
print("steps", len(lens), "mean", lens.mean(), "p50", np.median(lens), "p99", np.percentile(lens,99), "max", lens.max())
for nn in range(1, 25):
    print(nn, (lens >= nn).mean())
# expected max among 64 and among 65536 draws
rng = np.random.default_rng(0)
draw = rng.choice(lens, size=(2000, 64))
print("E max of 64:", draw.max(axis=1).mean())
draw = rng.choice(lens, size=(50, 65536))
print("E max of 65536:", draw.max(axis=1).mean())

import sys, time
sys.path.insert(0,'.')
from rl_markets_amd import abi, engine
def run(move, steps=40):
    p=engine.default_params(); p.depth=10; p.algo=abi.ALGO_QLAMBDA
    g=engine.default_gen_params(); g.n_events=1200
    g.move_prob_q16=move; 
    if move>=65536: g.spread2_prob_q16=0
    eng=engine.Engine(p,65536); eng.gen_events(g); eng.reset()
    eng.td_step(10); eng.sync()
    eng.kernel_timing(True)
    c0=eng.counters()
    t=time.perf_counter(); eng.td_step(steps); eng.sync(); dt=time.perf_counter()-t
    c1=eng.counters()
    print('move',move,'ms/step',round(dt/steps*1e3,3),'events/step',round((c1[1]-c0[1])/(c1[0]-c0[0]),3), {k:round(eng.kernel_time_ms(k)[0],3) for k in ('act_kernel','env_kernel','learn_kernel','update_kernel')})
    eng.close()
run(65536); run(int(0.35*65536)); run(int(0.1*65536), steps=20)

"""One process per GPU on ONE node, without an external launcher.

`bench.py --gpus N` and the tests call spawn_ranks() when no launcher has set
WORLD_SIZE: it starts N copies of the command with RANK / LOCAL_RANK /
WORLD_SIZE and a rendezvous file path in LOB_RDZV (rl_markets_amd.comm.RcclComm
reads the RCCL token from it), relays rank 0's stdout, prefixes the other
ranks' output on stderr, and tears every rank down as soon as one fails.
Under `python -m torch.distributed.run` the same variables come from the
launcher and the rendezvous path is derived from its MASTER_PORT and process id
(all workers of one node share that parent)."""
import os
import subprocess
import sys
import tempfile
import threading
import time


def rank_env():
    """(rank, local_rank, world) from the environment a launcher set; (0, 0, 1) without one."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))),
            int(os.environ.get("WORLD_SIZE", "1")))


def rendezvous_path():
    p = os.environ.get("LOB_RDZV")
    if p:
        return p
    # torchrun: every worker of this node is a child of the same agent process
    return os.path.join(tempfile.gettempdir(), "lob_rdzv_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid()))


def _pump(stream, sink, prefix):
    for line in iter(stream.readline, ""):
        sink.write(prefix + line)
        sink.flush()
    stream.close()


def spawn_ranks(argv, n, env=None, timeout=None):
    """Run `argv` as n ranks; returns the first non-zero exit code, else 0."""
    base = dict(os.environ if env is None else env)
    rdzv = os.path.join(tempfile.gettempdir(), "lob_rdzv_%d_%d" % (os.getpid(), int(time.time() * 1e3) & 0xffffff))
    for stale in (rdzv, rdzv + ".tmp"):
        if os.path.exists(stale):
            os.unlink(stale)
    procs, pumps = [], []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOB_RDZV=rdzv)
        p = subprocess.Popen(argv, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, bufsize=1)
        procs.append(p)
        pumps.append(threading.Thread(target=_pump, args=(p.stdout, sys.stdout if r == 0 else sys.stderr, "" if r == 0 else "[rank %d] " % r), daemon=True))
        pumps.append(threading.Thread(target=_pump, args=(p.stderr, sys.stderr, "[rank %d] " % r), daemon=True))
    for t in pumps:
        t.start()
    rc, t0 = 0, time.time()
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                sys.stderr.write("[launch] rank %d exited with %d: stopping the other ranks\n" % (r, code))
                for q in live:
                    procs[q].terminate()   # exactly the processes started above
        if timeout is not None and time.time() - t0 > timeout and live:
            rc = rc or 124
            for q in live:
                procs[q].kill()
        time.sleep(0.05)
    for t in pumps:
        t.join(timeout=5)
    for stale in (rdzv, rdzv + ".tmp"):
        if os.path.exists(stale):
            os.unlink(stale)
    return rc

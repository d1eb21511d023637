"""One process per GPU without an external launcher: `python bench.py --gpus N` (or any script of this repository) started
plainly, i.e. with no WORLD_SIZE in the environment, re-runs ITSELF N times as child processes with the torch.distributed
environment of `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR =
127.0.0.1, a free MASTER_PORT) and relays their output.  The reference has nothing comparable: `src/config/config.py:12` pins
CUDA_VISIBLE_DEVICES=0 (single GPU).

The parent must not have touched the GPU when it calls this (it never does afterwards either): children are ordinary
subprocesses (fork + exec of the interpreter), which is only allowed on this pool from a process without a HIP context.
"""
import os
import socket
import subprocess
import sys
import threading


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def need_self_launch(n_ranks):
    """True when this process was started plainly (no launcher environment) and more than one rank is asked for."""
    return n_ranks > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ


def launch_ranks(n_ranks, argv=None, env=None, timeout=None, relay_rank=0, out=None):
    """Start `n_ranks` copies of `argv` (default: this very command line), rank r with LOCAL_RANK = RANK = r.  stdout of rank
    `relay_rank` is passed through line by line (the bench's ONE JSON line); every rank's stderr goes to ours.  Returns the
    first non-zero exit code (0 if all ranks succeeded); when one rank fails the others are terminated."""
    argv = list(argv) if argv is not None else [sys.executable] + sys.argv
    out = out or sys.stdout
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), LOCAL_WORLD_SIZE=str(n_ranks))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this pool
    procs = []
    for r in range(n_ranks):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(argv, env=e, stdout=subprocess.PIPE if r == relay_rank else subprocess.DEVNULL, stderr=None,
                                      text=True, start_new_session=False))

    def pump(p):
        for line in p.stdout:
            out.write(line)
            out.flush()
    t = threading.Thread(target=pump, args=(procs[relay_rank],), daemon=True)
    t.start()
    rc = 0
    try:
        pending = list(procs)
        import time
        deadline = None if timeout is None else time.time() + timeout
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:          # a dead rank leaves the others in a collective forever
                        q.terminate()
            if deadline is not None and time.time() > deadline:
                rc = rc or 124
                for q in pending:
                    q.terminate()
                deadline = None
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        t.join(timeout=5)
    return rc

"""Start one process per GPU on this node -- what ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N``
does, reduced to what the hot path needs (the reference has no launcher: it is single-process).

The parent must not have touched the GPU: on this pool a process that has initialised HIP may neither exec another
program nor safely fork workers that use the device.  So this module imports nothing but the standard library, the
children are fresh interpreters started with ``subprocess`` (never a re-exec of the caller), and ``bench.py`` calls
``spawn_ranks`` before it imports ``physicl_amd._hip`` or torch.
"""
import ctypes
import os
import signal
import socket
import subprocess
import sys
import threading
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(base, rank, world, port):
    """Environment of rank ``rank``: the variables torch.distributed.run would set, rendezvous on 127.0.0.1
    (the container hostname may not resolve), dmabuf IPC for RCCL on this driver."""
    env = dict(base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return env


def spawn_ranks(world, cmd, env=None, poll_s=0.05, grace_s=10.0):
    """Run ``cmd`` (argv list) ``world`` times, rank r with ``rank_env(env, r, world, port)``.  Rank 0's stdout is
    captured and returned; every other stream goes to this process's stderr.  Returns (exit code, rank-0 stdout):
    the code is 0 only if every rank exited 0; as soon as one rank fails the others are terminated (they would wait
    in a collective forever)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    base = dict(os.environ if env is None else env)
    port = free_port()
    procs = []
    # a SIGTERM / SIGHUP / SIGINT to this process (a driver's timeout, Ctrl-C of a wrapper) must take the ranks down
    # with it: they would otherwise sit in a collective, holding their GPUs, until its timeout.  The handlers only RECORD
    # the signal -- nothing is raised inside Popen (a rank forked but not yet in ``procs`` would be lost) or inside the
    # clean-up -- and the loops below look at the record; should this process die without running its clean-up (SIGKILL),
    # the kernel sends every rank SIGTERM (PR_SET_PDEATHSIG, set in the child between fork and exec).
    try:
        _libc()                                  # loaded here, not between fork and exec
    except OSError:
        pass
    got = []                                     # signal numbers, in order of arrival

    def _record(signum, frame):
        got.append(signum)

    old_handlers = {}
    if threading.current_thread() is threading.main_thread():
        for sig in (signal.SIGTERM, signal.SIGHUP, signal.SIGINT):
            old_handlers[sig] = signal.signal(sig, _record)
    rc = 0
    captured, reader = [], None
    parent = os.getpid()
    try:
        try:
            for r in range(world):
                if got:
                    break
                procs.append(subprocess.Popen(list(cmd), env=rank_env(base, r, world, port),
                                              preexec_fn=lambda: _die_with_parent(parent),
                                              stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
            if procs:
                reader = threading.Thread(target=lambda: captured.append(procs[0].stdout.read()), daemon=True)
                reader.start()
            while not got:
                states = [p.poll() for p in procs]
                bad = [s for s in states if s not in (None, 0)]
                if bad:
                    rc = bad[0] if bad[0] > 0 else 1          # killed by a signal -> 1
                    break
                if all(s == 0 for s in states):
                    break
                time.sleep(poll_s)
            if got:
                rc = 128 + got[0]
        finally:
            _stop(procs, grace_s)                # (a second signal during the clean-up is recorded, not raised)
    finally:
        for sig, h in old_handlers.items():
            signal.signal(sig, h)
    if reader is not None:
        reader.join(timeout=grace_s)
    out = captured[0] if captured else b""
    return rc, out.decode("utf-8", "replace")


def _die_with_parent(parent_pid=None):
    """In the child, before exec: ask the kernel for SIGTERM when the launching process dies (Linux prctl).  A launcher
    that died between the fork and the prctl is noticed by comparing the parent's pid afterwards."""
    try:
        _libc().prctl(1, int(signal.SIGTERM), 0, 0, 0)     # PR_SET_PDEATHSIG = 1
    except Exception:                            # noqa: BLE001 -- not Linux / no prctl: the handlers above still apply
        return
    if parent_pid is not None and os.getppid() != parent_pid:
        os._exit(1)


_LIBC = []


def _libc():
    if not _LIBC:
        _LIBC.append(ctypes.CDLL(None, use_errno=True))
    return _LIBC[0]


def _stop(procs, grace_s):
    """Terminate exactly the processes started here (by PID), then kill what ignores the signal."""
    live = [p for p in procs if p.poll() is None]
    for p in live:
        try:
            p.send_signal(signal.SIGTERM)
        except OSError:
            pass
    t_end = time.time() + grace_s
    for p in live:
        try:
            p.wait(timeout=max(0.0, t_end - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()

"""One process per GPU without an external launcher (stdlib only: the parent never imports torch's GPU side and never touches a GPU).

Replaces the reference's single-process ``nn.DataParallel`` wrap (Train/temporal_transformer_shanghaitech.py:76-78, ``--gpu`` ->
``CUDA_VISIBLE_DEVICES`` :328): ``python Train/<script>.py --data_parallel --gpu 0,1,2,3`` and ``python bench.py --gpus N`` both
start their ranks through ``launch_ranks``: N fresh child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
(127.0.0.1, a free port), created BEFORE this process initialises any GPU state (a rank is never an exec from a GPU-initialised
process).  Also runs unchanged under ``python -m torch.distributed.run``: the callers skip the launcher when WORLD_SIZE is set.
"""
from __future__ import annotations

import os
import selectors
import socket
import subprocess
import sys
import tempfile
import time


def _tail(path, n=25):
    try:
        with open(path, "r", errors="replace") as f:
            return f.readlines()[-n:]
    except OSError:
        return []


def mark_rank_ready():
    """Called by a rank once it has joined its process group: tells the launcher's init watchdog that this rank is past the
    rendezvous / RCCL communicator set-up (the one place a multi-GPU job hangs without a trace)."""
    d = os.environ.get("LSTC_READY_DIR")
    if d:
        try:
            open(os.path.join(d, "rank%s" % os.environ.get("RANK", "0")), "w").close()
        except OSError:
            pass


def launch_ranks(n, argv, script=None, rank_timeout_s=600.0, relay="json", devices=None, tag="bench", extra_env=None,
                 init_timeout_s=0.0):
    """Start ``n`` rank processes of ``script`` (default: the calling program) with ``argv``; fail if any rank fails.

    relay = "json" (bench.py): rank 0's stdout is scanned for ONE JSON line, printed when every rank has exited (0 then means the
    line exists); other stdout lines go to stderr.  relay = "all" (Train/*.py): rank 0's stdout is passed through line by line.
    ``devices``: the GPU ids the ranks share (``--gpu 0,1,2,3``): exported to every rank as HIP_VISIBLE_DEVICES (and
    CUDA_VISIBLE_DEVICES, which ROCm honours too and the reference sets); LOCAL_RANK r selects the r-th of them.

    Watchdog: a rank stuck in RCCL initialisation (or anywhere else) would otherwise hang the parent until the caller's own
    timeout with nothing to read.  After ``rank_timeout_s`` seconds (0 = none) the exact child PIDs started here are terminated
    (then killed), every rank's last stderr lines are printed and the launcher returns 1.  Every rank's stderr goes to its own
    temporary file (relayed to this process's stderr at the end), so the tails exist whichever rank is the stuck one.
    ``init_timeout_s`` > 0: a second watchdog for the START of the job only - every rank must call ``mark_rank_ready()`` (after
    ``init_process_group``) within that many seconds, else all ranks are stopped the same way; a job that got past its rendezvous
    may then run for days (the Train/*.py path: no overall limit by default, 15 minutes for the rendezvous).
    Returns 0, or 1 (with the reason on stderr).  The last stdout line of rank 0 that starts with ``LSTC_RESULT `` is kept in
    ``launch_ranks.last_result`` (Train/*.py hand their return value back through it)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs, errs = [], []
    t_start = time.monotonic()
    launch_ranks.last_result = None
    ready_dir = tempfile.mkdtemp(prefix=f"lstc_{tag}_ready_")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LSTC_LAUNCHED="1", LSTC_READY_DIR=ready_dir,
                   PYTHONUNBUFFERED="1")                        # rank 0's lines reach the relay as they are written
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this pool
        if devices:
            env["HIP_VISIBLE_DEVICES"] = env["CUDA_VISIBLE_DEVICES"] = ",".join(str(d) for d in devices)
        if extra_env:
            env.update(extra_env)
        ef = tempfile.NamedTemporaryFile("w+", prefix=f"lstc_{tag}_rank{r}_", suffix=".err", delete=False)
        errs.append(ef)
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(sys.argv[0])] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else ef, stderr=ef))
    line = None
    failed = None
    timed_out = False
    all_ready = init_timeout_s <= 0
    out0 = procs[0].stdout
    fd0 = out0.fileno()
    sel = selectors.DefaultSelector()
    sel.register(fd0, selectors.EVENT_READ)
    open0 = True
    pending = b""

    def handle(ln):
        nonlocal line
        if ln.startswith("LSTC_RESULT "):
            launch_ranks.last_result = ln[len("LSTC_RESULT "):].strip()
        elif relay == "all":
            sys.stdout.write(ln); sys.stdout.flush()
        elif ln.lstrip().startswith("{"):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    while True:
        if open0:
            for _key, _ in sel.select(timeout=0.5):
                # the raw descriptor, own line splitting: a buffered readline() takes a whole pipe chunk and leaves the further
                # complete lines in its buffer with the descriptor no longer readable - they came out at the next write or at EOF
                chunk = os.read(fd0, 65536)
                if chunk == b"":
                    open0 = False
                    sel.unregister(fd0)
                    if pending:
                        handle(pending.decode("utf-8", "replace") + "\n")
                        pending = b""
                    break
                pending += chunk
                *full, pending = pending.split(b"\n")
                for b_ in full:
                    handle(b_.decode("utf-8", "replace") + "\n")
        else:
            time.sleep(0.2)
        codes = [p.poll() for p in procs]
        bad = [i for i, c in enumerate(codes) if c not in (None, 0)]
        if bad and failed is None and not timed_out:
            failed = (bad[0], codes[bad[0]])
            for p in procs:                                      # exact PIDs we started, nothing by pattern
                if p.poll() is None:
                    p.terminate()
        if not all_ready:
            all_ready = all(os.path.exists(os.path.join(ready_dir, f"rank{r}")) for r in range(n))
        over = rank_timeout_s > 0 and time.monotonic() - t_start > rank_timeout_s
        init_over = not all_ready and time.monotonic() - t_start > init_timeout_s
        if not timed_out and failed is None and (over or init_over) and any(c is None for c in codes):
            timed_out = True
            stuck = [i for i, c in enumerate(codes) if c is None]
            if init_over and not over:
                late = [r for r in range(n) if not os.path.exists(os.path.join(ready_dir, f"rank{r}"))]
                sys.stderr.write(f"[{tag}] watchdog: rank(s) {late} have not joined the process group after {init_timeout_s:.0f} s "
                                 f"(rendezvous / RCCL initialisation); stopping all ranks\n")
            else:
                sys.stderr.write(f"[{tag}] watchdog: rank(s) {stuck} still running after {rank_timeout_s:.0f} s; stopping all ranks\n")
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.monotonic() + 10.0
            while time.monotonic() < t_kill and any(p.poll() is None for p in procs):
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
        if all(c is not None for c in codes) and (not open0 or timed_out):
            break
    for r, ef in enumerate(errs):                                # relay the ranks' stderr: rank 0 whole, the others' tails
        ef.flush(); ef.close()
        lines = _tail(ef.name, 10 ** 6 if r == 0 and failed is None and not timed_out else 25)
        if lines and (r == 0 or failed is not None or timed_out):
            sys.stderr.write(f"---- rank {r} stderr{' (last lines)' if (failed is not None or timed_out) else ''} ----\n" + "".join(lines))
        try:
            os.unlink(ef.name)
        except OSError:
            pass
    try:
        for f in os.listdir(ready_dir):
            os.unlink(os.path.join(ready_dir, f))
        os.rmdir(ready_dir)
    except OSError:
        pass
    if timed_out:
        return 1
    if failed is not None:
        sys.stderr.write(f"[{tag}] rank {failed[0]} exited with code {failed[1]}; all ranks stopped\n")
        return 1
    if relay == "json":
        if line is None:
            sys.stderr.write(f"[{tag}] rank 0 printed no JSON line\n")
            return 1
        print(line, flush=True)
    return 0


launch_ranks.last_result = None


def parse_devices(gpu) -> list:
    """``--gpu 0,1,2,3`` -> ["0", "1", "2", "3"] (the reference exports the string as CUDA_VISIBLE_DEVICES)."""
    return [d.strip() for d in str(gpu if gpu is not None else "0").split(",") if d.strip() != ""]

"""`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (launch_ranks); StripsGuard releases every rank of the
multi-GPU strips leg when one of them fails or hangs."""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys
import time

import torch

from .common import ROOT

BENCH_PY = os.path.join(ROOT, "bench.py")


def launch_ranks(n: int, script: str = None, argv=None) -> int:
    """Start n children (one per GPU) running this same script; the parent never initialises the GPU
    (torch.cuda.device_count() does not, on this image) and never re-execs itself.  All children are polled: when any of them
    exits non-zero the others are terminated within seconds (a rank that died at start-up would otherwise leave the rest in
    dist.barrier() until the collective timeout) and every rank's stderr tail is forwarded."""
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n_dev = torch.cuda.device_count()
    if n_dev < n and "BENCH_DEVICE" not in os.environ:
        print(f"bench.py: --gpus {n} but only {n_dev} HIP device(s) visible "
              "(BENCH_DEVICE=0 BENCH_BACKEND=gloo shares one GPU between the ranks for a functional check)", file=sys.stderr)
        return 2
    procs, logs = [], []
    out0 = tempfile.TemporaryFile("w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        logs.append(tempfile.TemporaryFile("w+"))
        procs.append(subprocess.Popen([sys.executable, script or BENCH_PY] + list(sys.argv[1:] if argv is None else argv), env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=logs[r], text=True))
    codes = [None] * n
    failed_first = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0) and failed_first is None:
                    failed_first = r
        if failed_first is not None:
            deadline = time.time() + 5.0
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()                      # exact children of this process, by handle
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            break
        time.sleep(0.05)
    out0.seek(0)
    text0 = out0.read()
    if text0:
        sys.stdout.write(text0)
        sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    for r, log in enumerate(logs):
        log.seek(0)
        tail = log.read()[-(4000 if bad else 1500):]
        if tail.strip():
            print(f"---- rank {r} stderr (tail) ----\n{tail}", file=sys.stderr)
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}; first failure: rank {failed_first}; the other ranks were terminated",
              file=sys.stderr)
        return 1
    return 0


class StripsGuard:
    """Watcher thread of the multi-GPU strips leg (armed on construction).  `fail(msg)` posts msg in the process group's store and leaves
    through `_bail`; the watcher polls the store (and the deadline) twice a second and leaves the same way when any rank has failed or
    the leg hangs: rank 0 prints `line_fn(msg)` - the slides-leg line with `strips_error` - and every rank exits with code 3 at once,
    without entering another collective.  Covers asymmetric failures (one rank raises, its peers sit in a collective)."""
    KEY = "bench_strips_failed"
    EXIT_CODE = 3

    def __init__(self, dist, rank, line_fn, deadline_s):
        import threading
        self.rank, self.line_fn, self.deadline_s = rank, line_fn, deadline_s
        self.lock, self.done = threading.Lock(), False
        try:
            self.store = dist.distributed_c10d._get_default_store()
        except Exception:                                           # noqa: BLE001  (no store: the deadline alone guards the leg)
            self.store = None
        self.t_end = time.time() + deadline_s
        threading.Thread(target=self._watch, daemon=True).start()

    def _watch(self):
        while not self.done:
            time.sleep(0.5)
            msg = None
            try:
                if self.store is not None and self.store.check([self.KEY]):
                    msg = self.store.get(self.KEY).decode(errors="replace")
            except Exception:                                       # noqa: BLE001  (rank 0 hosts the store: it has left already)
                msg = "the strips leg failed on another rank"
            if msg is None and time.time() > self.t_end:
                msg = f"the strips leg did not finish within {self.deadline_s:.0f} s (BENCH_STRIPS_DEADLINE_S)"
            if msg is not None:
                self._bail(msg)

    def fail(self, msg):
        try:
            if self.store is not None:
                self.store.set(self.KEY, f"rank {self.rank}: {msg}")
        except Exception:                                           # noqa: BLE001
            pass
        self._bail(f"rank {self.rank}: {msg}")
        time.sleep(3600)                                            # the watcher is already leaving for this process

    def _bail(self, msg):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.rank == 0:
                sys.stdout.write(json.dumps(self.line_fn(msg)) + "\n")
            else:
                time.sleep(1.0)                                     # rank 0 hosts the store: let the other watchers read the key first
            sys.stdout.flush()
            sys.stderr.write(f"bench.py: rank {self.rank} leaves the strips leg: {msg}\n")
            sys.stderr.flush()
            # a failed or hung strips leg is a FAILED job (exit code 3 on every rank; launch_ranks / torchrun report it): the slides line with
            # strips_error is still printed, so the collective-free figure is not lost - but nobody can take it for the strips headline
            os._exit(self.EXIT_CODE)

    def disarm(self):
        with self.lock:
            self.done = True


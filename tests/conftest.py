"""pytest config: registers the ``gpu`` marker and puts the repo root + the product package on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "neurips-cellseg_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _effective_cpus():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


try:   # the GPU box shows 256 logical CPUs under a 16-CPU cgroup quota: keep the torch-CPU oracle off 256 threads
    import torch
    torch.set_num_threads(min(_effective_cpus(), 16))
except ImportError:
    pass


# ----------------------------------------------------------------------------------------------------------------------
# one short run of the training CLI per test session (GPU tests): work_dir + its console log, shared by the CLI tests
# ----------------------------------------------------------------------------------------------------------------------
import subprocess  # noqa: E402

import pytest  # noqa: E402


def cli_env(**extra):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([PKG, ROOT, os.environ.get("PYTHONPATH", "")]), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update({k: str(v) for k, v in extra.items()})
    return env


def run_bench(*args, expect_rc=0, timeout=900, **env):
    """`python bench.py ...` as a child process -> (compact, full): the LAST stdout line parsed (the driver's view: one JSON object under
    4 kB) and the full record the run wrote beside it (gpurun_out/<BENCH_FULL_NAME>, removed here).  stdout must be that ONE line."""
    import json
    import uuid
    name = f"bench_full_test_{uuid.uuid4().hex[:10]}.json"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout,
                         env=cli_env(BENCH_FULL_NAME=name, **env))
    assert out.returncode == expect_rc, (out.returncode, out.stderr[-3000:])
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096, (len(lines), [len(x) for x in lines])
    compact = json.loads(lines[0])
    path = os.path.join(ROOT, "gpurun_out", name)
    assert compact["full_record"] == f"gpurun_out/{name}" and os.path.exists(path)
    full = json.load(open(path))
    os.remove(path)
    return compact, full


def run_module(mod, *args, **env):
    return subprocess.run([sys.executable, "-m", mod, *args], capture_output=True, text=True, timeout=900, env=cli_env(**env))


@pytest.fixture(scope="session")
def trained_cli(tmp_path_factory):
    work = tmp_path_factory.mktemp("work")
    out = run_module("cellseg_amd.model_training_3class", "--synthetic", "24", "--max_epochs", "4", "--val_start_epoch", "1",
                     "--val_interval", "1", "--batch_size", "8", "--num_workers", "2", "--work_dir", str(work))
    assert out.returncode == 0, out.stderr[-2000:]
    return work, out.stdout

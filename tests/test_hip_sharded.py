"""The multi-GPU driver on one GPU: world_size 1 over RCCL exercises every piece the N>1 bench
uses (HipExecutor on an external HIP stream, raw device pointers wrapped as torch tensors,
enqueue_ops between barriers, in-place all_gather_into_tensor) and must reproduce plain stepping.
Runs in a child process: see tests/sharded_ws1_worker.py for why."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_sharded_driver_world_size_one():
    worker = Path(__file__).parent / "sharded_ws1_worker.py"
    r = subprocess.run([sys.executable, str(worker)], capture_output=True, text=True, timeout=580)
    assert r.returncode == 0 and "SHARDED_WS1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.timeout(600)
def test_native_shard_path_world_size_one():
    """The same step behind the C ABI (zgml_hip_shard_*: the library owns the RCCL communicator and enqueues the
    all-gathers itself; graph and eager forms) — what `bench.py --gpus N` runs on every rank."""
    worker = Path(__file__).parent / "sharded_native_worker.py"
    r = subprocess.run([sys.executable, str(worker)], capture_output=True, text=True, timeout=580)
    assert r.returncode == 0 and "SHARDED_NATIVE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]

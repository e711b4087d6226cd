"""bench.py under a launcher on the GPU box: one rank, but the SAME torch.distributed / RCCL calls the 8-GPU run makes (process group on
the device, barrier, per-rank gather of the wall times, max-reduce) -- `use_dist` is keyed on the launcher's environment, not on the world
size, precisely so that this path is exercised on a 1-GPU lease.  And the refusal: `--gpus 2` on a box with one device must fail loudly."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_one_rank_under_torchrun_uses_rccl():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                          "--no-cpu-baseline", "--sustain-seconds", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["collective_ranks"] == 1 and line["collective_backend"] == "nccl (RCCL)"
    assert len(line["ms_per_step_per_rank"]) == 1 and line["value"] > 1000


def test_bench_refuses_two_gpus_on_a_one_gpu_box():
    from parakeet_cpp_amd import capi
    if capi.device_count() != 1:
        pytest.skip("needs exactly one visible device")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "only 1 MI355X device(s) visible" in (out.stdout + out.stderr)

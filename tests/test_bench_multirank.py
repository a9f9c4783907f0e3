"""The N > 1 code path of bench.py inside `pytest -m gpu`: rendezvous on 127.0.0.1, barriers, max-over-ranks timing, the rank-0 JSON line and -
in `--mode train` - the bucketed gradient all-reduce issued under the encoder backward (train.py:184-193's hook point).  The GPU box has ONE
device and RCCL refuses two ranks on one device, so the two ranks share GPU 0 and talk over gloo (`L2S_BENCH_ONE_DEVICE`, `L2S_BENCH_BACKEND`:
test hooks of bench.py); everything else is the code the driver's 2/4/8-GPU runs execute."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(extra, timeout=900):
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, L2S_BENCH_ONE_DEVICE="1", L2S_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--skip-cpu-baseline"] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_inference_two_ranks():
    line = _launch(["--steps", "4", "--warmup", "2", "--group", "2"])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["scaling"] == "weak" and line["unit"] == "mel-frames/s"
    assert line["config"]["ranks"] == 2 and line["config"]["collective_backend"] == "gloo"
    assert line["value"] > 0 and line["cpu_baseline"] is None
    # whole-job aggregate: both ranks' frames over the max-over-ranks time
    assert abs(line["value"] - 2 * 32 * 300 * 4 / (line["ms_per_step"] * 4 / 1e3)) / line["value"] < 1e-6
    assert 0 < line["roofline"]["frac"] < 1


@pytest.mark.gpu
def test_bench_train_two_ranks_allreduce():
    line = _launch(["--mode", "train", "--steps", "2", "--warmup", "1"])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["unit"] == "clips/s"
    assert "dp2" in line["config"]["parallelism"]
    assert line["value"] > 0 and line["final_loss"] == line["final_loss"]      # finite (NaN != NaN)

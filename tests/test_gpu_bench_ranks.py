"""bench.py's --gpus N path end to end on one GPU: four ranks share the device, the exchange is
staged through host memory (gloo) instead of RCCL, everything else -- owner planning with a
time-sliced hot key, device routing, halo copies, the join and range stats on what each rank
received, the row-conservation check, the single JSON line -- is the code the multi-GPU run
executes."""
import json
import os
import socket
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_four_ranks_one_gpu():
    port = str(_free_port())
    env = dict(os.environ, TEMPO_BENCH_BACKEND="gloo", TEMPO_BENCH_VERBOSE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=port)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"),
           "--gpus", "4", "--rows", "4000000", "--steps", "2", "--warmup", "1", "--cpu-sample", "0"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=110)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # exactly one JSON line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["scaling"] == "weak" and out["steps"] == 2
    assert out["value"] > 0 and out["config"]["rows_per_gpu_per_op"] == 4000000
    assert out["config"]["operators_on_held_rows_ms_per_step"] > 0   # the step less its shuffle, timed beside it
    held = out["config"]["rows_held_per_rank"]
    assert len(held) == 4 and sum(r[0] for r in held) == 4 * 2000000 and sum(r[2] for r in held) == 4 * 4000000
    assert "join keys sliced: True" in p.stderr        # the Zipf head is cut along the time axis at 4 ranks

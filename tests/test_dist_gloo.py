"""World-size-2 test of the key shuffle on CPU (gloo): shard by key owner, run the operator per
rank, and the union of the ranks' results must equal the single-process result row for row."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_key_shuffle_two_ranks(tmp_path, orc):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    rng = np.random.default_rng(1234)
    n, nkeys = 4000, 37
    base = 1_596_240_000_000_000
    key = rng.integers(0, nkeys, n).astype(np.int32)
    ts = base + rng.integers(0, 600, n) * 1_000_000
    x = rng.normal(100, 3, n)
    ref = orc.range_stats(key, ts, x, None, 30)
    want = {int(r): (ref["mean"][p_], ref["count"][p_], ref["stddev"][p_], ref["max"][p_])
            for p_, r in enumerate(ref["perm"])}
    seen = 0
    for rank in range(2):
        d = np.load(tmp_path / f"rank{rank}.npz")
        for i, r in enumerate(d["rid"]):
            m, c, s, mx = want[int(r)]
            assert d["count"][i] == c and d["mx"][i] == mx
            assert abs(d["mean"][i] - m) <= 1e-9 * abs(m) and abs(d["stddev"][i] - s) <= 1e-9 * max(abs(s), 1e-12)
            seen += 1
    assert seen == n

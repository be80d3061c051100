"""Random cases for the time-range sharding of series that cannot be hashed: simulated ranks
(2 to 8) own consecutive time ranges of a table split among them in row chunks; the union of the
ranks' results must be the operator on the whole table (the HIP operators on both sides; the
helpers are those of tests/timesplit_cases.py).

  python tools/fuzz_timesplit.py [--seconds 240] [--seed 1] [--log gpurun_out/fuzz_timesplit.log]
"""
import argparse
import json
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import timesplit_cases as tc  # noqa: E402
from tempo_amd import ops, timesplit  # noqa: E402


def one(rng):
    world = int(rng.choice([2, 3, 5, 8]))
    n = int(rng.choice([rng.integers(world, 60), rng.integers(60, 5_000), rng.integers(5_000, 90_000)]))
    n_keys = int(rng.choice([1, 1, 2, 4]))
    seed = int(rng.integers(0, 1 << 30))
    op = str(rng.choice(["range_stats", "ema", "lookback", "join"]))
    comm = timesplit.SimComm(world)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    if op == "range_stats":
        window = int(rng.choice([0, 1, 20, 300, 2000]))
        desc = "range_stats world=%d n=%d keys=%d window=%d seed=%d" % (world, n, n_keys, window, seed)
        yield desc
        have, want = tc.check_range_stats(comm, ops, T, n=n, n_keys=n_keys, window=window, seed=seed)
        assert have.keys() == want.keys(), "rows differ"
        tc.compare_range_stats(have, want)
    elif op == "ema":
        window = int(rng.choice([1, 2, 7, 60]))
        with_seq = bool(rng.random() < 0.4)
        desc = "ema world=%d n=%d keys=%d window=%d seq=%s seed=%d" % (world, n, n_keys, window, with_seq, seed)
        yield desc
        have, want = tc.check_ema(comm, ops, T, n=n, n_keys=n_keys, window=window, seed=seed, with_seq=with_seq)
        assert have == want, "ema differs"
    elif op == "lookback":
        size = int(rng.choice([1, 3, 12]))
        exact = bool(rng.random() < 0.5)
        desc = "lookback world=%d n=%d keys=%d size=%d exact=%s seed=%d" % (world, n, n_keys, size, exact, seed)
        yield desc
        have, want = tc.check_lookback(comm, ops, T, n=n, n_keys=n_keys, size=size, seed=seed, exact=exact)
        tc.compare_lookback(have, want)
    else:
        skip = bool(rng.random() < 0.6)
        with_seq = bool(rng.random() < 0.3)
        desc = "join world=%d n=%d keys=%d skip_nulls=%s seq=%s seed=%d" % (world, n, n_keys, skip, with_seq, seed)
        yield desc
        have, want = tc.check_join(comm, ops, T, n=n, n_keys=n_keys, seed=seed, skip_nulls=skip, with_seq=with_seq)
        assert have == want, "joined rows differ"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--log", default=os.path.join(ROOT, "gpurun_out", "fuzz_timesplit.log"))
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.log), exist_ok=True)
    t_end, done, bad = time.time() + args.seconds, 0, []
    with open(args.log, "w") as log:
        while time.time() < t_end:
            rng = np.random.default_rng(args.seed * 1_000_003 + done)
            desc, t0 = "", time.time()
            try:
                for desc in one(rng):
                    log.write("%5d start %s\n" % (done + 1, desc))
                    log.flush()
                status = "ok"
            except Exception as err:  # noqa: BLE001
                status = "BAD"
                text = [t.strip() for t in str(err).strip().splitlines() if t.strip()]
                bad.append(dict(case=done, what=desc, error="%s: %s" % (type(err).__name__, " / ".join(text[:4])),
                                trace=traceback.format_exc().splitlines()[-8:]))
            done += 1
            log.write("%5d %-5s %.2fs %s\n" % (done, status, time.time() - t0, desc))
            log.flush()
    print(json.dumps(dict(cases=done, disagreements=len(bad), seed=args.seed, first_bad=bad[:8])))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

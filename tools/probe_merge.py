import sys, os, time, json
sys.path.insert(0, os.getcwd())
import torch
import bench
from tempo_amd import _lib, ops
dev = torch.device("cuda", 0)
n = 250_000_000
tq = bench.gen_trades_quotes(dev, n, n, 10_000, 4567)
torch.cuda.synchronize()
_lib.profile_enable(True)
for skew in (10, 1e9, None):
    for rep in range(2):
        _lib.profile_reset()
        j = ops.asof_join_table(tq["key_l"], tq["ts_l"], tq["key_r"], tq["ts_r"], 10_000, [tq["trade_pr"]],
                                [tq["ts_r"], tq["bid"], tq["ask"]], [None, tq["bid_valid"], tq["ask_valid"]],
                                skip_nulls=True, ts_partition_val=skew, fraction=0.1)
        torch.cuda.synchronize()
        del j
    t = {}
    for name, tag, ms in _lib.profile_read():
        t[name] = t.get(name, 0) + ms
    print(skew, {k: round(v, 2) for k, v in t.items() if v > 0.5})

"""Read + write rate of a plain elementwise kernel (one 8-byte read and one 8-byte write per element,
seven output arrays off one input, 2 GiB each): what a streaming kernel reaches on this device, beside
the device-to-device copy of tools/copy_ceiling.py.  Prints one JSON line."""
import json

import torch


def timed(f, reps=5):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda", 0)
    m = 1 << 28
    x = torch.randn(m, dtype=torch.float64, device=dev)
    outs = [torch.empty(m, dtype=torch.float64, device=dev) for _ in range(7)]

    def adds():
        for o in outs:
            torch.add(x, 1.0, out=o)

    ms = timed(adds)
    n = 1 << 30
    a = torch.empty(n, dtype=torch.float64, device=dev)
    out = {"seven_adds_2GiB_each_read_plus_write_GBps": round(14 * m * 8 / ms / 1e6, 1),
           "sum_8GiB_read_GBps": round(n * 8 / timed(lambda: a.sum()) / 1e6, 1),
           "fill_8GiB_write_GBps": round(n * 8 / timed(lambda: a.fill_(1.5)) / 1e6, 1),
           "command": "python tools/rw_ceiling.py"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

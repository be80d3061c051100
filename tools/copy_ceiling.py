"""Device-to-device copy ceiling of the GPU at hand: the practical HBM bound next to the 8 TB/s
nominal peak.  Prints one JSON line; read + write bytes both count."""
import json
import torch


def main():
    dev = torch.device("cuda", 0)
    out = {}
    for gib in (1, 4, 16):
        n = gib * (1 << 30) // 8
        a = torch.empty(n, dtype=torch.int64, device=dev).fill_(3)
        b = torch.empty_like(a)
        for _ in range(3):
            b.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[f"copy_{gib}GiB_GBps"] = round(2 * n * 8 / (ms * 1e-3) / 1e9, 1)
        # read-only ceiling: a sum reduction
        for _ in range(2):
            a.sum()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            a.sum()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[f"read_{gib}GiB_GBps"] = round(n * 8 / (ms * 1e-3) / 1e9, 1)
        del a, b
    print(json.dumps(out))


if __name__ == "__main__":
    main()

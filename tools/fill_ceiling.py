"""Write-only rate of the device: time a fill of 16 GiB (what an operator whose output dwarfs its
input, such as the lookback feature tensor, can reach at best)."""
import json
import torch
x = torch.empty(2 << 30, dtype=torch.float64, device="cuda")
for _ in range(2):
    x.fill_(1.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    x.fill_(2.0)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(json.dumps({"fill_16GiB_GBps": round(x.numel() * 8 / ms / 1e6, 1), "ms": round(ms, 3)}))

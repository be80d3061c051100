"""Where does a multi-GB all-to-all on one rank go wrong?  Sends an index pattern through
all_to_all_single (with and without explicit split sizes) and through the list form
(grouped send/recv) at sizes around 2^31 and 2^32 bytes and reports the first wrong element.
Run on one GPU: python tools/a2a_probe.py"""
import json
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29599")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
out = []
for dtype, esz in ((torch.int64, 8), (torch.uint8, 1)):
    for nbytes in (1 << 30, (1 << 31) - 4096, (1 << 31) + 4096, (1 << 32) - 4096, (1 << 32) + 4096, 8_000_000_000):
        n = nbytes // esz
        if dtype == torch.int64:
            src = torch.arange(n, dtype=dtype, device=dev)
        else:
            src = (torch.arange(n, dtype=torch.int64, device=dev) % 251).to(torch.uint8)
        for form in ("single_splits", "single_nosplit", "list"):
            got = torch.zeros(n, dtype=dtype, device=dev)
            try:
                if form == "single_splits":
                    dist.all_to_all_single(got, src, output_split_sizes=[n], input_split_sizes=[n])
                elif form == "single_nosplit":
                    dist.all_to_all_single(got, src)
                else:
                    dist.all_to_all([got], [src])
                torch.cuda.synchronize()
                bad = (got != src)
                nbad = int(bad.sum())
                first = int(torch.nonzero(bad)[0]) if nbad else -1
                last = int(torch.nonzero(bad)[-1]) if nbad else -1
                rec = dict(dtype=str(dtype), bytes=nbytes, form=form, wrong=nbad, first_wrong=first, last_wrong=last,
                           first_wrong_byte=first * esz if nbad else -1)
                if nbad:
                    rec["got_at_first"] = int(got[first])
                del bad
            except Exception as e:  # noqa: BLE001
                rec = dict(dtype=str(dtype), bytes=nbytes, form=form, error=repr(e)[:300])
            print(json.dumps(rec), flush=True)
            out.append(rec)
            del got
        del src
dist.destroy_process_group()

"""What a plain streaming kernel reaches on this GPU at the sizes of the tile kernels: device-to-device copies through
torch (read n bytes + write n bytes), timed with events.  Context for roofline.frac: the 8 TB/s peak is not what a
40-us kernel over ~170 MB can see."""
import json

import torch

out = {}
for mb in (85, 171, 512, 2048):
    n = mb * 1000 * 1000 // 8
    a = torch.empty(n, dtype=torch.float64, device="cuda")
    b = torch.empty_like(a)
    a.fill_(1.0)
    for _ in range(5):
        b.copy_(a)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    ev[0].record()
    for k in range(40):
        b.copy_(a)
        ev[k + 1].record()
    torch.cuda.synchronize()
    us = sorted(ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(40))[20]
    out[f"copy_{mb}MB_each_way"] = {"median_us": us, "GBps_read_plus_write": 2 * mb * 1e6 / (us * 1e-6) / 1e9}
print(json.dumps(out))

"""Device sanity numbers: properties, copy bandwidth, launch overhead (used to calibrate roofline expectations)."""
import time
import torch

p = torch.cuda.get_device_properties(0)
print("name", p.name, "CUs", p.multi_processor_count, "mem GB", round(p.total_memory / 2**30, 1), "clock", getattr(p, "clock_rate", None),
      "L2", getattr(p, "L2_cache_size", None), "gcn", getattr(p, "gcnArchName", None))
for mb in (64, 512, 2048, 8192):
    n = mb * 2**20
    a = torch.empty(n, dtype=torch.uint8, device="cuda")
    b = torch.empty(n, dtype=torch.uint8, device="cuda")
    a.fill_(1)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"copy {mb} MB: {ms*1e3:.1f} us  -> {2*n/ms/1e6:.0f} GB/s (read+write)")
    e0.record()
    for _ in range(reps):
        a.fill_(2)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"fill {mb} MB: {ms*1e3:.1f} us  -> {n/ms/1e6:.0f} GB/s (write)")
    x = a.view(torch.float32)
    e0.record()
    for _ in range(reps):
        s = x.sum()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"sum  {mb} MB: {ms*1e3:.1f} us  -> {n/ms/1e6:.0f} GB/s (read)")
    del a, b

"""Developer probe: streaming read bandwidth vs buffer size (is there an Infinity-Cache plateau?)."""
import torch, time
dev = torch.device("cuda")
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * 1024 * 1024 // 8
    x = torch.ones(n, dtype=torch.float64, device=dev)
    for _ in range(5): x.sum()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    s.record()
    for _ in range(reps): x.sum()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    print("read  %5d MB  %7.1f us  %6.2f TB/s" % (mb, us, mb * 1.048576 / us))
    y = torch.empty_like(x)
    for _ in range(5): y.copy_(x)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps): y.copy_(x)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    print("copy  %5d MB  %7.1f us  %6.2f TB/s (read+write)" % (mb, us, 2 * mb * 1.048576 / us))
    del x, y

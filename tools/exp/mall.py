"""does the 256 MB memory-side cache show? read-only (sum), write-only (fill) and copy bandwidth over buffer sizes, repeated on the SAME buffer
(so a buffer that fits is resident from the pass before)"""
import torch
for mb in (8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    res = []
    for name, fn, bytes_ in (("read", lambda: a.sum(), 4 * n), ("write", lambda: b.fill_(1.0), 4 * n), ("copy", lambda: b.copy_(a), 8 * n),
                             ("write-then-read", lambda: (b.fill_(2.0), b.sum()), 8 * n)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(5, 2048 // mb)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append("%s %.2f TB/s" % (name, bytes_ * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12))
    print("%5d MB: " % mb + "  ".join(res))

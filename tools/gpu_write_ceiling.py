"""What a write-only kernel can reach on this box: torch fill_ / copy_ rates, the ceiling K assembly is measured against."""
import torch
dev = torch.device('cuda:0')
for mb in (200, 536, 2147):
    n = mb * 1000 * 1000 // 8
    a = torch.empty(n, dtype=torch.float64, device=dev)
    b = torch.empty(n, dtype=torch.float64, device=dev)
    for name, fn, bytes_ in (('fill_', lambda: a.fill_(1.5), 8 * n), ('copy_', lambda: a.copy_(b), 16 * n),
                             ('mul_', lambda: a.mul_(1.0001), 16 * n)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f'{name:6s} {mb:5d} MB  {ms:8.3f} ms  {bytes_ / ms / 1e6:8.0f} GB/s')

"""What a pure write stream reaches on this card (the saving forward writes 15.8 GB per launch: is 3.4 TB/s its roof?)."""
import torch
n = 15_800_000_000 // 4
x = torch.empty(n, device='cuda', dtype=torch.float32)
y = torch.empty(n // 2, device='cuda', dtype=torch.float32)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: x.fill_(1.0)); print('fill   %.1f GB in %.2f ms = %.2f TB/s written' % (n * 4e-9, ms, n * 4e-9 / ms))
ms = t(lambda: x.zero_()); print('zero   %.1f GB in %.2f ms = %.2f TB/s written' % (n * 4e-9, ms, n * 4e-9 / ms))
ms = t(lambda: y.copy_(x[:n // 2])); print('copy   %.1f GB read + %.1f GB written in %.2f ms = %.2f TB/s total' % (n * 2e-9, n * 2e-9, ms, n * 4e-9 / ms))
ms = t(lambda: x.mul_(1.0001)); print('scale  %.1f GB read + written in place in %.2f ms = %.2f TB/s total' % (n * 4e-9, ms, n * 8e-9 / ms))
ms = t(lambda: torch.sum(x)); print('sum    %.1f GB read in %.2f ms = %.2f TB/s read' % (n * 4e-9, ms, n * 4e-9 / ms))

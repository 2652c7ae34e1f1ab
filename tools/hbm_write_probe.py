"""What HBM gives a pure store stream / a copy on this box (torch fill_ / copy_ of 10.7 GB, HIP events): the ceiling the covariance build's
stores are measured against."""
import torch
n = int(10.7e9 / 8)
x = torch.empty(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: x.fill_(1.0)); print("fill_  10.7 GB: %.3f ms = %.2f TB/s written" % (ms, 8 * n / ms / 1e9))
ms = t(lambda: x.zero_()); print("zero_  10.7 GB: %.3f ms = %.2f TB/s written" % (ms, 8 * n / ms / 1e9))
ms = t(lambda: y.copy_(x)); print("copy_  10.7 GB: %.3f ms = %.2f TB/s read + %.2f TB/s written" % (ms, 8 * n / ms / 1e9, 8 * n / ms / 1e9))
ms = t(lambda: torch.sum(x)); print("sum    10.7 GB: %.3f ms = %.2f TB/s read" % (ms, 8 * n / ms / 1e9))

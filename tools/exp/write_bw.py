"""Developer probe: what a pure write stream sustains on this GPU (the floor of conv1_1, which writes 1.11 GB of fp32 planes and
reads 17 MB): torch.Tensor.zero_/fill_ and hipMemsetAsync on 1.11 GB, and a device-to-device copy of the same size."""
import torch

dev = torch.device("cuda", 0)
n = 32 * 369 * 369 * 64
x = torch.empty(n, device=dev)
y = torch.empty(n, device=dev)
for name, fn in (("fill_(1.0)", lambda: x.fill_(1.0)), ("zero_()", lambda: x.zero_()), ("copy_ (read + write)", lambda: y.copy_(x))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%-22s %.3f ms per %.2f GB -> %.2f TB/s written" % (name, ms, n * 4 / 1e9, n * 4 / 1e9 / ms))

"""What a pure-write / copy kernel reaches on this chip (torch fill_ / copy_ on 1 GiB): the ceiling the write-bound
kernels (embed, prior) should be read against."""
import time, torch
n = 256 * 1024 * 1024
a = torch.empty(n, device="cuda", dtype=torch.float32)
b = torch.empty(n, device="cuda", dtype=torch.float32)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for name, fn, bytes_ in (("fill_ (write only)", lambda: a.fill_(1.0), 4 * n), ("zero_ (memset)", lambda: a.zero_(), 4 * n),
                         ("copy_ (read + write)", lambda: b.copy_(a), 8 * n), ("sum (read only)", lambda: a.sum(), 4 * n)):
    s = t(fn)
    print(f"{name:22s} {bytes_ / s / 1e12:.2f} TB/s")

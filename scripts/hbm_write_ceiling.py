"""Reference point for the write-dominated first conv: how fast can this GPU stream writes / copies? (dev helper)"""
import time, torch
x = torch.empty(983_040_000 // 4, dtype=torch.float32, device="cuda")      # the first conv's output per volume: 983 MB
y = torch.empty_like(x)
for name, fn, nbytes in (("fill 983 MB", lambda: x.fill_(1.0), x.numel() * 4), ("copy 983 MB (read + write)", lambda: y.copy_(x), 2 * x.numel() * 4)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {dt*1e3:.3f} ms = {nbytes/dt/1e12:.2f} TB/s")

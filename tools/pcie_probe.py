"""Link probe (run on the GPU box): pinned H2D / D2H rates, both at once, and a strided (2-D) D2H."""
import time
import torch
n = 1 << 30
h_up = torch.empty(n, dtype=torch.uint8).pin_memory()
h_dn = torch.empty(n, dtype=torch.uint8).pin_memory()
d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
up = t(lambda: d_a.copy_(h_up, non_blocking=True))
dn = t(lambda: h_dn.copy_(d_b, non_blocking=True))
def both():
    with torch.cuda.stream(s1): d_a.copy_(h_up, non_blocking=True)
    with torch.cuda.stream(s2): h_dn.copy_(d_b, non_blocking=True)
bo = t(both)
print("H2D %.1f GB/s  D2H %.1f GB/s  both at once: %.1f + %.1f GB/s" % (n / up / 1e9, n / dn / 1e9, n / bo / 1e9, n / bo / 1e9))
# strided D2H: 50000 rows of 12032 of 20000 bytes
d2 = torch.empty((50000, 20000), dtype=torch.uint8, device="cuda")
h2 = torch.empty((50000, 20000), dtype=torch.uint8).pin_memory()
st = t(lambda: h2[:, :12032].copy_(d2[:, :12032], non_blocking=True))
print("2-D D2H 50000 x 12032 of 20000: %.1f GB/s" % (50000 * 12032 / st / 1e9))
pg = torch.empty(n, dtype=torch.uint8)
pu = t(lambda: d_a.copy_(pg), 3)
pd = t(lambda: pg.copy_(d_b), 3)
print("pageable H2D %.1f GB/s  D2H %.1f GB/s" % (n / pu / 1e9, n / pd / 1e9))

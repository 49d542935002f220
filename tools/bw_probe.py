"""Practical HBM bandwidth of this box for streaming kernels of the node-GEMM operand size (torch elementwise ops)."""
import torch, time
M, H = 153600, 256
x = torch.randn(M, H, device="cuda"); y = torch.randn(M, H, device="cuda"); z = torch.empty_like(x)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
nb = x.numel() * 4
for name, f, k in [("copy  (1R+1W)", lambda: z.copy_(x), 2), ("add   (2R+1W)", lambda: torch.add(x, y, out=z), 3),
                   ("sum   (1R)", lambda: x.sum(), 1), ("fill  (1W)", lambda: z.fill_(1.0), 1)]:
    dt = t(f); print(f"{name}: {dt*1e6:8.1f} us  {k*nb/dt/1e12:6.2f} TB/s")
big = torch.randn(8 * M, H, device="cuda"); bz = torch.empty_like(big)
dt = t(lambda: bz.copy_(big)); print(f"copy 1.26 GB (beyond the 256 MB infinity cache): {dt*1e6:8.1f} us  {2*big.numel()*4/dt/1e12:6.2f} TB/s")

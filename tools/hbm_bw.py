import torch, time
for n in (134217728, 268435456):
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): y.copy_(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print(f"copy {n*4/1e6:.0f} MB: {2*n*4/dt/1e9:.0f} GB/s (R+W)")
    for _ in range(3): s = x.sum()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): s = x.sum()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print(f"read-only sum {n*4/1e6:.0f} MB: {n*4/dt/1e9:.0f} GB/s")
    for _ in range(3): torch.add(x, 1.0, out=y)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): torch.add(x, 1.0, out=y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print(f"add scalar {n*4/1e6:.0f} MB: {2*n*4/dt/1e9:.0f} GB/s (R+W)")

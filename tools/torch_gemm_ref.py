"""Reference point: what the vendor fp16 GEMM (torch.matmul -> hipBLASLt/rocBLAS) reaches on the S1 layer shapes."""
import torch
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, M, K, N in [("tdnn2 fwd", 24576, 2560, 512), ("tdnn3 fwd", 23808, 3584, 512), ("tdnn4 fwd", 23808, 512, 512), ("tdnn5 fwd", 23808, 512, 1500),
                      ("tdnn2 wgrad (TN)", 2560, 24576, 512), ("square 8192", 8192, 8192, 8192)]:
    for dt in (torch.float16, torch.float32):
        a = torch.randn(M, K, device="cuda", dtype=dt)
        bt = torch.randn(N, K, device="cuda", dtype=dt)
        us = timeit(lambda: torch.matmul(a, bt.t()))
        print("%-18s %-8s M=%6d K=%6d N=%5d  %8.1f us  %7.1f TF" % (name, str(dt).split(".")[1], M, K, N, us, 2.0 * M * K * N / us / 1e6))

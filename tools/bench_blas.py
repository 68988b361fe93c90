"""What the vendor fp32 GEMM reaches on the head GEMM shapes (reference point for the igemm kernels; not used by the product)."""
import torch, time
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
torch.backends.cuda.matmul.allow_tf32 = False
for M, K, N in [(18432, 4736, 1536), (18432, 1536, 1024), (18432, 1024, 768), (8192, 4608, 512), (32768, 2304, 256), (131072, 1152, 128), (524288, 576, 64), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device="cuda"); b = torch.randn(K, N, device="cuda"); bt = torch.randn(N, K, device="cuda")
    t = timeit(lambda: torch.mm(a, b)); t2 = timeit(lambda: torch.mm(a, bt.t()))
    print("M %6d K %5d N %5d  NN %.3f ms %.1f TF | NT %.3f ms %.1f TF" % (M, K, N, t, 2.0 * M * K * N / t / 1e9, t2, 2.0 * M * K * N / t2 / 1e9), flush=True)

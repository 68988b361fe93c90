"""What the vendor library reaches on the same GEMM shapes (a measuring stick only: the product never calls it)."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, NX, M, N, K in [("l1", 36, 32768, 64, 64), ("l2", 36, 8192, 128, 128), ("l3", 36, 2048, 256, 256), ("l4", 36, 512, 512, 512), ("up1f", 36, 2048, 256, 512),
                          ("head1", 1, 18432, 1536, 4736), ("head2", 1, 18432, 1024, 1536), ("head3", 1, 18432, 768, 1024), ("head4", 1, 18432, 512, 768),
                          ("l1conv_as_gemm", 1, 524288, 64, 576)]:
    a = torch.randn(NX, M, K, device="cuda"); b = torch.randn(NX, N, K, device="cuda")
    ms = t(lambda: torch.bmm(a, b.transpose(1, 2)))
    print("%-15s %2d x [%6d x %4d x %4d]  bmm(NT) %8.1f us  %6.1f TF" % (name, NX, M, N, K, ms * 1e3, 2.0 * NX * M * N * K / ms / 1e9), flush=True)

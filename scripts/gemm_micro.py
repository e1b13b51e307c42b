"""Time the dense products of the north-star step (fp32, N = 1M) under the BLAS back ends torch offers."""
import sys, torch
N = 1_000_000
dev = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x128 = torch.randn(N, 128, device=dev); x256 = torch.randn(N, 256, device=dev); g256 = torch.randn(N, 256, device=dev)
w1 = torch.randn(256, 128, device=dev); w2 = torch.randn(256, 256, device=dev); b = torch.randn(256, device=dev)
w1t = w1.t().contiguous(); w2t = w2.t().contiguous()
for lib in ("hipblaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    print("==", lib)
    for name, fn, gf in [
        ("L1 fwd mm(x128, w1t)", lambda: torch.mm(x128, w1t), 2 * N * 128 * 256),
        ("L1 fwd addmm_act", lambda: torch._addmm_activation(b, x128, w1t), 2 * N * 128 * 256),
        ("L1 fwd linear(x128,w1)", lambda: torch.nn.functional.linear(x128, w1), 2 * N * 128 * 256),
        ("head fwd mm(x256, w2t)", lambda: torch.mm(x256, w2t), 2 * N * 256 * 256),
        ("head fwd linear(x256, w2)", lambda: torch.nn.functional.linear(x256, w2), 2 * N * 256 * 256),
        ("head dx mm(g256, w2)", lambda: torch.mm(g256, w2), 2 * N * 256 * 256),
        ("head dW g^T x (one gemm)", lambda: torch.mm(g256.t(), x256), 2 * N * 256 * 256),
        ("head dW bmm split 128", lambda: torch.bmm(g256.view(128, N // 128, 256).transpose(1, 2), x256.view(128, N // 128, 256)).sum(0), 2 * N * 256 * 256),
        ("head dW bmm split 64", lambda: torch.bmm(g256.view(64, N // 64, 256).transpose(1, 2), x256.view(64, N // 64, 256)).sum(0), 2 * N * 256 * 256),
        ("head dW bmm split 16", lambda: torch.bmm(g256.view(16, N // 16, 256).transpose(1, 2), x256.view(16, N // 16, 256)).sum(0), 2 * N * 256 * 256),
        ("L1 dW bmm split 128", lambda: torch.bmm(g256.view(128, N // 128, 256).transpose(1, 2), x128.view(128, N // 128, 128)).sum(0), 2 * N * 128 * 256),
        ("L1 dW bmm split 64", lambda: torch.bmm(g256.view(64, N // 64, 256).transpose(1, 2), x128.view(64, N // 64, 128)).sum(0), 2 * N * 128 * 256),
        ("L1 dW^T bmm split 64", lambda: torch.bmm(x128.view(64, N // 64, 128).transpose(1, 2), g256.view(64, N // 64, 256)).sum(0), 2 * N * 128 * 256),
        ("head dW^T bmm split 64", lambda: torch.bmm(x256.view(64, N // 64, 256).transpose(1, 2), g256.view(64, N // 64, 256)).sum(0), 2 * N * 256 * 256),
    ]:
        try:
            ms = t(fn)
            print(f"{name:32s} {ms:7.3f} ms  {gf / ms / 1e9:7.1f} TF", flush=True)
        except Exception as e:
            print(name, "failed", repr(e)[:100])

"""Numerical check behind DESIGN.md's next-list item (0): an fp32 product evaluated as six bf16 x bf16 products with fp32
accumulation (operands split exactly into three bf16 terms) against the plain fp32 product, both measured against fp64.
CPU only; emulates the MFMA's arithmetic (exact bf16 products, fp32 sums)."""
import torch

torch.manual_seed(0)


def split3(x):
    a1 = x.to(torch.bfloat16).float()
    a2 = (x - a1).to(torch.bfloat16).float()
    a3 = (x - a1 - a2).to(torch.bfloat16).float()
    return a1, a2, a3


for (n, k, m), scale in (((4096, 128, 256), 1.0), ((4096, 256, 256), 1.0), ((2048, 256, 256), 1e-3)):
    a = torch.randn(n, k) * scale
    b = torch.randn(m, k) / k ** 0.5
    ref = a.double() @ b.double().t()
    plain = a @ b.t()
    A, B = split3(a), split3(b)
    assert torch.equal(A[0] + A[1] + A[2], a) and torch.equal(B[0] + B[1] + B[2], b)  # the split is exact
    terms = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]
    acc = torch.zeros(n, m)
    for i, j in reversed(terms):  # small terms first
        acc = acc + A[i] @ B[j].t()
    den = ref.abs().max()
    e_plain = float((plain.double() - ref).abs().max() / den)
    e_split = float((acc.double() - ref).abs().max() / den)
    acc3 = A[0] @ B[0].t() + A[0] @ B[1].t() + A[1] @ B[0].t()
    e_3 = float((acc3.double() - ref).abs().max() / den)
    print(f"[{n}x{k}]x[{k}x{m}] scale {scale:g}: max err / max|ref|  fp32 {e_plain:.2e}   6 bf16 products {e_split:.2e}   3 products {e_3:.2e}")

"""VectorQuantizer — drop-in for the reference's `quantizer.VectorQuantizer`
(/root/reference/scripts/quantizer.py:6-67; callers scripts/models.py:22,32,50).

Same constructor `(n_e, e_dim, beta)`, same parameter (`embedding.weight`, initialised
U(-1/n_e, 1/n_e) as quantizer.py:22-23), same 5-tuple
`(loss, z_q, perplexity, min_encodings, min_encoding_indices)` (quantizer.py:67), same gradients
(straight-through to z, beta-weighted codebook term — SURVEY.md §3.2).  The arithmetic runs in
hand-written MFMA kernels behind `ceeg_vq_forward` / `ceeg_vq_backward` (include/ceeg.h); the
[B, n_e] distance matrix and one-hot of the reference are never materialised by the kernels.
"""
import torch
import torch.nn as nn

from . import _lib


class _VQFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, weight, beta):
        _lib.require_cuda(z, weight)
        if z.dim() != 2 or weight.dim() != 2 or z.shape[1] != weight.shape[1]:
            raise ValueError(f"z must be [B, e_dim] and the codebook [n_e, e_dim]; got {tuple(z.shape)} and {tuple(weight.shape)}")
        if z.dtype == torch.float32:
            dtype = _lib.CEEG_F32
        elif z.dtype == torch.bfloat16:
            dtype = _lib.CEEG_BF16
        else:
            raise TypeError(f"VectorQuantizer supports float32 and bfloat16 latents, got {z.dtype}")
        if weight.dtype != torch.float32:
            raise TypeError("the codebook (embedding.weight) must be float32")
        L = _lib.lib()
        zc, wc = z.contiguous(), weight.contiguous()
        B, D = zc.shape
        n_e = wc.shape[0]
        dev = zc.device
        z_q = torch.empty_like(zc)
        idx = torch.empty(B, 1, dtype=torch.int64, device=dev)
        scal = torch.empty(2, dtype=torch.float32, device=dev)  # loss, perplexity
        ws = _lib.workspace(L.ceeg_vq_forward_ws_bytes(B, n_e, D, dtype), dev)
        _lib.check(L.ceeg_vq_forward(zc.data_ptr(), wc.data_ptr(), B, n_e, D, dtype, float(beta), z_q.data_ptr(),
                                     idx.data_ptr(), scal[0:].data_ptr(), scal[1:].data_ptr(), None,
                                     ws.data_ptr(), ws.numel(), _lib.stream()), "ceeg_vq_forward")
        ctx.save_for_backward(zc, wc, idx)
        ctx.beta, ctx.dtype = float(beta), dtype
        loss, ppl = scal[0], scal[1]
        ctx.mark_non_differentiable(ppl, idx)
        ctx.set_materialize_grads(False)  # an undifferentiated z_q (or loss) comes back as None, not as a [B, D] tensor of zeros
        return loss, z_q, ppl, idx

    @staticmethod
    def backward(ctx, g_loss, g_zq, _g_ppl, _g_idx):
        zc, wc, idx = ctx.saved_tensors
        L = _lib.lib()
        B, D = zc.shape
        n_e = wc.shape[0]
        dev = zc.device
        need_z, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        s = torch.zeros(1, dtype=torch.float32, device=dev) if g_loss is None else g_loss.reshape(1).to(torch.float32).contiguous()
        g = None if g_zq is None else g_zq.contiguous()
        dz = torch.empty_like(zc) if need_z else None
        dW = torch.empty_like(wc) if need_w else None
        ws = _lib.workspace(L.ceeg_vq_backward_ws_bytes(B, n_e, D, ctx.dtype), dev) if need_w else None
        _lib.check(L.ceeg_vq_backward(zc.data_ptr(), wc.data_ptr(), idx.data_ptr(), _lib.ptr(g), s.data_ptr(), B, n_e, D,
                                      ctx.dtype, ctx.beta, _lib.ptr(dz), _lib.ptr(dW), _lib.ptr(ws),
                                      0 if ws is None else ws.numel(), _lib.stream()), "ceeg_vq_backward")
        return dz, dW, None


class VectorQuantizer(nn.Module):
    """EA-VQ-VAE discretisation bottleneck (reference: scripts/quantizer.py:6-67).

    Inputs:
    - n_e : number of embeddings
    - e_dim : dimension of embedding
    - beta : weight of the codebook term of the loss (placement as in quantizer.py:57-58)
    - return_one_hot : build the dense [B, n_e] `min_encodings` tensor the reference returns
      (both reference callers discard it, scripts/models.py:32,50); False returns None in its slot and
      saves B*n_e*4 bytes of HBM traffic.
    """

    def __init__(self, n_e, e_dim, beta, return_one_hot=True):
        super().__init__()
        self.n_e = n_e
        self.e_dim = e_dim
        self.beta = beta
        self.return_one_hot = return_one_hot
        self.embedding = nn.Embedding(self.n_e, self.e_dim)
        self.embedding.weight.data.uniform_(-1.0 / self.n_e, 1.0 / self.n_e)

    MAX_N_E, MAX_E_DIM = 4096, 256   # what libceeg's kernels are tested for (include/ceeg.h, ceeg_vq_forward)

    def forward(self, z):
        if self.n_e > self.MAX_N_E or self.e_dim > self.MAX_E_DIM:
            raise _lib.CeegError(f"VectorQuantizer: n_e = {self.n_e}, e_dim = {self.e_dim} is outside what the HIP kernels take "
                                 f"(n_e <= {self.MAX_N_E}, e_dim <= {self.MAX_E_DIM}); there is no fallback path")
        loss, z_q, perplexity, min_encoding_indices = _VQFunction.apply(z, self.embedding.weight, self.beta)
        min_encodings = None
        if self.return_one_hot:
            min_encodings = torch.zeros(min_encoding_indices.shape[0], self.n_e, device=z.device)
            min_encodings.scatter_(1, min_encoding_indices, 1)
        return loss, z_q, perplexity, min_encodings, min_encoding_indices

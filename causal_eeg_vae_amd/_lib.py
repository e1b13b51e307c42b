"""ctypes binding of libceeg.so (include/ceeg.h) — the only way the package reaches the GPU kernels.

There is NO CPU fallback: if the shared library is missing or the call fails, this raises.  PyTorch is
used only for device memory (tensors own every buffer, workspaces included) and the current stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libceeg.so")

CEEG_F32, CEEG_BF16 = 0, 1

_c_i64, _c_int, _c_f32, _c_u64, _c_sz, _vp = C.c_int64, C.c_int, C.c_float, C.c_uint64, C.c_size_t, C.c_void_p

# name -> (restype, argtypes); mirrors include/ceeg.h one to one
SIGNATURES = {
    "ceeg_version": (_c_int, []),
    "ceeg_source_hash": (C.c_char_p, []),
    "ceeg_strerror": (C.c_char_p, [_c_int]),
    "ceeg_check_device": (_c_int, []),
    "ceeg_vq_forward_ws_bytes": (_c_sz, [_c_i64, _c_int, _c_int, _c_int]),
    "ceeg_vq_forward": (_c_int, [_vp, _vp, _c_i64, _c_int, _c_int, _c_int, _c_f32, _vp, _vp, _vp, _vp, _vp, _vp, _c_sz, _vp]),
    "ceeg_vq_forward_static_max_bytes": (_c_sz, []),
    "ceeg_vq_forward_static_prepare": (_c_int, []),
    "ceeg_vq_forward_screened": (_c_int, [_vp, _c_sz, _c_i64, _c_int, _c_int, _c_int, _vp, _vp]),
    "ceeg_vq_forward_static": (_c_int, [_vp, _vp, _c_i64, _c_int, _c_int, _c_int, _c_f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ceeg_vq_backward_ws_bytes": (_c_sz, [_c_i64, _c_int, _c_int, _c_int]),
    "ceeg_vq_backward": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_i64, _c_int, _c_int, _c_int, _c_f32, _vp, _vp, _vp, _c_sz, _vp]),
    "ceeg_csr_build_ws_bytes": (_c_sz, [_c_i64, _c_i64, _c_int]),
    "ceeg_csr_build_blocks_supported": (_c_int, [_c_i64, _c_i64]),
    "ceeg_csr_build_blocks_ws_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64]),
    "ceeg_csr_build_blocks": (_c_int, [_vp, _vp, _c_i64, _c_i64, _vp, _vp, _c_i64, _c_i64, _vp, _vp, _vp, _vp, _vp, _vp, _c_sz, _vp]),
    "ceeg_csr_build": (_c_int, [_vp, _vp, _c_i64, _c_i64, _c_int, _vp, _vp, _vp, _vp, _vp, _c_sz, _vp]),
    "ceeg_csr_transpose_ws_bytes": (_c_sz, [_c_i64, _c_i64]),
    "ceeg_csr_transpose": (_c_int, [_vp, _vp, _c_i64, _c_i64, _vp, _vp, _vp, _vp, _c_sz, _vp]),
    "ceeg_gcn_norm": (_c_int, [_vp, _vp, _vp, _vp, _c_i64, _c_i64, _vp, _vp, _vp]),
    "ceeg_gcn_norm_t": (_c_int, [_vp, _vp, _vp, _vp, _c_i64, _c_i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ceeg_csr_long_items_ws_bytes": (_c_sz, [_c_i64]),
    "ceeg_csr_long_items": (_c_int, [_vp, _c_i64, _c_int, _vp, _c_i64, _vp, _vp, _c_sz, _vp]),
    "ceeg_csr_add_self_loops_ws_bytes": (_c_sz, [_c_i64]),
    "ceeg_csr_add_self_loops": (_c_int, [_vp] * 6 + [_c_i64] + [_vp] * 6 + [_vp, _c_sz, _vp]),
    "ceeg_spmm_ws_bytes": (_c_sz, [_c_i64, _c_i64, _c_int]),
    "ceeg_spmm_csr": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_i64, _c_i64, _c_i64, _c_int, _vp, _c_int, _c_int, _vp, _c_i64, _c_i64, _vp, _c_sz, _vp]),
    "ceeg_spmm_csr_items": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_i64, _c_i64, _c_i64, _c_int, _vp, _c_int, _c_int, _vp, _c_i64, _c_i64, _vp, _c_i64, _vp, _c_sz, _vp]),
    "ceeg_spmm_csr_masked": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_i64, _c_i64, _c_i64, _c_int, _vp, _c_i64, _vp, _c_i64, _c_i64, _vp, _c_i64, _vp, _c_sz, _vp]),
    "ceeg_spmm_csr_cv": (_c_int, [_vp, _vp, _vp, _c_i64, _c_i64, _c_i64, _c_int, _vp, _c_int, _vp, _c_i64, _vp, _c_i64, _c_i64, _vp, _c_i64, _vp, _c_sz, _vp]),
    "ceeg_spmm_csr_cv_bf16": (_c_int, [_vp, _vp, _vp, _c_i64, _c_i64, _c_i64, _c_int, _vp, _c_int, _vp, _c_i64, _vp, _c_i64, _c_i64, _vp, _c_i64, _vp, _c_sz, _vp]),
    "ceeg_spmm_gather_probe": (_c_int, [_vp, _vp, _vp, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_int, _vp, _c_sz, _vp]),
    "ceeg_reparam_kl_ws_bytes": (_c_sz, [_c_i64, _c_int]),
    "ceeg_reparam_kl_fwd": (_c_int, [_vp, _c_i64, _vp, _c_i64, _vp, _c_u64, _c_i64, _c_int, _c_f32, _c_int, _c_int, _vp, _vp, _vp, _c_sz, _vp]),
    "ceeg_reparam_kl_bwd": (_c_int, [_vp, _c_i64, _vp, _c_i64, _vp, _c_u64, _vp, _vp, _c_i64, _c_int, _c_f32, _c_int, _c_int, _vp, _c_i64, _vp, _c_i64, _vp]),
    "ceeg_reparam_vq_bwd_ws_bytes": (_c_sz, [_c_i64, _c_int]),
    "ceeg_reparam_vq_bwd": (_c_int, [_vp, _c_i64, _vp, _c_i64, _vp, _c_u64, _vp, _vp, _vp, _vp, _c_int, _vp, _vp, _c_i64, _c_int, _c_f32, _c_int,
                                     _c_int, _vp, _c_i64, _vp, _c_i64, _vp, _vp, _c_sz, _vp]),
    "ceeg_edge_bce_ws_bytes": (_c_sz, [_c_i64, _c_int]),
    "ceeg_edge_bce_fwd": (_c_int, [_vp, _c_i64, _vp, _vp, _vp, _c_i64, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_sz, _vp]),
    "ceeg_decoder_fused_ws_bytes": (_c_sz, [_c_i64, _c_i64, _c_int]),
    "ceeg_decoder_fused_fwd": (_c_int, [_vp, _c_i64, _vp, _vp, _vp, _c_i64, _c_i64, _c_int, _c_i64, _c_i64, _c_i64, _vp, _vp, _c_int, _vp, _vp, _c_i64, _vp, _c_i64, _vp, _c_sz, _vp]),
    "ceeg_decoder_fused_bwd": (_c_int, [_vp, _c_i64, _vp, _vp, _vp, _vp, _vp, _vp, _c_i64, _c_i64, _c_int, _c_i64, _c_i64, _c_i64, _c_i64, _vp, _vp, _c_int, _vp, _vp, _c_i64, _vp, _vp, _c_i64, _c_i64, _c_i64, _vp, _c_sz, _vp]),
    "ceeg_pair_bce_codes_supported": (_c_int, [_c_int, _c_int, _c_i64, _c_i64]),
    "ceeg_pair_bce_codes_ws_bytes": (_c_sz, [_c_i64, _c_i64, _c_int]),
    "ceeg_pair_bce_codes_fwd": (_c_int, [_vp, _c_i64, _c_int, _c_int, _vp, _vp, _vp, _c_i64, _c_i64, _c_int, _vp, _vp, _vp, _vp, _c_sz, _vp]),
    "ceeg_pair_bce_codes_bwd": (_c_int, [_vp, _c_i64, _c_int, _c_int, _vp, _vp, _vp, _c_i64, _c_i64, _vp, _c_int, _vp, _c_i64, _vp, _c_sz, _vp]),
    "ceeg_edge_bce_codes_ws_bytes": (_c_sz, [_c_i64, _c_int]),
    "ceeg_edge_bce_codes_fwd": (_c_int, [_vp, _c_i64, _c_int, _c_int, _vp, _vp, _vp, _c_i64, _c_int, _vp, _vp, _vp, _c_sz, _vp]),
    "ceeg_spmm_csr_table": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_i64, _vp, _c_i64, _c_i64, _c_int, _c_int, _vp, _c_i64, _c_i64, _vp, _c_sz, _vp]),
    "ceeg_neg_sample_pair_bce_codes_fwd": (_c_int, [_vp, _c_i64, _c_int, _c_int, _vp, _vp, _vp, _c_i64, _c_i64, _c_u64, _vp, _c_int, _vp, _vp, _vp, _vp,
                                                    _vp, _vp, _c_sz, _vp]),
    "ceeg_decoder_codes_bwd_pairs_ws_bytes": (_c_sz, [_c_i64, _c_i64, _c_int, _c_int, _c_i64, _c_i64, _c_i64]),
    "ceeg_decoder_codes_bwd_pairs": (_c_int, [_vp, _c_i64, _c_int, _c_int, _vp, _vp, _vp, _c_i64, _c_i64, _vp, _c_i64, _vp, _c_i64, _c_i64, _c_i64,
                                              _vp, _vp, _vp, _vp, _c_i64, _vp, _vp, _c_i64, _vp, _c_sz, _vp]),
    "ceeg_decoder_codes_chunk": (_c_int, []),
    "ceeg_decoder_codes_supported": (_c_int, [_c_int, _c_int]),
    "ceeg_decoder_codes_ws_bytes": (_c_sz, [_c_i64, _c_i64, _c_int, _c_int, _c_i64, _c_i64]),
    "ceeg_decoder_codes_fwd": (_c_int, [_vp, _c_i64, _c_int, _c_int, _vp, _vp, _vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _vp, _c_i64, _c_i64, _vp, _vp, _vp, _vp, _c_sz, _vp]),
    "ceeg_decoder_codes_bwd": (_c_int, [_vp, _c_i64, _c_int, _c_int, _vp, _vp, _vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _vp, _c_i64, _vp, _c_i64, _c_i64, _c_i64, _vp, _vp, _vp, _c_i64, _vp, _vp, _c_sz, _vp]),
    "ceeg_row_sqnorm": (_c_int, [_vp, _c_i64, _c_i64, _c_int, _vp, _vp]),
    "ceeg_concat_condition": (_c_int, [_vp, _c_i64, _c_i64, _c_int, _vp, _c_int, _c_i64, _vp, _vp, _vp]),
    "ceeg_bias_relu": (_c_int, [_vp, _c_i64, _c_i64, _c_int, _vp, _c_int, _vp]),
    "ceeg_colsum_ws_bytes": (_c_sz, [_c_i64, _c_int]),
    "ceeg_relu_bwd_colsum": (_c_int, [_vp, _c_i64, _vp, _c_i64, _vp, _c_i64, _c_i64, _c_int, _vp, _vp, _c_sz, _vp]),
    "ceeg_loss_sum": (_c_int, [_vp, _vp, _vp, _c_f32, _vp, _vp, _vp]),
    "ceeg_recon_finish": (_c_int, [_vp, _vp, _vp, _c_f32, _vp, _vp]),
    "ceeg_colsum": (_c_int, [_vp, _c_i64, _c_i64, _c_int, _vp, _vp, _c_sz, _vp]),
    "ceeg_gat_softmax_fwd": (_c_int, [_vp, _vp, _vp, _vp, _c_i64, _c_i64, _c_f32, _vp, _vp]),
    "ceeg_gat_attn_bwd": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _c_i64, _vp, _c_i64, _c_i64, _c_i64, _c_int, _c_f32, _vp, _vp, _vp]),
    "ceeg_segment_sum_perm": (_c_int, [_vp, _vp, _vp, _c_i64, _vp, _vp]),
    "ceeg_segment_sum_items_ws_bytes": (_c_sz, [_c_i64]),
    "ceeg_segment_sum_perm_items": (_c_int, [_vp, _vp, _vp, _c_i64, _vp, _c_i64, _c_int, _vp, _vp, _c_sz, _vp]),
    "ceeg_gat_scores_fwd": (_c_int, [_vp, _c_i64, _vp, _vp, _c_i64, _c_int, _c_int, _vp, _vp, _vp]),
    "ceeg_gat_scores_bwd_ws_bytes": (_c_sz, [_c_i64, _c_int, _c_int]),
    "ceeg_gat_scores_bwd": (_c_int, [_vp, _c_i64, _vp, _vp, _vp, _vp, _c_i64, _c_int, _c_int, _vp, _c_i64, _vp, _vp, _vp, _c_sz, _vp]),
    "ceeg_graph_layernorm_ws_bytes": (_c_sz, [_c_i64, _c_int, _c_i64]),
    "ceeg_graph_layernorm_fwd": (_c_int, [_vp, _c_i64, _vp, _c_i64, _c_int, _c_i64, _vp, _vp, _c_f32, _c_int, _vp, _c_i64, _vp, _vp, _vp, _c_sz, _vp]),
    "ceeg_graph_layernorm_bwd": (_c_int, [_vp, _c_i64, _vp, _c_i64, _vp, _c_i64, _vp, _c_i64, _c_int, _c_i64, _vp, _vp, _vp, _c_int, _vp, _c_i64,
                                          _vp, _vp, _vp, _c_sz, _vp]),
    "ceeg_segment_ids": (_c_int, [_vp, _c_i64, _c_i64, _vp, _vp]),
    "ceeg_collate_edges": (_c_int, [_vp, _c_i64, _vp, _vp, _c_i64, _c_i64, _c_int, _vp, _c_i64, _vp, _vp]),
    "ceeg_collate_edges_checked": (_c_int, [_vp, _c_i64, _vp, _vp, _c_i64, _c_i64, _vp, _c_i64, _vp, _vp, _vp]),
    "ceeg_top_predecessors": (_c_int, [_vp, _vp, _vp, _vp, _c_i64, _c_int, _c_i64, _vp, _vp]),
    "ceeg_ancestors_cap": (_c_int, [_c_int, _c_int]),
    "ceeg_ancestors": (_c_int, [_vp, _c_i64, _c_int, _vp, _c_i64, _c_int, _c_int, _vp, _vp, _vp]),
    "ceeg_ancestor_subgraph_count": (_c_int, [_vp, _vp, _vp, _vp, _c_i64, _c_int, _vp, _vp]),
    "ceeg_ancestor_subgraph_fill": (_c_int, [_vp, _vp, _vp, _vp, _c_i64, _c_int, _vp, _vp, _vp, _vp]),
    "ceeg_token_pool_fwd": (_c_int, [_vp, _c_i64, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp]),
    "ceeg_token_pool_bwd": (_c_int, [_vp, _c_i64, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ceeg_neg_sample_uniform": (_c_int, [_vp, _vp, _c_i64, _c_i64, _c_u64, _vp, _vp, _vp]),
    "ceeg_neg_filter_bits": (_c_int, [_c_i64]),
    "ceeg_neg_filter_build": (_c_int, [_vp, _vp, _c_i64, _c_int, _vp, _vp]),
    "ceeg_neg_sample_uniform_filtered": (_c_int, [_vp, _vp, _c_i64, _c_i64, _c_u64, _vp, _c_int, _vp, _vp, _vp]),
}

_lib = None


class CeegError(RuntimeError):
    pass


def lib():
    """Load libceeg.so (once).  Raises CeegError when it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CeegError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(make -C causal_eeg_vae_amd/csrc).  The HIP extension is required; there is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check_build_matches_sources():
    """Raise CeegError when libceeg.so was built from other sources than the ones next to it (a stale binary): the hash the
    Makefile baked in (ceeg_source_hash) against the hash of csrc/*.hip, csrc/*.h and include/ceeg.h in the tree."""
    from ._srchash import source_hash
    built, tree = lib().ceeg_source_hash().decode(), source_hash()
    if built != tree:
        raise CeegError(f"libceeg.so is stale: built from sources {built}, the tree holds {tree}; rebuild (make -C causal_eeg_vae_amd/csrc)")
    return built


def check(rc, what):
    if rc != 0:
        raise CeegError(f"{what} failed: {lib().ceeg_strerror(rc).decode()} (code {rc})")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous where the ABI says so."""
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def workspace(nbytes, device):
    """Caller-owned workspace: a uint8 tensor from torch's caching allocator (256-byte aligned)."""
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


_TUNED_GEMMS = None


def enable_tuned_gemms(path=None):
    """The dense products of the step stay library GEMMs (torch.mm / addmm / bmm -> hipBLASLt / rocBLAS); this selects, for
    exactly the shapes of BASELINE's configs, the library kernel a PyTorch TunableOp sweep on an MI355X found fastest
    (tuned_gemm_gfx950.csv next to this file: lookup only, nothing is tuned or written at run time; shapes that are not in
    the file keep the library's own heuristic).  The file carries validators (PyTorch / HIP / hipBLASLt / rocBLAS versions,
    gfx950): on any other stack PyTorch rejects it and the default kernels run.  CEEG_TUNED_GEMM=0 switches it off.
    Returns True when the table is active.  Called by Trainer and bench.py; process-wide PyTorch state."""
    global _TUNED_GEMMS
    if _TUNED_GEMMS is not None:
        return _TUNED_GEMMS
    _TUNED_GEMMS = False
    if os.environ.get("CEEG_TUNED_GEMM", "1") != "1" or not torch.cuda.is_available():
        return False
    try:
        import torch.cuda.tunable as tn
        if tn.is_enabled():          # the application manages TunableOp itself: leave it alone
            return False
        path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gemm_gfx950.csv")
        if not os.path.exists(path):
            return False
        tn.enable(True)
        tn.tuning_enable(False)
        _TUNED_GEMMS = bool(tn.read_file(path))
        if not _TUNED_GEMMS:
            tn.enable(False)
    except Exception:               # an older / different PyTorch: the default kernels run
        _TUNED_GEMMS = False
    return _TUNED_GEMMS


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise CeegError("causal_eeg_vae_amd runs on MI355X (ROCm) tensors only; got a CPU tensor. "
                            "There is no CPU fallback in the product path (the CPU oracle lives under oracle/ for tests).")


# ---------------------------------------------------------------------------------------------------
# Optional per-call HIP-event timing (bench.py): PROFILE = {} switches it on; entries are
# name -> [(start_event, end_event), ...] recorded on the stream the kernels are launched on.
PROFILE = None


class timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if PROFILE is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.e.record()
            PROFILE.setdefault(self.name, []).append((self.s, self.e))
        return False


def profile_summary():
    """name -> (launches, mean ms).  Call after torch.cuda.synchronize()."""
    out = {}
    for k, evs in (PROFILE or {}).items():
        ms = [s.elapsed_time(e) for s, e in evs]
        out[k] = (len(ms), sum(ms) / max(len(ms), 1))
    return out

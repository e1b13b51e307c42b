/*
 * ceeg.h — C-ABI of libceeg.so: the MI355X (gfx950) hot path of the VGAE + VQ training step.
 *
 * The reference (bendavidsteel/causal-eeg-vae) exposes NO FFI/operator interface: its boundary is
 * the Python nn.Module call convention.  Each entry point below names the
 * reference (or upstream-PyG) call it stands behind.  Citations are relative to /root/reference.
 *
 * Conventions (all entry points):
 *   - plain C types only: raw DEVICE pointers + explicit sizes; contiguous row-major tensors;
 *   - the CALLER owns every buffer including `ws` (workspace).  Nothing here calls hipMalloc/hipFree.
 *     Workspace sizes come from the matching ceeg_*_ws_bytes() query (pure host arithmetic);
 *   - stream-ordered on `stream` (a hipStream_t passed as void*; NULL = the null stream);
 *     no host synchronisation inside, re-entrant, no global mutable state;
 *   - returns CEEG_OK (0) or a negative CEEG_E* code; ceeg_strerror() gives the text.  Host-side
 *     argument/shape validation happens BEFORE any launch, so a bad call never faults the GPU;
 *   - results are bitwise reproducible run to run: no float atomics anywhere.
 */
#ifndef CEEG_H
#define CEEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CEEG_VERSION 100 /* 0.1.0 */

enum {
  CEEG_OK = 0,
  CEEG_EINVAL = -1,       /* null pointer / negative size / misaligned pointer */
  CEEG_EUNSUPPORTED = -2, /* shape or dtype outside what the kernels implement */
  CEEG_EWORKSPACE = -3,   /* ws_bytes smaller than the ws_bytes() query */
  CEEG_ELAUNCH = -4,      /* hipGetLastError() != hipSuccess after a launch */
  CEEG_ENODEVICE = -5,    /* no HIP device / wrong architecture */
  CEEG_ETOOBIG = -6       /* sizes beyond the fixed scratch of an entry point that takes no workspace */
};

/* element types of z / z_q / features */
enum { CEEG_F32 = 0, CEEG_BF16 = 1 };

int ceeg_version(void);
/* Hash of the sources this binary was built from (the .hip and .h files under csrc/ and this header; causal_eeg_vae_amd/_srchash.py computes
 * the same value for a source tree).  No reference counterpart: it lets smoke() / the tests refuse a stale libceeg.so. */
const char* ceeg_source_hash(void);
const char* ceeg_strerror(int code);
/* Reads hipGetDeviceProperties of the current device: returns CEEG_OK on gfx950, else CEEG_ENODEVICE. */
int ceeg_check_device(void);

/* ------------------------------------------------------------------------------------------------
 * A2  VectorQuantizer.forward — scripts/quantizer.py:25-67 (callers scripts/models.py:32,50)
 *
 *   d[b,c] = fl(fl(zz[b] + ee[c]) - 2*dot[b,c])            (quantizer.py:43-45; op order kept)
 *   idx[b] = argmin_c d[b,c], LOWEST index on exact ties   (quantizer.py:48)
 *   z_q[b] = W[idx[b]]                                     (quantizer.py:49-54; one-hot never built)
 *   sqerr  = sum_b sum_k (z_q[b,k]-z[b,k])^2               (quantizer.py:57-58, both terms share it)
 *   hist[c]= #{b : idx[b]==c}                              (quantizer.py:64)
 *   loss = (1+beta)*sqerr/(B*D); perplexity = exp(-sum p log(p+1e-10)), p = hist/B  (:57-58,:64-65)
 *
 * dtype CEEG_F32 : z,z_q fp32; distance GEMM on v_mfma_f32_16x16x4_f32.  dot is the k-ascending
 *                  fmaf chain from 0 (bit-identical to oracle/ceeg_oracle.c:vq_forward_f32).
 * dtype CEEG_BF16: z,z_q bf16 (codebook stays fp32, rounded RNE to bf16 for the GEMM);
 *                  v_mfma_f32_16x16x32_bf16, fp32 accumulate; score = dot - ee/2 (row-constant zz
 *                  dropped); no reference counterpart.  z_q[b] = bf16(W[idx[b]]) and
 *                  sqerr is taken against that returned row.  Two kernels, bitwise-equal idx and z_q:
 *                  D in {64,128,256} with n_e padded to 128/256/512 runs with the codebook resident in
 *                  registers (persistent, one workgroup per CU); every other shape with the rows in
 *                  registers.  Environment CEEG_VQ_BF16_BS=0 forces the second one (a checker's switch).
 *
 * z [B,D], W [n_e,D] fp32, z_q [B,D] (dtype), idx [B] int64, loss/perplexity: 1 float each (device),
 * hist [n_e] uint32 (device, may be NULL).  Limits: 1 <= D <= 256, 1 <= n_e <= 4096 (the largest codebook the tests exercise), B >= 1
 * (B == 0 is CEEG_EINVAL: the reference divides by zero there).
 * ----------------------------------------------------------------------------------------------*/
size_t ceeg_vq_forward_ws_bytes(int64_t B, int n_e, int D, int dtype);
int ceeg_vq_forward(const void* z, const float* W, int64_t B, int n_e, int D, int dtype, float beta,
                    void* z_q, int64_t* idx, float* loss, float* perplexity, uint32_t* hist,
                    void* ws, size_t ws_bytes, void* stream);

/* ceeg_vq_forward WITHOUT a workspace argument and without any allocation.  No counterpart in the reference: it exists
 * for callers that cannot allocate device memory around the call - code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  Same kernels, same results bit for bit as ceeg_vq_forward; the scratch the kernels
 * need (codebook image, code norms, per-block error partials, per-slice histograms) is a fixed array inside the library,
 * ceeg_vq_forward_static_max_bytes() = 4 MiB per GPU the library runs on, held until the process ends.  A problem whose
 * ceeg_vq_forward_ws_bytes() exceeds that (e.g. n_e = 4096 with D = 256 in fp32: a 4 MiB codebook image alone) returns CEEG_ETOOBIG before any
 * launch; config 3 (4 Mi x 512 x 256, 0.63 MiB in bf16, 0.88 MiB in fp32) fits.  Calls no hipMalloc/hipFree or allocating routine.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads) RACE - serialise
 * them.  Nothing in the array has to be zero between calls (every word a kernel reads is written earlier in the same
 * call).  The one exception to "no global mutable state" above: the per-device address of the array, looked up once.
 * Set-up: the array's device address is looked up (hipGetSymbolAddress, under a mutex) on the FIRST call on each device; that
 * lookup may load the code object and is not capture-safe.  Call ceeg_vq_forward_static_prepare() once per device, with that device
 * current, outside any stream capture (or make the first ceeg_vq_forward_static call there); calls after it only launch kernels. */
/* fp32 forwards whose bf16 codebook fits 128 KB of LDS (n_e * D * 2 bytes, D a power of two >= 32) run in a SCREENED form: bf16 MFMA
 * scores of all codes with a rigorous error bound pick the few codes that can win a row, and only those get the arithmetic above -
 * same codes, same z_q, bit for bit (csrc/vq.hip, vq_screen_f32); whatever the bound cannot decide (near-tie codebooks, NaN / inf)
 * falls to the full kernel on the device, without a host decision.  CEEG_VQ_SCREEN=0 keeps the full kernel alone.  This helper (no
 * counterpart in the reference; tests and measurements) reports which of the two decided the last forward on workspace `ws`:
 * *out = 1 screened, 0 full kernel, -1 the call does not screen.  It synchronises the stream. */
int ceeg_vq_forward_screened(const void* ws, size_t ws_bytes, int64_t B, int n_e, int D, int dtype, int* out, void* stream);
int ceeg_vq_forward_static_prepare(void);
size_t ceeg_vq_forward_static_max_bytes(void);
int ceeg_vq_forward_static(const void* z, const float* W, int64_t B, int n_e, int D, int dtype, float beta,
                           void* z_q, int64_t* idx, float* loss, float* perplexity, uint32_t* hist, void* stream);

/* Backward of A2 (closed form; autograd of quantizer.py:54-61):
 *   dz[b]  = g[b] + s*(2/(B*D))*(z[b]-W[idx[b]])              g = dL/dz_q (NULL = 0), s = dL/dloss
 *   dW[c]  = s*beta*(2/(B*D)) * sum_{b: idx[b]==c} (W[c]-z[b])   (deterministic segmented sum)
 * s is read from DEVICE memory (*s_dev, 1 float) so autograd never syncs.  dz (dtype) and/or dW (fp32)
 * may be NULL to skip. */
size_t ceeg_vq_backward_ws_bytes(int64_t B, int n_e, int D, int dtype);
int ceeg_vq_backward(const void* z, const float* W, const int64_t* idx, const void* g, const float* s_dev,
                     int64_t B, int n_e, int D, int dtype, float beta, void* dz, float* dW,
                     void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * A5  GCNConv / gcn_norm — upstream torch_geometric (un-vendored, unpinned: environment.yml:13);
 *     edge direction old_id -> new_id as built at scripts/dataset.py:147-153,217-224.
 *
 * ceeg_csr_build: edge list (src[e] -> dst[e], int64, PyG edge_index rows 0 and 1) -> CSR grouped by
 *   dst, entries ordered by (dst, src, original edge id): THE canonical edge order.  With
 *   self_loops=1 it applies add_remaining_self_loops: every existing self-loop is dropped and exactly
 *   one (i,i) entry per node is added (its perm is the LAST dropped self-loop's edge id, whose weight
 *   PyG keeps, or -1 when the node had none).  Outputs:
 *     rowptr [N+1] int32, col [nnz] int32 (source node), perm [nnz] int32 (original edge id; see above for
 *     self-loops), nnz_out: 1 int64 on the DEVICE.  nnz upper bound = E + (self_loops ? N : 0):
 *     col/perm must have that capacity.
 * ceeg_csr_transpose: CSR (rows=dst) -> CSR of the transpose (rows=src), entries ordered by
 *   (src, dst, position); tperm[q] = position p in the input CSR, so values permute as val_t[q]=val[tperm[q]].
 *   `nnz` may be an upper bound (the capacity of col / t_col / t_perm) when the true count rowptr[N] is still on the device.
 * ceeg_gcn_norm: val[p] = dinv[col[p]] * w[p] * dinv[row(p)], deg[i] = sum_{p in row i} w[p]
 *   (in-degree at the target incl. the self-loop), dinv = deg^-1/2 (0 where deg == 0).  w == NULL means
 *   all ones; else w[p] = ew[perm[p]] for real edges and 1 for added self-loops (ew: [E] fp32).
 * ----------------------------------------------------------------------------------------------*/
size_t ceeg_csr_build_ws_bytes(int64_t E, int64_t N, int self_loops);
int ceeg_csr_build(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int self_loops,
                   int32_t* rowptr, int32_t* col, int32_t* perm, int64_t* nnz_out,
                   void* ws, size_t ws_bytes, void* stream);
/* ceeg_csr_build for a COLLATED batch (PyG Batch.from_data_list as scripts/run.py:310 collates: block-diagonal, every edge inside its
 * graph's block of node ids): batch [N] int64 = graph of every node, node_ptr [G+1] int64 = first node of every graph,
 * max_block_nodes >= the largest graph.  The sort key holds the source relative to its graph, bits(N) + bits(max_block_nodes) <= 32
 * (ceeg_csr_build_blocks_supported says 1): a 32-bit key-value sort instead of a 64-bit one.  Same rowptr / col / perm as
 * ceeg_csr_build(self_loops = 0), bit for bit.  *outside (device int32) is set to 1 when an edge is NOT inside one graph's block - such an
 * edge is dropped and the caller must build with ceeg_csr_build instead - else 0. */
int ceeg_csr_build_blocks_supported(int64_t N, int64_t max_block_nodes);
size_t ceeg_csr_build_blocks_ws_bytes(int64_t E, int64_t N, int64_t max_block_nodes);
int ceeg_csr_build_blocks(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, const int64_t* batch, const int64_t* node_ptr,
                          int64_t G, int64_t max_block_nodes, int32_t* rowptr, int32_t* col, int32_t* perm, int64_t* nnz_out,
                          int32_t* outside, void* ws, size_t ws_bytes, void* stream);
size_t ceeg_csr_transpose_ws_bytes(int64_t nnz, int64_t N);
int ceeg_csr_transpose(const int32_t* rowptr, const int32_t* col, int64_t nnz, int64_t N,
                       int32_t* t_rowptr, int32_t* t_col, int32_t* t_perm,
                       void* ws, size_t ws_bytes, void* stream);
int ceeg_gcn_norm(const int32_t* rowptr, const int32_t* col, const int32_t* perm, const float* ew,
                  int64_t nnz, int64_t N, float* deg_inv_sqrt, float* val, void* stream);
/* ceeg_gcn_norm followed by the values in the transposed structure's order, tval[q] = val[t_perm[q]] for q < t_rowptr[N] (what the
 * backward aggregation streams); `nnz` may be a capacity as above. */
int ceeg_gcn_norm_t(const int32_t* rowptr, const int32_t* col, const int32_t* perm, const float* ew, int64_t nnz, int64_t N,
                    float* deg_inv_sqrt, float* val, const int32_t* t_rowptr, const int32_t* t_perm, float* tval, void* stream);
/* What host code must know about a freshly built CSR before it can launch on it, produced on the DEVICE in one small record so
 * that a graph built inside the training loop (a new PyG minibatch every step, scripts/run.py:310-326) costs ONE host read:
 *   record[0] = rowptr[N] (the entry count; ceeg_csr_transpose and ceeg_gcn_norm accept a capacity >= it as their `nnz`, the
 *               array tails beyond it are never read), record[1] = the longest row, record[2] = number of pieces;
 *   items     = the piece list of the rows longer than `chunk` entries in the layout of ceeg_spmm_csr_items /
 *               ceeg_decoder_codes_*: int32 [n][4] = (lo, hi, chunk_index*2 + slot, row), rows and pieces ascending; at most
 *               items_cap are written (3 * nnz / chunk always suffices; record[2] > items_cap tells the host it did not).
 * record: int64 [3] on the device.  No counterpart in the reference (PyG's collation keeps its slice_dict on the host). */
size_t ceeg_csr_long_items_ws_bytes(int64_t N);
int ceeg_csr_long_items(const int32_t* rowptr, int64_t N, int chunk, int32_t* items, int64_t items_cap, int64_t* record,
                        void* ws, size_t ws_bytes, void* stream);
/* add_remaining_self_loops on a structure that is already sorted: from the raw-edge pair (ceeg_csr_build with self_loops = 0 and
 * its ceeg_csr_transpose) to the pair of A + I, entry for entry what ceeg_csr_build(self_loops = 1) + ceeg_csr_transpose give
 * (every input self-loop dropped, one (i,i) entry per node at its sorted place, g_perm = edge ids with the last dropped loop's id
 * or -1 on the loop entries, gt_perm = positions in the new forward structure) - without a second sort.  For loops that build both
 * structures of a new minibatch every step (scripts/run.py:310-326: the encoder's gcn_norm graph and the decoder's edge list).
 * Output capacity: rowptr[N] + N entries each.  All arrays int32 on the device. */
size_t ceeg_csr_add_self_loops_ws_bytes(int64_t N);
int ceeg_csr_add_self_loops(const int32_t* rowptr, const int32_t* col, const int32_t* perm, const int32_t* t_rowptr,
                            const int32_t* t_col, const int32_t* t_perm, int64_t N, int32_t* g_rowptr, int32_t* g_col,
                            int32_t* g_perm, int32_t* gt_rowptr, int32_t* gt_col, int32_t* gt_perm, void* ws, size_t ws_bytes,
                            void* stream);

/* Deterministic CSR segmented SpMM (the GCNConv aggregation and every transposed/backward pass):
 *   out[i,:] = act( sum_{p=rowptr[i]}^{rowptr[i+1]-1} v(p) * X[col[p],:]  + bias )      i in [0,N)
 *   v(p) = val[p]            if vperm == NULL
 *        = val[vperm[p]]     otherwise (values kept in the order of ANOTHER structure, e.g. the CSC pass)
 * Summation order is p ascending in fixed blocks (SPMM_CHUNK) whose partial sums are added in block order
 * -> bitwise reproducible, no atomics.  X [*,ldx] fp32, out [N,ldo] fp32, F % 4 == 0, 4 <= F <= 1024,
 * ldx/ldo multiples of 4, pointers 16-byte aligned.  bias [F] or NULL; relu != 0 applies max(.,0).
 * accumulate != 0 adds into `out` instead of overwriting it.  max_row_nnz: the longest row if known (lets the
 * long-row pass be skipped when no row exceeds the chunk), < 0 when unknown. */
size_t ceeg_spmm_ws_bytes(int64_t N, int64_t nnz, int F);
int ceeg_spmm_csr(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* vperm,
                  const float* X, int64_t ldx, int64_t N, int64_t nnz, int F, const float* bias, int relu,
                  int accumulate, float* out, int64_t ldo, int64_t max_row_nnz, void* ws, size_t ws_bytes, void* stream);

/* ceeg_spmm_csr with the pieces of the long rows given as a list built once per graph (the graph is static across steps):
 * long_items int32 [n_long][4] = (lo, hi, chunk*2 + slot, row) for every row longer than 256 entries, cut at the global
 * positions k*256; slot = 1 for a piece that starts its row in the middle of a chunk, else 0 (the layout of
 * ceeg_decoder_codes_*).  Same sums in the same order as ceeg_spmm_csr - bitwise equal - without the per-chunk row
 * search: one wave per listed piece.  max_row_nnz must be the true longest row. */
int ceeg_spmm_csr_items(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* vperm,
                        const float* X, int64_t ldx, int64_t N, int64_t nnz, int F, const float* bias, int relu,
                        int accumulate, float* out, int64_t ldo, int64_t max_row_nnz, const int32_t* long_items,
                        int64_t n_long, void* ws, size_t ws_bytes, void* stream);

/* ceeg_spmm_csr_items whose finished rows pass a ReLU's backward mask: out[i,:] = (sum_p v(p) X[col[p],:]) * (mask[i,:] > 0).
 * For dh = A_hat^T g followed by autograd's ReLU backward (h = relu(.) is the aggregated layer's input, scripts/run.py:96 via
 * PyG GCNConv): the mask rides in the epilogue of the pass that produces dh - bitwise what ceeg_relu_bwd_colsum's gm is -
 * instead of a separate read-modify-write of the [N, F] gradient.  mask: [N, ld_mask] fp32 (the ReLU's OUTPUT), 16-byte aligned. */
int ceeg_spmm_csr_masked(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* vperm,
                         const float* X, int64_t ldx, int64_t N, int64_t nnz, int F, const float* mask, int64_t ld_mask,
                         float* out, int64_t ldo, int64_t max_row_nnz, const int32_t* long_items, int64_t n_long,
                         void* ws, size_t ws_bytes, void* stream);

/* ceeg_spmm_csr_items (mask == NULL) / ceeg_spmm_csr_masked with the CSR's (column, value) pairs interleaved into ONE 8-byte
 * stream: colval = int32 pairs { col[p], bits of val[p] }, built once per graph.  Same sums in the same order (bitwise equal);
 * one lane-parallel 8-byte load per 64 entries instead of two 4-byte ones. */
int ceeg_spmm_csr_cv(const int32_t* rowptr, const void* colval, const float* X, int64_t ldx, int64_t N, int64_t nnz, int F,
                     const float* bias, int relu, const float* mask, int64_t ld_mask, float* out, int64_t ldo,
                     int64_t max_row_nnz, const int32_t* long_items, int64_t n_long, void* ws, size_t ws_bytes, void* stream);

/* ceeg_spmm_csr_cv with bf16 STORAGE (the opt-in reduced-precision step; no reference counterpart - north star: "rtol 1e-2
 * (bf16)"): X, mask and out hold bf16 (2-byte elements, ldx / ld_mask / ldo in elements, 8-byte aligned rows), every sum is
 * accumulated in fp32 in the order of the fp32 form and rounded to nearest-even once, on the store.  A gathered row is half the
 * bytes: the far-entry fetch that bounds the fp32 aggregation halves. */
int ceeg_spmm_csr_cv_bf16(const int32_t* rowptr, const void* colval, const void* X, int64_t ldx, int64_t N, int64_t nnz, int F,
                          const float* bias, int relu, const void* mask, int64_t ld_mask, void* out, int64_t ldo,
                          int64_t max_row_nnz, const int32_t* long_items, int64_t n_long, void* ws, size_t ws_bytes, void* stream);

/* ceeg_spmm_csr whose gathered matrix is a TABLE read through a row index: out[i,:] (+)= sum_p v(p) * table[xidx[col[p]], :].
 * For a quantised latent (z[n] = codebook[code[n]] bit for bit) this is the same sum as gathering z itself - bitwise equal -
 * with the rows coming from a few hundred KB that stay in L2 (the backward of the edge-list decoder over sampled negatives). */
int ceeg_spmm_csr_table(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* vperm,
                        const float* table, int64_t ldt, const int32_t* xidx, int64_t N, int64_t nnz, int F, int accumulate,
                        float* out, int64_t ldo, int64_t max_row_nnz, void* ws, size_t ws_bytes, void* stream);

/* Measurement entry (no reference counterpart; used by bench.py / scripts/spmm_probe.py, never by the model): the SpMM row
 * kernel's traversal with everything but the row gathers removed - no value stream, no FMAs, no epilogue, no output.  Its
 * time is the empirical ceiling of a whole-row gather SpMM on this graph.  mode bit 0: gather the entries with
 * |col - row| <= window, bit 1: the others (3 = all); bit 2: the FEATURE-SLICED schedule instead (XCD x walks all rows and
 * gathers only columns [x F/8, (x+1) F/8) of each source row; F % 32 == 0).  F % 4 == 0, F <= 256; ws: ceeg_spmm_ws_bytes(N, nnz, F). */
int ceeg_spmm_gather_probe(const int32_t* rowptr, const int32_t* col, const float* X, int64_t ldx, int64_t N, int64_t nnz,
                           int F, int mode, int window, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * A3+A4  VGAE.reparametrize + VGAE.kl_loss (upstream PyG; anchors: MAX_LOGSTD scripts/run.py:13,
 *        BaseCVAE.reparameterize scripts/models.py:38-43), fused in one pass.
 *   ls = min(logstd, max_logstd);  z = mu + eps*exp(ls)   (log_var != 0: z = mu + eps*exp(0.5*ls),
 *        the in-repo models.py:40 variant)
 *   kl = -0.5/N * sum_{n,d} (1 + 2*ls - mu^2 - exp(2*ls))
 * mu, logstd: [N,D] fp32 views with row strides ld_mu/ld_ls (so the fused [N,2D] SpMM output is read in
 * place).  eps: [N,D] fp32 explicit noise, or NULL -> Philox4x32-10(seed, element index) + Box-Muller,
 * restated bit-for-bit (integers) in oracle/ceeg_oracle.c.  training == 0: z = mu (eval).
 * kl: 1 float on the device.
 * Backward: dmu = gz + gkl*mu/N ; dlogstd = [logstd<=max] * (gz*eps*exp(ls) + gkl*(exp(2ls)-1)/N). */
size_t ceeg_reparam_kl_ws_bytes(int64_t N, int D);
int ceeg_reparam_kl_fwd(const float* mu, int64_t ld_mu, const float* logstd, int64_t ld_ls, const float* eps,
                        uint64_t seed, int64_t N, int D, float max_logstd, int log_var, int training,
                        float* z, float* kl, void* ws, size_t ws_bytes, void* stream);
int ceeg_reparam_kl_bwd(const float* mu, int64_t ld_mu, const float* logstd, int64_t ld_ls, const float* eps,
                        uint64_t seed, const float* gz, const float* gkl_dev, int64_t N, int D,
                        float max_logstd, int log_var, int training,
                        float* dmu, int64_t ld_dmu, float* dlogstd, int64_t ld_dls, void* stream);

/* Fused backward of the pair  z = reparametrize(mu, logstd);  (loss_vq, z_q) = VectorQuantizer(z)  (the order
 * BaseCVAE.forward uses, scripts/models.py:30-32): one pass computes
 *   gz = g_zq + g_z + s*(2/(N*D))*(z - W[idx])     (ceeg_vq_backward's dz with z recomputed from mu, logstd, eps/seed)
 *   dmu, dlogstd = ceeg_reparam_kl_bwd(gz, gkl)     (bitwise equal to running the two entry points one after the other)
 *   colsum[0:D | D:2D] = column sums of dmu | dlogstd (optional: the bias gradient of the encoder head; fixed order)
 * so dz is never written or re-read.  g_zq / g_z: [N,D] fp32 or NULL; s_dev / gkl_dev: 1 float on the device or NULL;
 * idx: int64 [N]; D % 4 == 0, 16-byte aligned operands.  The codebook gradient stays with ceeg_vq_backward (dz = NULL). */
size_t ceeg_reparam_vq_bwd_ws_bytes(int64_t N, int D);
int ceeg_reparam_vq_bwd(const float* mu, int64_t ld_mu, const float* logstd, int64_t ld_ls, const float* eps, uint64_t seed,
                        const float* g_zq, const float* g_z, const float* W, const int64_t* idx, int n_e,
                        const float* s_dev, const float* gkl_dev, int64_t N, int D, float max_logstd, int log_var,
                        int training, float* dmu, int64_t ld_dmu, float* dlogstd, int64_t ld_dls, float* colsum,
                        void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * A6  InnerProductDecoder.forward + GAE.recon_loss (upstream PyG), literal "+EPS" form, EPS = 1e-15:
 *   v_e = <z[i_e], z[j_e]> ; pos: -log(sigmoid(v)+EPS) ; neg: -log(1-sigmoid(v)+EPS)
 * Edge-list form (any edge order; used for user-supplied pos/neg edge_index and for sampled negatives):
 *   i,j: [E] int32 node ids; a pair with a negative id (a sample ceeg_neg_sample_uniform gave up on) is SKIPPED: term 0,
 *   coef 0, not counted.  negative bit 0 selects the neg term; bit 1 makes prob[e] the raw inner product v_e.  Outputs: coef[e] = d(term_e)/dv_e (fp32, unscaled),
 *   loss_sum: TWO floats (device) = { sum_e term_e, number of pairs counted }  (the caller divides: mean),
 *   optionally prob[e] = sigmoid(v_e) (decoder(z, edge_index, sigmoid=True)).
 *   zidx (int32 [N] or NULL): quantised latent - node n's row is z[zidx[n]] and `z` is the codebook (as in the fused form).
 * The backward  dz[n] = sum_{e: i_e=n} c_e z[j_e] + sum_{e: j_e=n} c_e z[i_e]  is two ceeg_spmm_csr calls
 * over the edge list's CSR and its transpose (val = coef via perm) — atomics-free by construction. */
size_t ceeg_edge_bce_ws_bytes(int64_t E, int D);
int ceeg_edge_bce_fwd(const float* z, int64_t ldz, const int32_t* zidx, const int32_t* i, const int32_t* j, int64_t E, int D,
                      int negative, float* coef, float* prob, float* loss_sum,
                      void* ws, size_t ws_bytes, void* stream);

/* Fused CSR-ordered decoder with permutation-structured negatives (the trainer's fast path):
 * for CSR row r (= dst) and entry p (src = col[p]):
 *   positive pair (col[p], r);   negative pair (col[p], sigma(r)),
 *   sigma(x) = (x - x % m) + (a*(x % m) + b) % m with m = sig_mod | N, gcd(a, m) = 1: m = N permutes all nodes,
 *   m = nodes per graph keeps the negatives of a block-diagonal batch (equal-sized graphs) inside their graph.
 *   A negative is INVALID (masked out, coef 0) when col[p]==sigma(r) or (col[p] -> sigma(r)) is an edge
 *   (binary search in row sigma(r), rows being src-sorted).
 * One gather of z[col[p]] serves both dots.  Outputs coef_pos[p*coef_stride], coef_neg[p*coef_stride] (unscaled d term/dv;
 * coef_stride 1 = two dense arrays, 2 = one interleaved [nnz][2] array with coef_neg = coef_pos + 1, which lets the
 * backward fetch a pair from one memory sector),
 * sums[0]=sum pos terms, sums[1]=sum valid neg terms, sums[2]=number of valid negatives (as float),
 * all written to `sums` (3 floats, device).  rowptr/col: CSR WITHOUT self-loops (ceeg_csr_build
 * self_loops=0).
 * zidx (int32 [N] or NULL): when the latent is QUANTISED, node n's row is z[zidx[n]] and `z` is the codebook
 * ([n_e, ldz], a few hundred KB that stay in L1/L2) - the gathers then cost 4 bytes of HBM per edge instead of a
 * D*4-byte row.  Entries must lie in [0, n_e); the result is identical to passing the expanded [N, D] matrix.
 * grow_pos / grow_neg ([N, ldg] fp32, both or neither; NULL when no gradient is wanted): the row-side halves of
 * dz, accumulated while the gathered pieces are still in registers (unscaled):
 *   grow_pos[r] = sum_{p in row r} cp[p] z[col p],   grow_neg[sigma(r)] = sum_{p in row r} cn[p] z[col p].
 * Handing them to ceeg_decoder_fused_bwd leaves it two gather passes instead of four.
 * max_row_nnz: longest CSR row if the caller knows it (-1: unknown).  Rows longer than 256 entries (in-degree hubs) are
 * cut at global positions k*256 and computed piecewise by a second kernel; a hint <= 256 skips that launch. */
size_t ceeg_decoder_fused_ws_bytes(int64_t N, int64_t nnz, int D);
int ceeg_decoder_fused_fwd(const float* z, int64_t ldz, const int32_t* zidx, const int32_t* rowptr, const int32_t* col,
                           int64_t N, int64_t nnz, int D, int64_t sig_a, int64_t sig_b, int64_t sig_mod,
                           float* coef_pos, float* coef_neg, int coef_stride, float* grow_pos, float* grow_neg, int64_t ldg,
                           float* sums, int64_t max_row_nnz, void* ws, size_t ws_bytes, void* stream);
/* Backward of the fused decoder: dz[n] = gp * (pos terms) + gn * (neg terms) where gp/gn are read from
 * the device (gscale[0], gscale[1]: upstream grad times 1/E_pos resp. 1/n_valid_neg).
 *   dz[n] = sum_{p in row n} gp*cp[p] z[col p]                 (n as dst of positives)
 *         + sum_{q in trow n} gp*cp[tperm q] z[tcol q]         (n as src of positives)
 *         + sum_{p in row sigma^-1(n)} gn*cn[p] z[col p]       (n as negative endpoint sigma(r))
 *         + sum_{q in trow n} gn*cn[tperm q] z[sigma(tcol q)]  (n as src of negatives)
 * sig_ainv = a^-1 mod N.  With grow_pos/grow_neg from the forward (both non-NULL) the first and third sums are
 * taken from those tables and ONE fused kernel adds both transposed sums (bitwise equal to running them as two SpMM
 * passes); with NULL all four passes run. */
int ceeg_decoder_fused_bwd(const float* z, int64_t ldz, const int32_t* zidx, const int32_t* rowptr, const int32_t* col,
                           const int32_t* t_rowptr, const int32_t* t_col, const int32_t* t_perm,
                           int64_t N, int64_t nnz, int D, int64_t sig_a, int64_t sig_b, int64_t sig_ainv, int64_t sig_mod,
                           const float* coef_pos, const float* coef_neg, int coef_stride, const float* grow_pos,
                           const float* grow_neg, int64_t ldg, const float* gscale,
                           float* dz, int64_t lddz, int64_t max_row_nnz, int64_t max_trow_nnz,
                           void* ws, size_t ws_bytes, void* stream);

/* Fused decoder for a QUANTISED latent, code form (the trainer's path when the decoder input is VectorQuantizer's z_q:
 * z[n] = W[code[n]] bit for bit).  Same pairs, masks, terms and sums as ceeg_decoder_fused_fwd/_bwd (upstream PyG
 * InnerProductDecoder + GAE.recon_loss, literal "+EPS" form), evaluated through the codebook:
 *   forward : G = W W^T (n_e x n_e, k-ascending fmaf chains) once per call; <z_i, z_j> = G[code i][code j];
 *             sig_a == 0 means POSITIVES ONLY (every cn = 0, sums[1] = sums[2] = 0; the backward takes sig_a == 0 the same way and
 *             then walks only the positive lists);
 *             rec[p] = { cp[p], cn[p] (0 when the negative is masked), code[r] | code[sigma r] << 16, code[col p] }
 *             for CSR entry p of row r (16 bytes, uint32 x 4; cp/cn as float bits); sums[0..2] as the fused decoder.
 *   backward: dz[n] = gp*(sum_{p in row n} cp W[kc] + sum_{q in trow n} cp[tperm q] W[kr]) +
 *                     gn*(sum_{p in row sigma^-1 n} cn W[kc] + sum_{q in trow n} cn[tperm q] W[ksr])
 *             with a 64-column slab of the whole codebook resident in LDS (no global row gathers).
 * Shapes: ceeg_decoder_codes_supported(n_e, D) != 0  <=>  n_e <= 576 and D in {64, 128, 192, 256}; code: int32 [N] in
 * [0, n_e) (clamped).  Lists longer than ceeg_decoder_codes_chunk() (= 256) entries are cut at global positions
 * k*256; the caller passes them as piece lists built once per graph: long_items / t_long_items, int32 [n][4] =
 * (lo, hi, chunk*2 + slot, row) with slot = 1 for a piece that starts its row in the middle of a chunk, else 0
 * (CSR rows resp. transposed rows; required when max_row_nnz / max_trow_nnz > 256, else NULL / 0).
 * Conditional decoders (scripts/models.py:259 `cat((latent, condition))` with a condition that is constant inside a graph,
 * block-diagonal batch, sig_mod = nodes per graph): the condition's share of every dot product of row r is the constant
 * |c_graph(r)|^2 - pass it as row_add[r] (float [N], NULL = 0); the backward's coef_sum[n] (float [N], NULL = not wanted)
 * is the summed, scaled coefficient of all pairs of node n, so d loss / d condition_columns[n] = coef_sum[n] * c_graph(n).
 * No counterpart in the reference beyond A6 itself; an algebraic re-association for the quantised case. */
/* The edge-list form (ceeg_edge_bce_fwd) on a quantised latent in code form: v_e = G[code i_e][code j_e] with G = W W^T rebuilt per
 * call (as ceeg_decoder_codes_fwd).  Same terms, skipping rule and outputs (coef[e], loss_sum = {sum of terms, pairs counted});
 * the dot products differ from the row-gather form only by the summation order of the fmaf chain.  n_e <= 4096, D % 4 == 0. */
size_t ceeg_edge_bce_codes_ws_bytes(int64_t E, int n_e);
int ceeg_edge_bce_codes_fwd(const float* W, int64_t ldw, int n_e, int D, const int32_t* code, const int32_t* i, const int32_t* j,
                            int64_t E, int negative, float* coef, float* loss_sum, void* ws, size_t ws_bytes, void* stream);
/* recon_loss over an ARBITRARY pair list on a quantised latent, with its gradient, WITHOUT a CSR of the list: the path of the
 * negatives that upstream PyG `negative_sampling` draws afresh every step (GAE.recon_loss; call site scripts/run.py:93-95 of the
 * north-star step).  ceeg_pair_bce_codes_supported(n_e, D, N, E) != 0 <=> ceeg_decoder_codes_supported(n_e, D), N < 2^22 - 1
 * and E < 2^30.  A pair's term and coefficient depend only on its two codes (v = G[code i][code j], G = W W^T):
 *   forward : coef_table (float [n_e * n_e], caller-owned, kept for the backward) receives C[a][b] = d term / dv at G[a][b];
 *             loss_sum = {sum of terms, pairs counted} as ceeg_edge_bce_codes_fwd (pairs with an id outside [0, N) are
 *             skipped); with keys non-NULL (uint32 [2E], 8-byte aligned) it also leaves the 2E INCIDENCES of the list:
 *             keys[2k] = j_k << 10 | code[i_k], keys[2k+1] = i_k << 10 | code[j_k]; node field N for skipped pairs;
 *   backward: one stable device radix sort of the keys on their node bits (order inside a node = incidence order: fixed
 *             summation order, no float atomics), then dz[n] (+)= gscale[0] * sum_{keys of n} C[code n][key & 1023] * W[key & 1023]
 *             from an LDS-resident 64-column codebook slab.  code: the same int32 [N] as in the forward; gscale: device float
 *             (upstream gradient / pairs counted); accumulate != 0 adds to dz. */
int ceeg_pair_bce_codes_supported(int n_e, int D, int64_t N, int64_t E);
size_t ceeg_pair_bce_codes_ws_bytes(int64_t E, int64_t N, int n_e);
int ceeg_pair_bce_codes_fwd(const float* W, int64_t ldw, int n_e, int D, const int32_t* code, const int32_t* i, const int32_t* j,
                            int64_t E, int64_t N, int negative, uint32_t* keys, float* coef_table, float* loss_sum, void* ws,
                            size_t ws_bytes, void* stream);
int ceeg_pair_bce_codes_bwd(const float* W, int64_t ldw, int n_e, int D, const int32_t* code, const uint32_t* keys,
                            const float* coef_table, int64_t E, int64_t N, const float* gscale, int accumulate, float* dz,
                            int64_t lddz, void* ws, size_t ws_bytes, void* stream);
/* negative_sampling fused into the pair-list forward: pair k is DRAWN inside the kernel (the samples of
 * ceeg_neg_sample_uniform_filtered bit for bit, written to neg_src / neg_dst as that call writes them) and goes straight into the
 * term lookup and the incidence keys - loss_sum, keys and coef_table equal the two calls run one after the other, exactly; the
 * 128 MB pair list is written once and never re-read.  rowptr / col: the positives' CSR the sampler rejects against. */
int ceeg_neg_sample_pair_bce_codes_fwd(const float* W, int64_t ldw, int n_e, int D, const int32_t* code, const int32_t* rowptr,
                                       const int32_t* col, int64_t N, int64_t num, uint64_t seed, const uint32_t* filter,
                                       int filter_bits, int32_t* neg_src, int32_t* neg_dst, uint32_t* keys, float* coef_table,
                                       float* loss_sum, void* ws, size_t ws_bytes, void* stream);
/* The backward of BOTH pair sets of the literal recon_loss in one gradient kernel: ceeg_decoder_codes_bwd with sig_a = 0 (the
 * positives' three lists) plus the sampled pairs' incidence keys (ceeg_pair_bce_codes_fwd) sorted by node as a FOURTH list of every
 * node: dz = gscale[0] * (positives) + gscale[1] * sum_keys C[code n][key & 1023] W[key & 1023], written once - bitwise what
 * ceeg_decoder_codes_bwd followed by ceeg_pair_bce_codes_bwd(accumulate = 1) leaves, without the second slab fill and the
 * read-modify-write of the [N, D] gradient. */
size_t ceeg_decoder_codes_bwd_pairs_ws_bytes(int64_t N, int64_t nnz, int n_e, int D, int64_t n_long, int64_t n_tlong, int64_t E);
int ceeg_decoder_codes_bwd_pairs(const float* W, int64_t ldw, int n_e, int D, const int32_t* rowptr, const int32_t* t_rowptr,
                                 const int32_t* t_perm, int64_t N, int64_t nnz, const int32_t* long_items, int64_t n_long,
                                 const int32_t* t_long_items, int64_t n_tlong, int64_t max_row_nnz, int64_t max_trow_nnz,
                                 const void* rec, const int32_t* code, const uint32_t* keys, const float* coef_table, int64_t E,
                                 const float* gscale, float* dz, int64_t lddz, void* ws, size_t ws_bytes, void* stream);
int ceeg_decoder_codes_chunk(void);
int ceeg_decoder_codes_supported(int n_e, int D);
size_t ceeg_decoder_codes_ws_bytes(int64_t N, int64_t nnz, int n_e, int D, int64_t n_long, int64_t n_tlong);
int ceeg_decoder_codes_fwd(const float* W, int64_t ldw, int n_e, int D, const int32_t* code, const int32_t* rowptr,
                           const int32_t* col, int64_t N, int64_t nnz, int64_t sig_a, int64_t sig_b, int64_t sig_mod,
                           const int32_t* long_items, int64_t n_long, int64_t max_row_nnz, const float* row_add, void* rec,
                           float* sums, void* ws, size_t ws_bytes, void* stream);
int ceeg_decoder_codes_bwd(const float* W, int64_t ldw, int n_e, int D, const int32_t* rowptr, const int32_t* t_rowptr,
                           const int32_t* t_perm, int64_t N, int64_t nnz, int64_t sig_a, int64_t sig_b, int64_t sig_ainv,
                           int64_t sig_mod, const int32_t* long_items, int64_t n_long, const int32_t* t_long_items,
                           int64_t n_tlong, int64_t max_row_nnz, int64_t max_trow_nnz, const void* rec, const float* gscale,
                           float* dz, int64_t lddz, float* coef_sum, void* ws, size_t ws_bytes, void* stream);

/* Streaming helpers around the dense transform of a GCN layer (no counterpart of their own in the reference:
 * they stand where its host code calls torch.relu / autograd's ReLU backward / the bias-gradient reduction).
 *   ceeg_bias_relu       : y[n,:] = act(y[n,:] + bias) in place; bias may be NULL, relu != 0 applies max(.,0)
 *   ceeg_relu_bwd_colsum : gm = g * (y > 0) (gm may alias g) and colsum[f] = sum_n gm[n,f], one pass
 *   ceeg_colsum          : colsum[f] = sum_n g[n,f]   (deterministic: fixed-order block partials, fp64 finish)
 *   ceeg_row_sqnorm      : out[n] = sum_k x[n,k]^2 (k-ascending fmaf chain): |c_graph(n)|^2, the condition's share of every dot product
 *                          of row n when the decoder runs on cat((latent, condition)) (scripts/models.py:259) in code form
 * y, g, x: [N, ld] fp32 row-major, F % 4 == 0, F <= 1024, ld % 4 == 0, 16-byte aligned. */
int ceeg_row_sqnorm(const float* x, int64_t ld, int64_t N, int F, float* out, void* stream);
/* cat((x, condition[batch]), dim=1) in one pass (scripts/models.py:162 `torch.cat((x, condition))` with the per-graph condition
 * broadcast to the nodes through the collated `batch` vector, scripts/models.py:121-122): out [N, F + C] contiguous,
 * cond [G, C] contiguous, batch [N] int64 graph ids (ids outside [0, G) are clamped), F % 4 == 0, C % 4 == 0. */
int ceeg_concat_condition(const float* x, int64_t ldx, int64_t N, int F, const float* cond, int C, int64_t G, const int64_t* batch,
                          float* out, void* stream);
int ceeg_bias_relu(float* y, int64_t ld, int64_t N, int F, const float* bias, int relu, void* stream);
size_t ceeg_colsum_ws_bytes(int64_t N, int F);
int ceeg_relu_bwd_colsum(const float* g, int64_t ldg, const float* y, int64_t ldy, float* gm, int64_t ldm, int64_t N, int F,
                         float* colsum, void* ws, size_t ws_bytes, void* stream);
int ceeg_colsum(const float* g, int64_t ldg, int64_t N, int F, float* colsum, void* ws, size_t ws_bytes, void* stream);

/* recon_loss's last step (upstream PyG GAE.recon_loss: pos_loss.mean() + neg_loss.mean()) in one launch:
 *   out[0] = pos_sum / pos_count + neg_sum / max(neg_count, 1);  out[1] = 1 / pos_count;  out[2] = 1 / max(neg_count, 1);
 *   out[3..5] = pos_sum, neg_sum, neg_count.   pos_sum / neg_sum / neg_count: one device float each; out: 6 device floats. */
int ceeg_recon_finish(const float* pos_sum, const float* neg_sum, const float* neg_count, float pos_count, float* out, void* stream);

/* The step's loss sum (scripts/run.py:95 `loss = recon_loss + embedding_loss`, plus the VGAE example's (1/N) kl_loss) in one
 * launch: g == NULL: out[0] = rec + kl * kl_scale + vq;  g != NULL (backward): out[0] = g, out[1] = g * kl_scale.
 * All operands are single device floats. */
int ceeg_loss_sum(const float* rec, const float* kl, const float* vq, float kl_scale, const float* g, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * N1  GATConv attention (upstream PyG GATConv, instantiated by the reference at scripts/models.py:106), one head per
 * call.  rowptr/col: CSR grouped by destination with one self-loop per node (ceeg_csr_build self_loops=1 on the
 * edges without their own self-loops, as PyG's remove_self_loops + add_self_loops).
 *   forward : e[p] = leaky_relu(a_src[col p] + a_dst[row p], slope);  w[p] = exp(e[p] - max_row) / (sum_row + 1e-16)
 *             (PyG utils.softmax); the aggregation out[i] = sum_p w[p] xp[col p] is ceeg_spmm_csr with val = w.
 *   backward: dw[p] = <g[row p], xp[col p]>;  de[p] = w[p] (dw[p] - sum_k w[k] dw[k]);
 *             dl[p] = de[p] * (a_src[col p] + a_dst[row p] > 0 ? 1 : slope);  d_adst[i] = sum_{p in row i} dl[p];
 *             d_asrc[j] = sum of dl over the entries whose source is j = ceeg_segment_sum_perm over the transposed CSR.
 * g, xp: [N, ld] fp32 with C columns (C % 4 == 0, C <= 2048 - the reference's layers are 2048 wide -; a head's column slice
 * of a wider matrix is fine). */
int ceeg_gat_softmax_fwd(const int32_t* rowptr, const int32_t* col, const float* a_src, const float* a_dst,
                         int64_t N, int64_t nnz, float slope, float* w, void* stream);
int ceeg_gat_attn_bwd(const int32_t* rowptr, const int32_t* col, const float* w, const float* a_src,
                      const float* a_dst, const float* g, int64_t ldg, const float* xp, int64_t ldx, int64_t N,
                      int64_t nnz, int C, float slope, float* dl, float* d_adst, void* stream);
/* Attention scores of every head in one pass over the projected features (PyG GATConv: (x_proj * att).sum(-1)):
 *   a_src[h*N + n] = <xp[n, h*C : (h+1)*C], att_src[h*C : (h+1)*C]>,  a_dst likewise.     xp [N, ldx], att_* [H*C].
 * Their backward, fused with the accumulation into the gradient of xp (dxp holds the aggregation's share on entry):
 *   dxp[n, h, :] += d_asrc[h*N+n] att_src[h, :] + d_adst[h*N+n] att_dst[h, :]
 *   d_att_src[h*C + c] = sum_n d_asrc[h*N+n] xp[n, h*C + c]   (d_att_dst likewise; block partials in a fixed order, fp64
 *   across blocks: deterministic).  C % 4 == 0, C <= 2048, H*C <= 16384. */
int ceeg_gat_scores_fwd(const float* xp, int64_t ldx, const float* att_src, const float* att_dst, int64_t N, int H, int C,
                        float* a_src, float* a_dst, void* stream);
size_t ceeg_gat_scores_bwd_ws_bytes(int64_t N, int H, int C);
int ceeg_gat_scores_bwd(const float* xp, int64_t ldx, const float* att_src, const float* att_dst, const float* d_asrc,
                        const float* d_adst, int64_t N, int H, int C, float* dxp, int64_t ldd, float* d_att_src,
                        float* d_att_dst, void* ws, size_t ws_bytes, void* stream);
/* out[r] = sum_{q in [t_rowptr[r], t_rowptr[r+1])} v[t_perm[q]]  (t_perm NULL: v[q]); fixed-order wave sums. */
int ceeg_segment_sum_perm(const int32_t* t_rowptr, const int32_t* t_perm, const float* v, int64_t N,
                          float* out, void* stream);
/* The same sums with the rows longer than `chunk` entries taken from the graph's static piece list (items [n][4] int32 =
 * (lo, hi, -, row), rows and pieces ascending, every row longer than chunk listed completely - CsrPair.long_items): one wave
 * per piece, pieces of a row added in list order.  A hub source no longer serialises on one wave. */
size_t ceeg_segment_sum_items_ws_bytes(int64_t n_items);
int ceeg_segment_sum_perm_items(const int32_t* t_rowptr, const int32_t* t_perm, const float* v, int64_t N, const int32_t* items,
                                int64_t n_items, int chunk, float* out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * N4  torch_geometric.nn.LayerNorm(in_channels, mode='graph') as the reference instantiates it after every GATConv
 * (scripts/models.py:107), optionally fused with the ReLU that follows it (scripts/models.py:108), over a collated batch:
 * ptr int32 [G+1] = row offsets of the graphs (the batch is sorted by graph), rows [ptr[g], ptr[g+1]) belong to graph g.
 *   mean_g, var_g over ALL rows and channels of graph g (biased variance; an empty graph counts as one row);
 *   out = (x - mean_g) / sqrt(var_g + eps) * weight + bias  [then max(., 0) when relu != 0]
 * Forward: two reads of x and one write; the statistics are fp64 sums in a fixed order (no atomics).  mean / rstd ([G]
 * DOUBLES: x - mean keeps the digits a rounded mean would lose when |mean| >> std) are outputs the backward takes back.  weight / bias may be NULL (affine=False).
 * Backward (y = the forward's output, needed only with relu != 0 for the mask): with gm = g * [y > 0], xhat = (x-mean) rstd,
 *   dx = rstd * (gm w - mean_g(gm w) - xhat mean_g(gm w xhat)),  dweight[f] = sum_n gm xhat,  dbias[f] = sum_n gm
 * (dweight / dbias may be NULL).  F % 4 == 0, F <= 2048, row strides multiples of 4, 16-byte aligned pointers. */
size_t ceeg_graph_layernorm_ws_bytes(int64_t N, int F, int64_t G);
int ceeg_graph_layernorm_fwd(const float* x, int64_t ldx, const int32_t* ptr, int64_t N, int F, int64_t G, const float* weight,
                             const float* bias, float eps, int relu, float* out, int64_t ldo, double* mean, double* rstd,
                             void* ws, size_t ws_bytes, void* stream);
int ceeg_graph_layernorm_bwd(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* y, int64_t ldy,
                             const int32_t* ptr, int64_t N, int F, int64_t G, const float* weight, const double* mean,
                             const double* rstd, int relu, float* dx, int64_t lddx, float* dweight, float* dbias,
                             void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * N2  PyG-style collation of disjoint graphs on the device - torch_geometric.loader.DataLoader(..., follow_batch=[...]) as
 * the reference calls it (scripts/run.py:310-326) and its inverse torch_geometric.data.separate (scripts/dataset.py:66-92).
 *   ceeg_segment_ids  : out[i] = g for ptr[g] <= i < ptr[g+1], i in [0, total) - the `batch` vector (ptr = node offsets)
 *                       and every followed key's `<key>_batch` vector (ptr = that key's row offsets).  ptr: int64 [G+1],
 *                       ascending, ptr[0] = 0, ptr[G] = total; empty segments allowed.
 *   ceeg_collate_edges: edges_out[r*ld_out + e] = edges_in[r*ld_in + e] + sign * node_ptr[g(e)] for r = 0, 1, where g(e) is
 *                       the graph of edge e by edge_ptr (int64 [G+1]); sign = +1 is the block-diagonal offsetting of the
 *                       concatenated per-graph edge lists, sign = -1 its inverse (what `separate` returns for a slice).
 *                       edge_batch (int64 [E] or NULL) receives g(e).  In place (edges_out == edges_in) is allowed. */
int ceeg_segment_ids(const int64_t* ptr, int64_t G, int64_t total, int64_t* out, void* stream);
int ceeg_collate_edges(const int64_t* edges_in, int64_t ld_in, const int64_t* edge_ptr, const int64_t* node_ptr, int64_t G,
                       int64_t E, int sign, int64_t* edges_out, int64_t ld_out, int64_t* edge_batch, void* stream);
/* ceeg_collate_edges with sign = +1 that also answers "does every edge stay inside its graph" on the device: *outside (one int32)
 * becomes 0, then 1 if some endpoint is not a local id in [0, node_ptr[g+1] - node_ptr[g]) of its graph g.  The per-graph negative
 * sampling of the conditional model (scripts/models.py:259 on a collated batch) relies on that layout. */
int ceeg_collate_edges_checked(const int64_t* edges_in, int64_t ld_in, const int64_t* edge_ptr, const int64_t* node_ptr, int64_t G,
                               int64_t E, int64_t* edges_out, int64_t ld_out, int64_t* edge_batch, int32_t* outside, void* stream);

/* ------------------------------------------------------------------------------------------------
 * N3  Ancestor-subgraph extraction of the reference's dataset builder for ALL targets at once (scripts/dataset.py:22-46,
 * 201-225), a frontier expansion on the device.  The graph comes as the CSR grouped by target of ceeg_csr_build
 * (self_loops = 0: rowptr / col / perm, perm = edge id) plus `entities` [E] int64 by edge id (how many CSV rows collapsed
 * into the edge, dataset.py:147-153; edges numbered in first-appearance order).
 *   ceeg_top_predecessors: top[v*n + i] = i-th of v's predecessors with entities >= min_entities, heaviest first, ties by edge
 *       id ascending, at most min(n, in-degree of v) of them, -1 padded (get_top_n_valid_predecessors, dataset.py:22-33).
 *   ceeg_ancestors: anc[t*cap + .] = sorted, duplicate-free union of the top predecessors over `gens` generations from
 *       targets[t] (NULL: t itself), -1 padded; cnt[t] = how many (get_n_gen_ancestors, dataset.py:35-46).
 *       cap >= ceeg_ancestors_cap(n, gens) = n + n^2 + ... + n^gens (39 for the reference's 3 and 3), <= 128.
 *   ceeg_ancestor_subgraph_count / _fill: the edges induced on each target's ancestors (dataset.py:201-225) with LOCAL ids
 *       (position in the sorted list).  count writes slot_count[t*cap + k] = edges INTO the k-th ancestor; the caller takes the
 *       exclusive prefix sum (slot_off) and fill writes the (source, target) pairs of target t's subgraph at
 *       [slot_off[t*cap], slot_off[(t+1)*cap]) ordered by (target slot, source id): a collated batch of subgraphs. */
int ceeg_top_predecessors(const int32_t* rowptr, const int32_t* col, const int32_t* perm, const int64_t* entities, int64_t N,
                          int n, int64_t min_entities, int32_t* top, void* stream);
int ceeg_ancestors_cap(int n, int gens);
int ceeg_ancestors(const int32_t* top, int64_t N, int n, const int32_t* targets, int64_t T, int gens, int cap, int32_t* anc,
                   int32_t* cnt, void* stream);
int ceeg_ancestor_subgraph_count(const int32_t* rowptr, const int32_t* col, const int32_t* anc, const int32_t* cnt, int64_t T, int cap,
                                 int64_t* slot_count, void* stream);
int ceeg_ancestor_subgraph_fill(const int32_t* rowptr, const int32_t* col, const int32_t* anc, const int32_t* cnt, int64_t T, int cap,
                                const int64_t* slot_off, int32_t* sub_src, int32_t* sub_dst, void* stream);

/* N4  Learned-attention token pooling of the reference's conditioners (scripts/models.py:118-119; the same pair of lines at
 * :138-139 and :170-171): attn = nn.Linear(H, 1); x = bmm(hidden.permute(0,2,1), attn(hidden)).squeeze(2), i.e.
 *   a[n,t] = <hidden[n,t,:], w> + b,   x[n,:] = sum_t a[n,t] hidden[n,t,:]        hidden [N, T, H] contiguous fp32
 * in ONE pass over hidden (one workgroup per node; tokens summed in ascending order).  a [N, T] is an output the backward
 * takes back.  Backward: da[n,t] = <g[n,:], hidden[n,t,:]>; dhidden[n,t,:] = a[n,t] g[n,:] + da[n,t] w (NULL: not wanted);
 * dw_part [N, H] / db_part [N, 4] are per-node partials of dw = sum_{n,t} da hidden and db = sum da (column 0) which the caller
 * reduces over nodes with ceeg_colsum (fixed order).  H % 4 == 0, T <= 512, 16-byte aligned pointers; b may be NULL. */
int ceeg_token_pool_fwd(const float* hidden, int64_t N, int T, int H, const float* w, const float* b, float* a, float* x, void* stream);
int ceeg_token_pool_bwd(const float* hidden, int64_t N, int T, int H, const float* w, const float* a, const float* g, float* dhidden,
                        float* dw_part, float* db_part, void* stream);

/* negative_sampling(edge_index, num_nodes, num_neg_samples) — upstream PyG (python `random`, cannot be
 * matched): uniform variant.  k-th sample: (src,dst) = Philox4x32-10(seed, k, attempt) mod N, rejected
 * when src==dst or src->dst is an edge of the given CSR (rows=dst, src-sorted), up to 16 attempts; a
 * sample that never succeeds is written as (-1,-1).  Bit-exact restatement: oracle/ceeg_oracle.c. */
int ceeg_neg_sample_uniform(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t num, uint64_t seed,
                            int32_t* neg_src, int32_t* neg_dst, void* stream);
/* The same samples, bit for bit, with an edge-set PREFILTER built once per graph: a bitmap of 2^bits bits (bits =
 * ceeg_neg_filter_bits(nnz): at least 16 bits per edge) with the bit of hash(src, dst) set for every edge.  A candidate whose
 * bit is clear is not an edge (one random read); only candidates with a set bit take the row's binary search.
 * filter: uint32 [2^bits / 32]; NULL = no prefilter (= ceeg_neg_sample_uniform). */
int ceeg_neg_filter_bits(int64_t nnz);
int ceeg_neg_filter_build(const int32_t* rowptr, const int32_t* col, int64_t N, int bits, uint32_t* filter, void* stream);
int ceeg_neg_sample_uniform_filtered(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t num, uint64_t seed,
                                     const uint32_t* filter, int filter_bits, int32_t* neg_src, int32_t* neg_dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CEEG_H */

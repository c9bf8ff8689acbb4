/*
 * dvt_hip.h -- C ABI of libdvt_hip.so: the MI355X (gfx950) kernels behind the
 * video-clip forward/backward hot path of data-efficient-video-transformers.
 *
 * The reference has no native layer: every entry point below replaces a
 * third-party PyTorch operator at the reference call site cited next to it
 * (paths relative to the reference repository root).  INTEGRATION.md shows the
 * ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - Plain C: raw device pointers, int64 sizes / element strides, enums as int.
 *   - Every function is asynchronous on `stream` (a hipStream_t passed as
 *     void*; NULL = the legacy default stream), never synchronises, never
 *     allocates or retains memory: the caller owns all buffers and workspaces.
 *   - Return value: 0 (DVT_OK) or a negative dvt_status; the message for the
 *     last failure on the calling thread is dvt_last_error().  No C++
 *     exception crosses the boundary.
 *   - Activations are `dtype` (DVT_BF16 or DVT_F32 ...); LayerNorm scale/shift,
 *     biases, positional tables, statistics and all parameter gradients are
 *     float32.  Accumulation is always float32.
 *   - Matrices are row-major with the last dimension contiguous unless an
 *     element stride says otherwise.
 */
#ifndef DVT_HIP_H
#define DVT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a descriptor struct changes layout (fields are only ever appended) or an entry point changes
 * signature.  v2: dvt_gemm_desc / dvt_conv_desc gained defer_reduce / pending / carry, dvt_splitk_pending and the
 * head-wise / folded-attention descriptors were added.  Callers MUST zero-initialise every descriptor (memset / = {0})
 * before filling it: a zero in a field this header adds later means "feature off".  v5: the `aux` matrix of DVT_EPI_GELU /
 * DVT_EPI_DGELU holds the activation's DERIVATIVE (was the pre-activation): a meaning changed, no layout. */
#define DVT_ABI_VERSION 5

typedef void* dvt_stream_t; /* hipStream_t */

enum dvt_dtype { DVT_F32 = 0, DVT_BF16 = 1, DVT_F16 = 2 };

enum dvt_status {
  DVT_OK = 0,
  DVT_ERR_BAD_ARG = -1,     /* null pointer, negative size, misaligned buffer ... */
  DVT_ERR_UNSUPPORTED = -2, /* shape / dtype this build has no kernel for */
  DVT_ERR_HIP = -3          /* a HIP runtime call or launch failed */
};

int dvt_version(void);
const char* dvt_last_error(void);
/* Fills compute-unit count, LDS bytes per CU and the gcnArchName of the
 * current device. */
int dvt_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len);

/* ---------------------------------------------------------------- casts / adds
 * dst[i] = (dst_dtype) src[i].  Replaces tensor.to(dtype) on the path (bf16
 * compute copies of the fp32 master weights, clip tensors). */
int dvt_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n,
             dvt_stream_t stream);
/* out = a + b (residual adds `attn(x) + x`, `ff(x) + x`: src/models/vit.py:73-74
 * when the producing GEMM is not the one fusing it). */
int dvt_add(const void* a, const void* b, void* out, int64_t n, int dtype, dvt_stream_t stream);
/* out[r, :] = x[r, :] + table[r / rows_per_entry, :]  -- sinusoidal PositionalEncoding on
 * seq-first [L, B, E] activations: `x + self.pe[:x.size(0)]`, frame_transformer.py:32-34
 * (rows_per_entry = B).  table: [rows / rows_per_entry, d] f32. */
int dvt_add_rowtable(const void* x, const float* table, void* out, int64_t rows, int64_t d,
                     int64_t rows_per_entry, int dtype, dvt_stream_t stream);
/* Strided 2-D copy: dst[r*dst_ld + c] = src[r*src_ld + c], r < rows, c < cols (elements).
 * src_ld == 0 broadcasts one row.  Used for the per-sample CLS-clip concatenation
 * (frame_transformer.py:194-197) and sequence slices. */
int dvt_copy2d(const void* src, void* dst, int64_t rows, int64_t cols, int64_t src_ld, int64_t dst_ld,
               int dtype, dvt_stream_t stream);
/* out[c] (+)= sum_{r<rows} src[r*row_stride + c]   (f32 out; gradient of a broadcast). */
int dvt_rows_sum(const void* src, int64_t row_stride, int64_t rows, int64_t cols, float* out,
                 int dtype, int accumulate, dvt_stream_t stream);
/* [A, B, C] -> [B, A, C]   ('b s d -> s b d', frame_transformer.py:205; transformer.py:76). */
int dvt_permute_021(const void* src, void* dst, int64_t A, int64_t B, int64_t C, int dtype,
                    dvt_stream_t stream);
/* dst(f32)[i] = beta * dst[i] + alpha * src[i]  (gradient accumulation). */
int dvt_axpby_f32(const void* src, int src_dtype, float alpha, float* dst, float beta, int64_t n,
                  dvt_stream_t stream);

/* Stand-alone activations (the 3-layer GELU head, src/models/frame_transformer.py:106,
 * where no GEMM epilogue is worth fusing into).  act: 1 = GELU(erf), 2 = ReLU, 3 = sigmoid (TPN.py:99).
 * fwd: y = act(x);  bwd: dx = dy * act'(x). */
int dvt_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, dvt_stream_t stream);
int dvt_act_bwd(const void* dy, const void* x, void* dx, int64_t n, int act, int dtype,
                dvt_stream_t stream);

/* Measurement aid: occupies the stream for `microseconds` (<= 1 s) with a single spinning lane, so that the host can
 * enqueue a whole step behind it and HIP-event brackets around individual launches then measure device time only
 * (bench.py's roofline pass). */
int dvt_device_delay(uint64_t microseconds, dvt_stream_t stream);
/* Asynchronous zero fill (a memset node inside a captured graph): the rows a gather's adjoint does not write --
 * `x[:, 0]` at src/models/vit.py:120,126 has a zero gradient on every other row. */
int dvt_zero(void* dst, size_t nbytes, dvt_stream_t stream);

/* ---------------------------------------------------------------- dropout
 * nn.Dropout(p) in training mode (frame_transformer.py:22,41-44 p = 0.5; TPN.py:92,95; vit.py:23,25,43,104):
 * y[i] = keep_i ? x[i] / (1 - p) : 0.  keep_i comes from Philox4x32-10 word (i & 3) of counter
 * rng_state[1] + call_offset + i / 4 under key rng_state[0]; rng_state is a device array {seed, step base offset}.
 * The same call with dy in place of x is the backward pass (no mask is stored).  call_offset separates the dropout
 * sites of one step (advance it by ceil(n / 4) per site); dvt_rng_advance moves the base once per step on the
 * device, so a captured hipGraph draws new masks on every replay.  torch's own generator stream is not reproduced. */
int dvt_dropout(const void* x, void* y, int64_t n, float p, const uint64_t* rng_state, uint64_t call_offset, int dtype,
                dvt_stream_t stream);
/* The same mask fused with its neighbours in nn.TransformerEncoderLayer's training-mode forward (frame_transformer.py:39-47:
 * x + dropout1(sa), dropout(relu(linear1 x)), x + dropout2(ff)):  y = residual? + keep * relu?(x) / (1 - p), residual nullable,
 * relu 0 / 1, mask drawn exactly like dvt_dropout at (rng_state, call_offset).  With rng_state == NULL and `gate` given (the
 * backward of the relu form, gate = the forward's output): y = gate != 0 ? x / (1 - p) : 0 -- no mask is re-drawn, an element
 * the mask dropped and one the ReLU zeroed both pass no gradient.  Exactly one of rng_state and gate is non-NULL.  The
 * backward of the residual form is dvt_dropout on dy (and dy itself for the residual). */
int dvt_dropout_fused(const void* x, const void* residual, const void* gate, void* y, int64_t n, float p,
                      const uint64_t* rng_state, uint64_t call_offset, int relu, int dtype, dvt_stream_t stream);
int dvt_rng_advance(uint64_t* rng_state, uint64_t delta, dvt_stream_t stream);

/* ---------------------------------------------------------------- patchify
 * Rearrange 'b t c (h p1) (w p2) -> b t (h w) (p1 p2 c)'  (src/models/vit.py:90).
 * x: [frames, C, H, W] in x_dtype; out: [frames * (H/P) * (W/P), P*P*C] in
 * out_dtype (patch-vector index = (p1*P + p2)*C + c). */
int dvt_patchify(const void* x, int x_dtype, void* out, int out_dtype, int64_t frames, int C,
                 int H, int W, int P, dvt_stream_t stream);
/* Adjoint of dvt_patchify (scatter back to pixel layout): needed when the clip
 * itself carries a gradient (learnable pixel-space CLS clip,
 * src/models/frame_transformer.py:105,195). */
int dvt_patchify_bwd(const void* dout, int dout_dtype, void* dx, int dx_dtype, int64_t frames,
                     int C, int H, int W, int P, dvt_stream_t stream);

/* ---------------------------------------------------------------- token assembly
 * ViViT.forward token plumbing, src/models/vit.py:113-115:
 *   out[s, 0, :]   = cls[:]            + pos[s % T, 0, :]
 *   out[s, 1+j, :] = emb[s*n + j, :]   + pos[s % T, 1+j, :]
 * emb: [S*n, d] dtype; cls: [d] f32; pos: [T, pos_rows, d] f32 (pos_rows >= n+1);
 * out: [S, n+1, d] dtype.  */
int dvt_tokens_assemble_fwd(const void* emb, const float* cls, const float* pos, void* out,
                            int64_t S, int64_t T, int64_t n, int64_t d, int64_t pos_rows,
                            int dtype, dvt_stream_t stream);
/* demb = dout[:, 1:, :]; dcls (+)= sum_s dout[s,0,:]; dpos[t,j,:] (+)= sum_{s%T==t} dout[s,j,:].
 * `accumulate` != 0 adds into dcls/dpos instead of overwriting. */
int dvt_tokens_assemble_bwd(const void* dout, void* demb, float* dcls, float* dpos, int64_t S,
                            int64_t T, int64_t n, int64_t d, int64_t pos_rows, int dtype,
                            int accumulate, dvt_stream_t stream);

/* Row gather with an optional leading token (vit.py:120-123, x[:, 0] then
 * cat(temporal_token, .)):  out[b, 0, :] = tok (if tok != NULL, then lead = 1)
 * out[b, lead + t, :] = src[(b*T + t) * src_row_stride : + d].  */
int dvt_rows_gather_fwd(const void* src, int64_t src_row_stride, const float* tok, void* out,
                        int64_t B, int64_t T, int64_t d, int dtype, dvt_stream_t stream);
/* dsrc rows (same addressing) = dout[b, lead+t, :]; dtok (+)= sum_b dout[b,0,:].
 * Rows of dsrc that are not gathered are NOT touched (caller zero-fills). */
int dvt_rows_gather_bwd(const void* dout, void* dsrc, int64_t src_row_stride, float* dtok,
                        int64_t B, int64_t T, int64_t d, int dtype, int accumulate,
                        dvt_stream_t stream);

/* Scaled sum over the middle dimension: out[b, :] = scale * sum_j x[b, j, :].  scale = 1/L is
 * `x.mean(dim=1)` (pool == 'mean', src/models/vit.py:126) and the global AvgPool2d of the feature
 * pyramid (src/models/TPN.py:6,20,33); scale = 1 is `sum_group` (TPN.py:64-72).
 * bwd: dx[b, j, :] = scale * dout[b, :]. */
int dvt_mean_rows_fwd(const void* x, void* out, int64_t B, int64_t L, int64_t d, float scale, int dtype,
                      dvt_stream_t stream);
int dvt_mean_rows_bwd(const void* dout, void* dx, int64_t B, int64_t L, int64_t d, float scale, int dtype,
                      dvt_stream_t stream);

/* ---------------------------------------------------------------- LayerNorm
 * nn.LayerNorm(d), eps 1e-5, affine: src/models/vit.py:11,64,105;
 * src/models/frame_transformer.py:117; torch TransformerEncoderLayer norm1/2.
 * Rows are indexed (i0, i1), i0 < n0, i1 < n1; row (i0,i1) of x starts at element
 * i0*xs0 + i1*xs1, of y at i0*ys0 + i1*ys1 (dense: n0 = rows, n1 = 1, xs0 = ys0 = d).
 * mean/rstd: [n0*n1] f32, written by fwd, read by bwd. */
int dvt_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                      float* rstd, int64_t n0, int64_t n1, int64_t d, int64_t xs0, int64_t xs1,
                      int64_t ys0, int64_t ys1, float eps, int dtype, dvt_stream_t stream);
size_t dvt_layernorm_bwd_workspace_bytes(int64_t d);
/* dy uses the y strides; x, dx and dx_add use the x strides.  dx = LN'(dy) (+ dx_add
 * when dx_add != NULL: the residual-branch gradient of `fn(LN(x)) + x`, vit.py:73-74,
 * so that the two gradient paths into x are summed inside this kernel).
 * dgamma/dbeta: [d] f32, overwritten (accumulate == 0) or added to.  workspace: at
 * least dvt_layernorm_bwd_workspace_bytes(d) bytes of device memory. */
int dvt_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                      const float* rstd, const void* dx_add, void* dx, float* dgamma,
                      float* dbeta, void* workspace,
                      int64_t n0, int64_t n1, int64_t d, int64_t xs0, int64_t xs1, int64_t ys0,
                      int64_t ys1, int dtype, int accumulate, dvt_stream_t stream);
/* The same with (a) two more operands that enter only the FIRST row of each group (i1 == 0), indexed by i0:
 * dy_first is added to dy before the backward (a second gradient path into LN(x)[:, 0]: the query projection
 * of a block whose output is read at the CLS row only, src/models/vit.py:119-120 `x[:, 0]`), dx_first is added
 * to dx (the residual path of that row); either may be NULL; and (b) separate accumulate flags for dgamma and
 * dbeta (the two may sit in different gradient buckets). */
int dvt_layernorm_bwd_first(const void* dy, const void* x, const float* gamma, const float* mean,
                            const float* rstd, const void* dx_add, void* dx, float* dgamma, float* dbeta,
                            void* workspace,
                            int64_t n0, int64_t n1, int64_t d, int64_t xs0, int64_t xs1, int64_t ys0,
                            int64_t ys1, const void* dy_first, int64_t dy_first_stride,
                            const void* dx_first, int64_t dx_first_stride, int dtype,
                            int accumulate_gamma, int accumulate_beta, dvt_stream_t stream);

/* The general forms.  _fwd_mixed: x and y may differ in element type -- fp32 on exactly one side, a 16-bit type on the
 * other: the launch-bound zone of the path (the 33-token temporal encoder of vit.py:122-128, 264 rows at B = 8) keeps its
 * residual stream and its row gradients in fp32 (there single rows carry whole gradients; storage there costs no
 * bandwidth), while the GEMM operands stay 16-bit.  _bwd_ex: dy / x / dx element types separately (all equal; or dy 16-bit,
 * x and dx fp32, plus dx_lp, a 16-bit copy of dx for the GEMMs behind; or dy fp32, x and dx 16-bit: the seam where the fp32
 * stream meets the space stack's rows); defer_reduce leaves the dgamma / dbeta reduce of the per-workgroup partial rows
 * undone and describes it in *pending -- dvt_layernorm_reduce_group then performs up to 32 of them in ONE launch (the zone
 * has a dozen LayerNorms whose reduces were a launch of ~5 us each).  The partial rows live in `workspace`
 * (>= dvt_layernorm_bwd_partial_bytes(rows, d)), which must stay untouched until that launch. */
typedef struct dvt_ln_pending {
  const float* partial;   /* [nparts][2][d] */
  int32_t nparts, d;
  float* dgamma;
  float* dbeta;
  int32_t accumulate;     /* bit 0: dgamma +=, bit 1: dbeta += */
  int32_t valid;
} dvt_ln_pending;
typedef struct dvt_ln_bwd_desc {
  const void* dy; int32_t dy_dtype;
  const void* x; int32_t x_dtype;
  const float* gamma; const float* mean; const float* rstd;
  const void* dx_add;                    /* dx_dtype, x strides; may be NULL */
  void* dx; int32_t dx_dtype;
  void* dx_lp; int32_t dx_lp_dtype;      /* optional second copy of dx, x strides */
  float* dgamma; float* dbeta; void* workspace;
  int64_t n0, n1, d, xs0, xs1, ys0, ys1;
  const void* dy_first; int64_t dy_first_stride;   /* dy_dtype */
  const void* dx_first; int64_t dx_first_stride;   /* dx_dtype */
  int32_t accumulate_gamma, accumulate_beta;
  int32_t defer_reduce;
  dvt_ln_pending* pending;
} dvt_ln_bwd_desc;
int dvt_layernorm_fwd_mixed(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype, float* mean,
                            float* rstd, int64_t n0, int64_t n1, int64_t d, int64_t xs0, int64_t xs1,
                            int64_t ys0, int64_t ys1, float eps, dvt_stream_t stream);
size_t dvt_layernorm_bwd_partial_bytes(int64_t rows, int64_t d);
int dvt_layernorm_bwd_ex(const dvt_ln_bwd_desc* desc, dvt_stream_t stream);
int dvt_layernorm_reduce_group(const dvt_ln_pending* list, int count, dvt_stream_t stream);

/* ---------------------------------------------------------------- GEMM family
 * C[M,N] = epilogue( sum_k A(m,k) * B(k,n) ).  One kernel family serves
 *   forward  Linear        y = x W^T          (A k-major, B = W[N,K] k-major)
 *   dgrad                  dx = dy W          (A k-major, B = W[K=N_out, N=K_in] mn-major)
 *   wgrad                  dW = dy^T x        (A, B mn-major; reduction over rows)
 * and so replaces nn.Linear at src/models/vit.py:20-25,39-43,91,106 and inside
 * torch's TransformerEncoderLayer (frame_transformer.py:41-44).
 *   a_kmajor: element (m,k) at A[m*lda + k]  else at A[k*lda + m]
 *   b_kmajor: element (k,n) at B[n*ldb + k]  else at B[k*ldb + n]
 */
enum dvt_epilogue {
  DVT_EPI_NONE = 0,          /* C = acc (+ bias) */
  DVT_EPI_GELU = 1,          /* C = gelu_erf(acc + bias); aux (if != NULL) = gelu_erf'(acc + bias), what DVT_EPI_DGELU takes */
  DVT_EPI_RELU = 2,          /* C = relu(acc + bias) */
  DVT_EPI_RESIDUAL = 3,      /* C = acc + bias + residual */
  DVT_EPI_DGELU = 4,         /* C = acc * aux          (aux = the derivative saved by DVT_EPI_GELU) */
  DVT_EPI_DRELU = 5          /* C = acc * (aux > 0)    (aux = saved post-activation) */
};

/* A split-K reduction that a dvt_gemm call left undone (desc.defer_reduce): C (+)= sum_z slab[z] in slice order, and
 * the fused bias gradient's partial rows.  It is handed to the NEXT dvt_gemm call of the same stream (desc.carry) --
 * the data gradient of the same Linear, src/models/vit.py:20-25,39-43 backward --, whose launch performs it in extra
 * workgroups at the END of its grid: a data-gradient launch is 1.5-6 rounds of tiles and leaves CUs idle in its last
 * round, which is where a reduce that reads 64 MB of slabs then runs instead of as a launch of its own.  A call that
 * cannot carry it (not the LDS-DMA kernel) launches it stand-alone first; dvt_splitk_reduce_pending does so explicitly.
 * The slabs live in the deferring call's workspace: it must stay untouched until the carrying call has been enqueued. */
typedef struct dvt_splitk_pending {
  const float* slab;       /* [splits][M][N] f32 */
  int32_t splits, valid;   /* valid == 0: nothing pending (the deferring call did not split K) */
  int64_t M, N;
  float* C;                /* f32 [M, ldc] */
  int64_t ldc;
  int32_t accumulate;
  int32_t cs_accumulate;
  const float* cs_slab;    /* [splits][M] f32 or NULL */
  float* cs_out;           /* [M] f32 */
  /* conv_taps > 0: the product is the weight gradient of a convolution, rows m = tap * conv_cin + ci, columns n = co, and
   * the reduce scatters it straight into the parameter's own layout C[co][ci][tap] (f32 [N][conv_cin][conv_taps], ldc
   * unused): no packed intermediate, no scatter launch behind the reduce. */
  int32_t conv_cin, conv_taps;
  /* channel-padded layers: only n < conv_cout_l and ci < conv_cin_l exist in the parameter f32 [conv_cout_l][conv_cin_l]
   * [conv_taps] (0: the GEMM's own N / conv_cin) */
  int32_t conv_cin_l, conv_cout_l;
} dvt_splitk_pending;

typedef struct dvt_gemm_desc {
  const void* A;
  const void* B;
  void* C;
  int64_t M, N, K;
  int64_t lda, ldb, ldc;
  int32_t a_kmajor, b_kmajor;
  int32_t in_dtype;     /* dtype of A, B, residual, aux */
  int32_t out_dtype;    /* dtype of C: in_dtype, or DVT_F32 (weight gradients) */
  int32_t epilogue;     /* enum dvt_epilogue */
  int32_t accumulate;   /* out_dtype == DVT_F32 only: C += result instead of C = result */
  const float* bias;    /* [N] f32 or NULL */
  const void* residual; /* [M, ldr] or NULL */
  int64_t ldr;
  void* aux;            /* [M, ldaux] or NULL */
  int64_t ldaux;
  float alpha;          /* result scale applied to acc before the epilogue (1.0f default) */
  int32_t split_k;      /* 0 = library chooses; >1 forces that many K slices */
  void* workspace;      /* >= dvt_gemm_workspace_bytes(desc) bytes (may be NULL if that is 0) */
  /* Optional, mn-major A only (weight gradients, A = dY): colsum_out[m] (+)= sum_k A(m,k),
   * i.e. the bias gradient, produced by the same pass over dY (an extra MFMA against an
   * all-ones fragment) instead of a separate reduction.  NULL = off. */
  float* colsum_out;
  int32_t colsum_accumulate;
  /* Optional (both may be NULL / 0).  defer_reduce: when this call splits K with the plain f32 reduce behind it
   * (weight gradients), leave the reduce undone and describe it in *pending (pending->valid = 0 when there is none).
   * carry: a pending reduce of an earlier call to perform with this call (see dvt_splitk_pending). */
  int32_t defer_reduce;
  dvt_splitk_pending* pending;
  const dvt_splitk_pending* carry;
  /* RESIDUAL epilogue with an fp32 residual [M, ldr] (and out_dtype DVT_F32): y32 = x W^T + b + r32 -- the residual stream of
   * the launch-bound zone (the 33-token temporal encoder) is kept in fp32 while the GEMM operands stay 16-bit.  Served by
   * the panel-streaming kernel (launch-bound shapes) only; other shapes report DVT_ERR_UNSUPPORTED. */
  int32_t residual_f32;
} dvt_gemm_desc;

size_t dvt_gemm_workspace_bytes(const dvt_gemm_desc* desc);
int dvt_gemm(const dvt_gemm_desc* desc, dvt_stream_t stream);
/* Which kernel family dvt_gemm would take for desc (no launch): 0 = the panel-streaming kernel of the launch-bound shapes
 * (the only one that serves residual_f32), 1 = the LDS-DMA / register-staged MFMA kernels, 2 = the generic fp32 kernel. */
int dvt_gemm_route(const dvt_gemm_desc* desc);
/* The two products of one Linear's backward (src/models/vit.py:20-25,39-43: dW = dy^T x with both operands mn-major,
 * dx = dy W with A k-major / B mn-major) as ONE launch when both are launch-bound shapes (the 33-token temporal encoder,
 * the CLS-row layers: a few hundred rows): they are independent, each fills a fraction of the chip, and between dependent
 * launches of a few microseconds the launch itself is the cost.  Any other pair: the two dvt_gemm calls one after the
 * other.  dvt_gemm_pair_fused tells which (1 = one launch). */
int dvt_gemm_pair_fused(const dvt_gemm_desc* wgrad, const dvt_gemm_desc* dgrad);
int dvt_gemm_pair(const dvt_gemm_desc* wgrad, const dvt_gemm_desc* dgrad, dvt_stream_t stream);
/* Performs a pending split-K reduction as a launch of its own. */
int dvt_splitk_reduce_pending(const dvt_splitk_pending* pending, dvt_stream_t stream);

/* out[n] (+)= sum_m x[m*ldx + n]  -- bias gradients. workspace >=
 * dvt_colsum_workspace_bytes(M, N). */
size_t dvt_colsum_workspace_bytes(int64_t M, int64_t N);
int dvt_colsum(const void* x, int64_t ldx, float* out, void* workspace, int64_t M, int64_t N,
               int dtype, int accumulate, dvt_stream_t stream);

/* ---------------------------------------------------------------- attention
 * o = softmax(q k^T * scale) v per (batch, head): src/models/vit.py:51-55, and
 * the core of nn.MultiheadAttention inside TransformerEncoderLayer
 * (frame_transformer.py:41-44).  Lq != Lk gives the cross-modal form
 * (frame_transformer.py:225-226 joint video/image tokens; BASELINE config 4).
 * Element (b, h, l, e) of q sits at q[b*q_sb + h*q_sh + l*q_sl + e]; likewise
 * k, v, o.  do/dq/dk/dv reuse the strides of o/q/k/v.  lse: [B, H, Lq] f32
 * (log-sum-exp of the scaled scores), written by fwd and read by bwd.
 * Scores are never materialised in HBM. */
typedef struct dvt_attn_desc {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  float* lse;
  const void* d_o; /* bwd only */
  void* dq;        /* bwd only */
  void* dk;
  void* dv;
  int64_t B, H, Lq, Lk, dh;
  int64_t q_sb, q_sh, q_sl;
  int64_t k_sb, k_sh, k_sl;
  int64_t v_sb, v_sh, v_sl;
  int64_t o_sb, o_sh, o_sl;
  float scale;
  int32_t dtype;
  void* workspace; /* bwd only: >= dvt_attention_bwd_workspace_bytes(desc) bytes (may be NULL if 0): holds
                    * delta = rowsum(dO * O), written by the dq pass and read by the dk/dv pass */
  /* Attention-probability dropout, nn.MultiheadAttention(dropout=p) in training mode (frame_transformer.py:41-44):
   * o = (softmax(s) * keep / (1 - p)) v, keep(b,h,i,j) drawn like dvt_dropout from rng_state (device {seed, base})
   * at rng_offset + (((b*H + h)*Lq + i)*Lk + j) / 4.  0 = off.  Served by the generic (non-MFMA) kernels. */
  float dropout_p;
  const uint64_t* rng_state;
  uint64_t rng_offset;
  /* bwd, 16-bit dh = 64 kernels: sequences whose two lengths pad to the same multiple of 32 <= 224 (the 197-token frames,
   * the 33-token clips) run as ONE pass -- Q, dO, K staged once, dK / dV kept in registers by key-owner waves, dQ formed by
   * a dedicated wave from 16-bit dS strips in LDS (5 MFMA products and 8 units of traffic instead of the 7 and 13 of the
   * dq + dk/dv pair).  bwd_two_pass != 0 forces the pair (longer sequences always use it). */
  int32_t bwd_two_pass;
} dvt_attn_desc;

size_t dvt_attention_bwd_workspace_bytes(const dvt_attn_desc* desc);
int dvt_attention_fwd(const dvt_attn_desc* desc, dvt_stream_t stream);
int dvt_attention_bwd(const dvt_attn_desc* desc, dvt_stream_t stream);

/* ---------------------------------------------------------------- single-query attention with folded K / V projections
 * The reference reads only row 0 of the space transformer's output (src/models/vit.py:119-120; :126 for the temporal
 * one under pool == 'cls'), so the attention of a stack's LAST layer (vit.py:46-58) has ONE query per (sequence, head),
 * and with one query the key / value projections of to_qkv (vit.py:39,48) commute with the attention sums:
 *     s_jh = scale q_h . (Wk_h LN(x_j)) = r_h . LN(x_j),   r_h = scale Wk_h^T q_h   (dvt_heads_expand)
 *     o_h  = sum_j p_jh Wv_h LN(x_j)    = Wv_h m_h,         m_h = sum_j p_jh LN(x_j) (dvt_heads_contract)
 * -- neither K, V nor LN(x) of the rows that are never read again is materialised: dvt_attn_cls_fwd is one pass over
 * the raw rows x[S][N][d] (LayerNorm statistics, H scores and H weighted sums per row), dvt_attn_cls_bwd one more, which
 * also is the LayerNorm backward of those rows (the gradient of row 0's own query / residual paths is added by a
 * dvt_layernorm_bwd call on the S first rows).  Row (s, j) of x sits at x + s*xs0 + j*xs1 (elements); dx uses the same
 * strides.  All [S, H, d] arrays are f32.  With n_j = (x_j - mean_j) rstd_j, LN(x_j) = gamma n_j + beta:
 *   fwd:  R -> A[s,h,:] = sum_j p_jh n_j (so m_h = gamma A_h + beta), lse[S,H], the probabilities P[S,N,8] (head h of row
 *         j at P[(s*N + j)*8 + h]; kept for the backward), mean / rstd [S*N]
 *   bwd:  dM (gradient of m) -> dx, G[s,h,:] = sum_j ds_jh n_j (so dr_h = gamma G_h), dgamma / dbeta (+= when the
 *         accumulate flags are set; workspace >= dvt_attn_cls_bwd_workspace_bytes holds one partial row per sequence).
 * 16-bit dtypes, d % 8 == 0, d <= 512, H <= 8, N <= 400 (sequences of more than 200 rows -- the 325-token frames of a 288^2
 * clip -- are walked in two chunks per pass); dvt_attn_cls_supported tells (callers keep the unfolded
 * dvt_layernorm_fwd -> dvt_gemm -> dvt_attention_fwd sequence otherwise). */
typedef struct dvt_attn_cls_desc {
  const void* x;
  int64_t xs0, xs1;
  const float* gamma;
  const float* beta;
  float eps;
  int64_t S, N, d, H;
  int32_t dtype;
  const float* R;
  float* A;
  float* lse;
  float* P;
  float* mean;
  float* rstd;
  /* backward only */
  const float* dM;
  void* dx;
  float* G;
  float* dgamma;
  float* dbeta;
  int32_t accumulate_gamma, accumulate_beta;
  void* workspace;
} dvt_attn_cls_desc;

int dvt_attn_cls_supported(const dvt_attn_cls_desc* desc);
size_t dvt_attn_cls_bwd_workspace_bytes(const dvt_attn_cls_desc* desc);
int dvt_attn_cls_fwd(const dvt_attn_cls_desc* desc, dvt_stream_t stream);
int dvt_attn_cls_bwd(const dvt_attn_cls_desc* desc, dvt_stream_t stream);
/* The head-wise products on either side of it, over S rows only; W is a row range of the packed to_qkv weight
 * (vit.py:39; [H*dh, ldw] in the 16-bit compute dtype), the [S, H, d] operands are f32 and, when gamma / beta are
 * given, enter as gamma * v + beta (either may be NULL):
 *   expand:   out[s, h, c]     = alpha sum_e in[s, h*dh + e] W[h*dh + e, c]                (q -> r, do -> dm)
 *   contract: out[s, h*dh + e] = alpha sum_c v[s, h, c] W[h*dh + e, c]                      (m -> o, dr -> dq)
 *   outer:    dW[h*dh + e, c] (+)= alpha sum_s a[s, h*dh + e] v[s, h, c]                    (weight gradients of Wv, Wk) */
int dvt_heads_expand(const void* in, int64_t ld_in, const void* W, int64_t ldw, float* out, int64_t S, int64_t H,
                     int64_t dh, int64_t d, float alpha, int dtype, dvt_stream_t stream);
int dvt_heads_contract(const float* in, const float* gamma, const float* beta, const void* W, int64_t ldw, void* out,
                       int64_t ld_out, int64_t S, int64_t H, int64_t dh, int64_t d, float alpha, int dtype,
                       dvt_stream_t stream);
int dvt_heads_outer(const void* a, int64_t lda, const float* b, const float* gamma, const float* beta, float* dW,
                    int64_t ldw, int64_t S, int64_t H, int64_t dh, int64_t d, float alpha, int accumulate, int dtype,
                    dvt_stream_t stream);
/* The backward runs them in two independent pairs, one launch each (workgroups of the first member in front):
 *   expand_outer:    out = alpha_out expand(in, W);          dW (+)= alpha_dw outer(in, gamma v + beta)     (dm and dWv from do)
 *   contract_outer:  out = alpha_out contract(gamma v, W);   dW (+)= alpha_dw outer(a, gamma v)             (dq and dWk from G) */
int dvt_heads_expand_outer(const void* in, int64_t ld_in, const void* W, int64_t ldw, float* out, float alpha_out,
                           const float* v, const float* gamma, const float* beta, float* dW, int64_t ld_dw, float alpha_dw,
                           int accumulate, int64_t S, int64_t H, int64_t dh, int64_t d, int dtype, dvt_stream_t stream);
int dvt_heads_contract_outer(const float* v, const float* gamma, const void* W, int64_t ldw, void* out, int64_t ld_out,
                             float alpha_out, const void* a, int64_t lda, float* dW, int64_t ld_dw, float alpha_dw,
                             int accumulate, int64_t S, int64_t H, int64_t dh, int64_t d, int dtype, dvt_stream_t stream);

/* ---------------------------------------------------------------- on-device input stage
 * transforms.Compose([Resize(resize), CenterCrop(crop), ToTensor(), Normalize(mean, std)]) of the reference's loader
 * (src/dataloaders/mmx/MMX_Light_dl.py:203-217; val_transform :195-201) applied to decoded uint8 RGB frames
 * src[F, H0, W0, 3] already resident in HBM; dst[F, 3, crop, crop] (f32: bit-exact with PIL + torch; bf16: that
 * value rounded to nearest even).  Resize = Pillow 8-bit bilinear resample (antialiased triangle filter, 22-bit fixed
 * point, horizontal then vertical pass through a uint8 intermediate in the workspace).  mean/std are host arrays of 3. */
size_t dvt_frames_preprocess_workspace_bytes(int64_t frames, int H0, int W0, int resize, int crop);
int dvt_frames_preprocess(const void* src, void* dst, int dst_dtype, int64_t frames, int H0, int W0, int resize,
                          int crop, const float* mean, const float* std, void* workspace, dvt_stream_t stream);

/* ---------------------------------------------------------------- multi-modal gating + contrastive loss
 * F.normalize(x) (collabgating.py:70) and the normaliser of F.cosine_similarity (ntxent.py:63):
 * y = x / max(||x||_2, eps) per row of [rows, D]; inv_norm[rows] is kept for backward. */
int dvt_l2norm_rows_fwd(const void* x, void* y, float* inv_norm, int64_t rows, int D, float eps, int dtype,
                        dvt_stream_t stream);
int dvt_l2norm_rows_bwd(const void* dy, const void* y, const float* inv_norm, void* dx, int64_t rows, int D, float eps,
                        int dtype, dvt_stream_t stream);
/* nn.CosineSimilarity(dim=1)(a, b) (frame_transformer.py:121, logged at :257): out[r] f32, no gradient. */
int dvt_cosine_rows(const void* a, const void* b, float* out, int64_t rows, int D, float eps, int dtype,
                    dvt_stream_t stream);
/* ContextGating: F.glu(cat(x, x + x1), -1) = x * sigmoid(x + x1) (collabgating.py:83-86): y = a * sigmoid(b). */
int dvt_gate_fwd(const void* a, const void* b, void* y, int64_t n, int dtype, dvt_stream_t stream);
int dvt_gate_bwd(const void* dy, const void* a, const void* b, void* da, void* db, int64_t n, int dtype,
                 dvt_stream_t stream);
/* ContrastiveLoss.forward on the cosine-similarity matrix sim[M, M] f32, M = 2 * batch (ntxent.py:63-75):
 * loss = mean_k( log sum_{j != k} exp(sim_kj / T) - sim_{k, (k + M/2) mod M} / T ).  row_lse[M], row_loss[M]: scratch /
 * saved for backward.  bwd: dsim = dloss/dsim scaled by gloss[0]. */
int dvt_contrastive_fwd(const float* sim, int M, float temperature, float* loss, float* row_lse, float* row_loss,
                        dvt_stream_t stream);
int dvt_contrastive_bwd(const float* sim, const float* row_lse, int M, float temperature, const float* gloss,
                        float* dsim, dvt_stream_t stream);

/* ---------------------------------------------------------------- evaluation reductions
 * scikit-learn calls of src/callbacks/callbacks.py:36-55 on running_logits / running_labels, on the device.
 * probs [N, C] f32 (sigmoid outputs, frame_transformer.py:331), labels [N, C] u8 (0 / non-zero).
 * dvt_f1_samples: out[j] = f1_score(labels, probs > thresholds[j], average="samples", zero_division=0);
 * thresholds is a host array of T <= 16 values (callbacks.py:38: 0, 0.1 .. 0.8).
 * dvt_average_precision: average_precision_score(labels, probs, average="samples" / "weighted" / None) ->
 * out_samples[1], out_weighted[1], out_per_class[C] (device f32); C <= 64. */
size_t dvt_f1_samples_workspace_bytes(int64_t N, int T);
int dvt_f1_samples(const float* probs, const unsigned char* labels, int64_t N, int C, const float* thresholds, int T,
                   float* out, void* workspace, dvt_stream_t stream);
size_t dvt_average_precision_workspace_bytes(int64_t N, int C);
int dvt_average_precision(const float* probs, const unsigned char* labels, int64_t N, int C, float* out_samples,
                          float* out_weighted, float* out_per_class, void* workspace, dvt_stream_t stream);

/* ---------------------------------------------------------------- per-frame CNN encoder
 * src/models/custom_resnet.py:19-153 (conv3x3 / 7x7 stem / 1x1 downsample, BatchNorm2d, ReLU,
 * MaxPool2d(3,2,1), residual adds).  Feature maps are NHWC = [N*H*W, C] matrices, so a
 * convolution is  dvt_im2col -> dvt_gemm -> [N*Ho*Wo, Cout];  weights nn.Conv2d [Cout,Cin,kh,kw]. */
/* out[(n,ho,wo), (ki*kw+kj)*C + c] = x[n, ho*sh-ph+ki, wo*sw-pw+kj, c] (0 outside); columns
 * [kh*kw*C, ld) are zero padding.  x is NCHW when x_nchw != 0 (the raw clip frames), else NHWC.
 * Rectangular kernels / strides / paddings also serve the factorised R(2+1)D convolutions
 * (frame_transformer.py:67): (1,3,3) spatial = per-frame 2-D conv; (3,1,1) temporal = a (3,1) conv over
 * the [T, H*W] view of each clip. */
int dvt_im2col(const void* x, int x_dtype, int x_nchw, void* out, int out_dtype, int64_t N, int C, int H,
               int W, int kh, int kw, int sh, int sw, int ph, int pw, int64_t ld, dvt_stream_t stream);
/* Raw NCHW frames x[N, C, H, W] (custom_resnet.py:138: the stem reads the clip frames) -> NHWC matrix y[N*H*W, Cpad] with
 * the channels C .. Cpad-1 zero.  Cpad = 8: one 16-byte chunk per pixel (C <= 8).  Cpad = 4 (C <= 4, W even): one chunk per
 * pair of horizontally adjacent pixels, i.e. the [N, H, W/2, 8] view in which a stride-2 stem (custom_resnet.py:100,
 * frame_transformer.py:67) is a stride-(sh, 1) convolution over pixel pairs with the weights of dvt_conv_weight_pairs --
 * 28 instead of 49 gathered chunks per output pixel (7 rows x 4 pixel pairs) and half the zero padding of the C = 8 form. */
int dvt_nchw_to_nhwc_pad(const void* x, int x_dtype, void* y, int y_dtype, int64_t N, int C, int H, int W, int Cpad,
                         dvt_stream_t stream);
/* Stem weights for the pixel-pair view: w[Cout, Cin <= 4, kh, kw] f32 -> wp[Cout, 8, kh, kwp] f32 with
 * wp[co, px*4 + c, ki, p] = w[co, c, ki, 2p - (pw & 1) + px], zero outside the kernel / beyond Cin (pw = the horizontal
 * padding of the stride-2 convolution; the pair convolution has kernel (kh, kwp), stride (sh, 1), padding
 * (ph, (pw + (pw & 1)) / 2)).  _bwd: the adjoint gather of the weight gradient (+= when accumulate). */
int dvt_conv_weight_pairs(const float* w, float* wp, int Cout, int Cin, int kh, int kw, int pw, int kwp, dvt_stream_t stream);
int dvt_conv_weight_pairs_bwd(const float* dwp, float* dw, int Cout, int Cin, int kh, int kw, int pw, int kwp, int accumulate,
                              dvt_stream_t stream);
/* Adjoint gather (data gradient of the convolution), NHWC.  Optional second gradient path into the same map, summed in
 * the same pass (a residual block's shortcut joining its first convolution's data gradient, custom_resnet.py:38-54):
 * add [N*H*W, C] when add_stride == 0; else the COMPACT gradient [N*ceil(H/s)*ceil(W/s), C] of the map's stride-s
 * subsampling (the input of a strided 1x1 downsample convolution), added at the pixels with h % s == 0 and w % s == 0. */
int dvt_col2im(const void* dcol, void* dx, int64_t N, int C, int H, int W, int kh, int kw, int sh, int sw,
               int ph, int pw, int64_t ld, const void* add, int add_stride, int dtype, dvt_stream_t stream);
/* Same adjoint written in NCHW order and in the clip's own dtype: the gradient of the stem w.r.t. the raw
 * frames (needed by the learnable pixel-space CLS clip, frame_transformer.py:105,195). */
int dvt_col2im_nchw(const void* dcol, int dtype, void* dx, int dx_dtype, int64_t N, int C, int H, int W, int kh,
                    int kw, int sh, int sw, int ph, int pw, int64_t ld, dvt_stream_t stream);
/* Implicit-GEMM convolution: y[N*Ho*Wo, Cout] = sum_{ki,kj,c} x[n, ho*sh-ph+ki, wo*sw-pw+kj, c] * w[co, (ki*kw+kj)*C + c]
 * with the gather fused into the GEMM's A-operand LDS-DMA -- no column matrix in HBM (nn.Conv2d of
 * custom_resnet.py:19-22,128; the (1,3,3)/(3,1,1) convolutions of R(2+1)D).  x NHWC, w = dvt_conv_weight_pack(...,
 * ld = kh*kw*C).  The data gradient of a stride-1 convolution is the same call on dz with the 180-degree-rotated,
 * channel-transposed weights and padding k-1-p.  16-bit dtypes, C % 64 == 0 (C % 32 == 0 when Cout <= 128),
 * Cout % 8 == 0; dvt_conv2d_implicit_supported tells (callers fall back to dvt_im2col + dvt_gemm otherwise). */
/* Channel padding of f32 parameter arrays viewed as [A, B, K] (conv weights: A = Cout, B = Cin, K = kh*kw; BatchNorm
 * vectors: B = K = 1): dst[Ap, Bp, K] = src zero-extended; dvt_unpad3_f32 is the adjoint slice (+= when accumulate).
 * Lets layers whose channel counts are not multiples of 8 (R(2+1)D mid planes 45 / 230 / 460 / 921,
 * frame_transformer.py:67) run on the MFMA kernels with exactly-zero padded channels. */
int dvt_pad3_f32(const float* src, float* dst, int A, int B, int K, int Ap, int Bp, dvt_stream_t stream);
int dvt_unpad3_f32(const float* src, float* dst, int A, int B, int K, int Bp, int accumulate, dvt_stream_t stream);
/* w[Cout,Cin,kh,kw] f32 -> dst[Cin, (kh*kw)*Cout]: taps rotated by 180 degrees, channels transposed (the weight
 * operand of the data-gradient convolution, see dvt_conv2d_implicit). */
int dvt_conv_weight_pack_dgrad(const float* w, void* dst, int dst_dtype, int Cout, int Cin, int kh, int kw,
                               dvt_stream_t stream);
/* Both packed forms of many convolution weights in ONE launch -- the weights change once per optimizer step, so a training
 * step needs one such launch instead of two per layer -- with the zero extension of channel-padded layers (cout_p >= cout_l,
 * cin_p >= cin_l) folded in.  kind 0: dvt_conv_weight_pack's layout dst[cout_p][ld] (column (ki*kw+kj)*cin_p + ci, zeros
 * beyond); kind 1: dvt_conv_weight_pack_dgrad's dst[cin_p][kh*kw*cout_p] (rotated taps, transposed channels). */
typedef struct dvt_pack_entry {
  const float* src;      /* f32 [cout_l][cin_l][kh*kw] */
  void* dst;
  int32_t cout_l, cin_l, kh, kw, cout_p, cin_p, ld, kind, dtype;
  /* kind 2 (ABI v4): the data-gradient operand of ONE parity class of a strided convolution -- dst[cin_p][nth*ntw*cout_p]
   * holding only the taps ki = cls_rh + j * cls_sh (j < nth = ceil((kh - cls_rh) / cls_sh)), kj likewise, in DECREASING
   * ki / kj order (so that tap t of the class reads dz row hq + t - pad'), channels transposed as in kind 1.  The class of
   * input pixels hi % sh == a uses cls_rh = (a + ph) % sh. */
  int32_t cls_sh, cls_sw, cls_rh, cls_rw;
} dvt_pack_entry;
int dvt_conv_weight_pack_group(const dvt_pack_entry* entries, int count, dvt_stream_t stream);
typedef struct dvt_conv_desc {
  const void* x;
  const void* w;
  void* y;
  int64_t N;
  int32_t H, W, C, Cout, kh, kw, sh, sw, ph, pw;
  int32_t dtype;
  void* workspace;   /* weight gradient; forward / data gradient when dvt_conv2d_implicit_workspace_bytes(desc) > 0 */
  /* forward only, optional: the BatchNorm that follows the convolution (custom_resnet.py:30,33,104: conv -> bn) needs the
   * column sums and sums of squares of y; when stats_partial != NULL (>= dvt_conv2d_implicit_stats_bytes(desc)) the GEMM
   * epilogue leaves them here from its fp32 accumulators, per 128-row block: [parts][2][Cout] f32 with
   * parts = dvt_conv2d_implicit_stats_parts(desc); dvt_bn_stats_from_partials turns them into mean / invstd.  Saves the
   * separate statistics pass over y. */
  float* stats_partial;
  /* output columns dropped at the right edge: Wo = (W + 2*pw - kw)/sw + 1 - trim_w (0: the usual convolution).  The
   * pixel-pair stem (dvt_conv_weight_pairs) pads two pair columns on the left and needs only one on the right. */
  int32_t trim_w;
  /* Optional, like the same three fields of the GEMM descriptor: the weight gradient (dvt_conv2d_implicit_wgrad) leaves its split-K
   * reduce undone when defer_reduce != 0 and describes it in *pending; the data gradient of the same layer
   * (dvt_conv2d_implicit on dz with the rotated weights) performs a pending reduce handed to it as `carry` in extra
   * workgroups at the end of its grid.  The weight-gradient scatter (dvt_conv_weight_unpack_grad_t) goes behind the carrier. */
  int32_t defer_reduce;
  dvt_splitk_pending* pending;
  const dvt_splitk_pending* carry;
  /* forward / data gradient, optional: y = conv(x) + residual, residual [N*Ho*Wo, Cout] in the map's dtype, added on the fp32
   * accumulators (the shortcut's gradient joining a block input's gradient, custom_resnet.py:38-54: no add kernel). */
  const void* residual;
  /* weight gradient, optional: y is the parameter's own gradient f32 [Cout][C][kh][kw] (+= when wgrad_accumulate) instead
   * of the packed dWt -- the split-K reduce scatters into it (dvt_splitk_pending.conv_taps). */
  int32_t wgrad_master_layout, wgrad_accumulate;
  int32_t wgrad_cout_l, wgrad_cin_l;   /* channel-padded layers: the parameter's own channel counts (0: Cout / C) */
  /* forward / data gradient, optional (ABI v4).  The data gradient of a STRIDED convolution (custom_resnet.py:19-22 with
   * stride 2, the 1 x 1 / 2 downsample of :124-130; R(2+1)D's strided halves) as implicit launches: the input pixels
   * (hi, wi) fall into sh x sw parity classes (hi % sh, wi % sw), and the gradient of one class is a stride-1 convolution of
   * dz with the class's taps (1 / 2 / 2 / 4 of a 3 x 3 / 2 filter: dvt_pack_entry kind 2), whose output rows are SCATTERED
   * into the full-size gradient.  out_h / out_w (> 0): the launch computes out_h x out_w output pixels per image instead of
   * the usual (H + 2 ph - kh) / sh + 1 (taps that fall outside the map read zeros, as padding does).  out_rows (int32
   * [N * out_h * out_w]): row m of the launch is row out_rows[m] of y (and of `residual`, unless residual_compact: then
   * residual row m joins -- the compact gradient of a strided 1 x 1 shortcut, which touches one class only).  No column
   * matrix, no col2im pass. */
  int32_t out_h, out_w;
  const int32_t* out_rows;
  int32_t residual_compact;
} dvt_conv_desc;
/* 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels (layer 1 of ResNet-18, custom_resnet.py:19-22,
 * 109; also its data gradient, with the rotated weights of dvt_conv_weight_pack_dgrad) from an LDS-resident halo patch: a
 * workgroup stages the (R + 2) x (W + 2) input patch of R whole output rows once and the nine taps read it from LDS, the
 * 64 x 576 weights stay resident (the implicit GEMM gathers every input element nine times from L2).  x, y NHWC
 * [N*H*W, 64], w [64][576] (dvt_conv_weight_pack, ld = 576).  stats_partial (optional): [dvt_conv3x3_c64_stats_parts + 64]
 * [2][64] f32 partial column sums for dvt_bn_stats_from_partials.  _supported: 16-bit dtype and W <= 56-ish (two patches
 * must fit beside the weights in 160 KiB); callers fall back to dvt_conv2d_implicit otherwise. */
int dvt_conv3x3_c64_supported(int64_t N, int H, int W, int dtype);
/* Weight gradient of the same layer from LDS-resident halo patches: a workgroup of the persistent grid stages the input
 * patch and the gradient tile of R output rows once each (the implicit form gathers the input nine times), accumulates
 * [co 64][tap * 64 + ci] over its tile sequence in registers and leaves ONE fp32 partial per workgroup in `workspace`
 * (dvt_conv3x3_c64_wgrad_workspace_bytes); the split-K reduce of the family sums the partials into the parameter's own
 * layout dw f32 [64][64][3][3] (+= when accumulate) -- right away, or (defer_reduce) described in *pending for a later
 * dvt_splitk_reduce_pending / a carrying launch.  x, dz NHWC [N*H*W, 64]. */
int dvt_conv3x3_c64_wgrad_supported(int64_t N, int H, int W, int dtype);
size_t dvt_conv3x3_c64_wgrad_workspace_bytes(int64_t N, int H, int W);
int dvt_conv3x3_c64_wgrad(const void* x, const void* dz, float* dw, void* workspace, int64_t N, int H, int W, int accumulate,
                          int defer_reduce, dvt_splitk_pending* pending, int dtype, dvt_stream_t stream);
/* The same with Cout output channels (>= 64 in steps of 16: the 64 -> 144 spatial half of R(2+1)D-18's layer-1 Conv2Plus1D,
 * video_resnet.py:25): the kernel runs once per group of dz channels -- groups of 64, the last one up to 80 wide (144 = 64 +
 * 80: a 16-channel launch of its own would re-stage every input patch for a ninth of the work) --, each group's partials
 * summed into its rows of dw f32 [Cout][64][3][3] before the next launch reuses the workspace (the size
 * dvt_conv3x3_c64_wgrad_workspace_bytes reports covers the widest group); defer_reduce defers the LAST group's reduce. */
int dvt_conv3x3_c64_wgrad_wide(const void* x, const void* dz, float* dw, void* workspace, int64_t N, int H, int W, int Cout,
                               int accumulate, int defer_reduce, dvt_splitk_pending* pending, int dtype, dvt_stream_t stream);
/* The temporal half of R(2+1)D-18's layer-1 Conv2Plus1D (torchvision r2plus1d_18 as used by frame_transformer.py:64-74: a
 * (3, 1, 1) convolution, 144 mid planes -> 64, stride 1, pad 1) from LDS-resident sliding windows: a workgroup stages a
 * segment of S pixels of one clip over all T + 2 frames once and the three taps read it at three position offsets.
 * x [N, T, L, 144] (L = H * W pixels per frame), w [64][ldw] (dvt_conv_weight_pack, column kt * 144 + ci), y / dz
 * [N, T, L, 64], dw f32 [64][144][3] (the Conv3d parameter's own layout; += when accumulate).
 * x_affine (optional, both entry points): x is the OUTPUT z of the spatial convolution in front and the BatchNorm (+ ReLU)
 * between the two halves -- y = relu(z * invstd * gamma + (beta - mean * invstd * gamma)), the formula and rounding of
 * dvt_bn_apply_fwd -- is applied to the staged window: the normalised 144-plane activation never exists in HBM.
 * stats_partial (forward): [dvt_conv3x1_fwd_stats_parts + 64][2][64] f32 partial column sums of the stored output for
 * dvt_bn_stats_from_partials.  workspace / defer_reduce / pending (weight gradient) as dvt_conv3x3_c64_wgrad.
 * The forward also takes Cin = 64 (since ABI v5): the temporal half of the STEM, Conv3d(45, 64, (3, 1, 1)) with its 45 mid
 * planes stored zero-padded to 64 -- x [N, T, L, 64], w [64][ldw] with column kt * 64 + ci, no x_affine -- and, given the
 * data-gradient pack of the weights (dvt_conv_weight_pack_dgrad), that layer's data gradient. */
typedef struct dvt_bn_affine {
  const float* mean;     /* [C] */
  const float* invstd;   /* [C] */
  const float* gamma;    /* [c_valid] */
  const float* beta;     /* [c_valid] */
  int32_t c_valid;       /* channels gamma / beta have (0: all) */
  int32_t relu;
} dvt_bn_affine;
int dvt_conv3x1_fwd_supported(int64_t N, int T, int L, int Cin, int Cout, int dtype);
int64_t dvt_conv3x1_fwd_stats_parts(int64_t N, int T, int L, int Cin);
int dvt_conv3x1_fwd(const void* x, const dvt_bn_affine* x_affine, const void* w, int64_t ldw, void* y, float* stats_partial,
                    int64_t N, int T, int L, int Cin, int dtype, dvt_stream_t stream);
int dvt_conv3x1_wgrad_supported(int64_t N, int T, int L, int Cin, int Cout, int dtype);
size_t dvt_conv3x1_wgrad_workspace_bytes(int64_t N, int T, int L);
int dvt_conv3x1_wgrad(const void* x, const dvt_bn_affine* x_affine, const void* dz, float* dw, void* workspace, int64_t N, int T,
                      int L, int accumulate, int defer_reduce, dvt_splitk_pending* pending, int dtype, dvt_stream_t stream);
/* The 7x7 / stride 2 / pad 3 stem on 3-channel frames, 64 output channels (reference custom_resnet.py:100 `conv1`; the
 * (1, 7, 7) spatial half of torchvision's R(2+1)D stem behind frame_transformer.py:64-74, its 45 planes zero-extended to 64),
 * from an LDS halo patch with the weights in registers.  x_pairs: the pixel-pair map [N, H, Wp = W / 2, 8] of
 * dvt_nchw_to_nhwc_pad(.., 4); w: [64][ldw] with column (ki * 4 + kj) * 8 + c8 (dvt_conv_weight_pairs + dvt_conv_weight_pack,
 * ldw >= 224); y: [N * (H / 2) * Wp, 64] -- the same result as dvt_conv2d_implicit with the pair geometry (kernel (7, 4),
 * stride (2, 1), pad (3, 2), trim_w 1).  stats_partial: [dvt_conv_stem7_stats_parts + 64][2][64] or NULL. */
int dvt_conv_stem7_supported(int64_t N, int H, int Wp, int dtype);
int64_t dvt_conv_stem7_stats_parts(int64_t N, int H, int Wp);
int dvt_conv_stem7(const void* x_pairs, const void* w, int64_t ldw, void* y, float* stats_partial, int64_t N, int H, int Wp,
                   int dtype, dvt_stream_t stream);
int64_t dvt_conv3x3_c64_stats_parts(int64_t N, int H, int W);
int dvt_conv3x3_c64(const void* x, const void* w, void* y, float* stats_partial, const void* residual, int64_t N, int H, int W,
                    int dtype, dvt_stream_t stream);   /* residual (optional): added to the output rows, like dvt_conv_desc.residual */
/* The same convolution where one side has 144 channels -- the spatial half of R(2+1)D-18's layer-1 Conv2Plus1D
 * (video_resnet.py: 64 -> 144 mid planes forward, 144 -> 64 as the data gradient with the rotated weights): the weights
 * (162 KiB) cannot stay in LDS, so they stream through a ring of 18 KiB stages issued by a producer wave while seven
 * compute waves run the taps of the staged patch (per 48-channel chunk at 144 input channels).  (Cin, Cout) = (64, 144)
 * or (144, 64) -- and (128, 288) / (288, 128), the same pair in layer 2, as one launch per 144- / 64-wide group of output
 * channels; x [N*H*W, Cin], w [Cout][9 * Cin] (k = tap * Cin + c), y [N*H*W, Cout].  stats_partial: only with
 * Cout = 144 / 288, [dvt_conv3x3_stream_stats_parts + 64][2][Cout]; residual: only with Cout = 64 / 128. */
int dvt_conv3x3_stream_supported(int64_t N, int H, int W, int Cin, int Cout, int dtype);
int64_t dvt_conv3x3_stream_stats_parts(int64_t N, int H, int W, int Cin, int Cout);
int dvt_conv3x3_stream(const void* x, const void* w, void* y, float* stats_partial, const void* residual, int64_t N, int H, int W,
                       int Cin, int Cout, int dtype, dvt_stream_t stream);
/* The (3, 1) sibling over the [T, L] view of N clips (L = H * W pixels per frame; rows of the "image" are L pixels apart): the
 * temporal half of the same Conv2Plus1D as its data gradient, 64 -> 144 channels, stride 1, pad (1, 0).  x [N*T*L, 64],
 * w [144][3 * 64], y [N*T*L, 144].  A tile is R frames of one column segment (a divisor of L); three weight stages per tile. */
int dvt_conv3x1_stream_supported(int64_t N, int T, int L, int Cin, int Cout, int dtype);
int dvt_conv3x1_stream(const void* x, const void* w, void* y, int64_t N, int T, int L, int Cin, int Cout, int dtype,
                       dvt_stream_t stream);
/* The same data gradient with the backward of the BatchNorm (+ ReLU) in FRONT of the temporal half fused in (the mid-plane
 * BatchNorm of Conv2Plus1D, video_resnet.py:30-31 of torchvision's layout; reference frame_transformer.py:64-74): dy [N*T*L, 64]
 * is the temporal convolution's output gradient, z [N*T*L, 144] the spatial half's output that `bn` normalised, dz [N*T*L, 144]
 * receives gamma * invstd * (d - sum d / rows - xhat * sum d xhat / rows) with d = (dy (*) w) under the ReLU mask recomputed
 * from z (training == 0: gamma * invstd * d); dgamma / dbeta [144] overwritten or accumulated.  The data gradient itself is
 * never stored: it is computed twice (sums, then the corrected gradient), which replaces a write and two reads of a
 * 144-plane map by one read of the 64-plane dy.  Same geometry as dvt_conv3x1_stream (dvt_conv3x1_stream_supported). */
size_t dvt_conv3x1_stream_bn_bwd_workspace_bytes(int64_t N, int T, int L);
int dvt_conv3x1_stream_bn_bwd(const void* dy, const void* w, const void* z, const dvt_bn_affine* bn, void* dz, float* dgamma,
                              float* dbeta, void* workspace, int64_t N, int T, int L, int training, int accumulate, int dtype,
                              dvt_stream_t stream);
int dvt_conv2d_implicit_supported(const dvt_conv_desc* desc);
/* Row length K of the packed weights w[Cout][K] the forward call expects: kh*kw*C -- rounded up to the kernel's k-tile in
 * the stem form C == 8 (the columns past kh*kw*8 must be zero: dvt_conv_weight_pack with ld = K writes them so). */
int64_t dvt_conv2d_implicit_k(const dvt_conv_desc* desc);
/* Scratch a forward / data-gradient launch of this descriptor needs (0 for most shapes): launches with few output rows and a
 * deep reduction (R(2+1)D layers 3 - 4) split K over workgroups into fp32 slabs in desc->workspace and sum them in a second
 * launch that also rounds, adds `residual` and leaves the BatchNorm partial sums (same contract as the one-launch form). */
size_t dvt_conv2d_implicit_workspace_bytes(const dvt_conv_desc* desc);
int dvt_conv2d_implicit(const dvt_conv_desc* desc, dvt_stream_t stream);
int64_t dvt_conv2d_implicit_stats_parts(const dvt_conv_desc* desc);
size_t dvt_conv2d_implicit_stats_bytes(const dvt_conv_desc* desc);
/* Weight gradient without a column matrix: with x = the layer input (NHWC), w = dz [N*Ho*Wo, Cout] and
 * y = dWt f32 [kh*kw*C, Cout] (overwritten): dWt[(ki*kw+kj)*C + c, co] = sum_rows gather(x)[row, .] * dz[row, co];
 * split-K over the rows with a fixed-order reduction (reproducible).  C % 8 == 0, Cout % 8 == 0, any number of output
 * pixels (rows past the end of the last k-tile are read as zeros).  dvt_conv_weight_unpack_grad_t scatters dWt into the
 * [Cout, Cin, kh, kw] master; or see wgrad_master_layout below. */
int dvt_conv2d_implicit_wgrad_supported(const dvt_conv_desc* desc);
size_t dvt_conv2d_implicit_wgrad_workspace_bytes(const dvt_conv_desc* desc);
int dvt_conv2d_implicit_wgrad(const dvt_conv_desc* desc, dvt_stream_t stream);
int dvt_conv_weight_unpack_grad_t(const float* gt, float* dw, int Cout, int Cin, int kh, int kw, int accumulate,
                                  dvt_stream_t stream);
/* w[Cout,Cin,kh,kw] f32 -> dst[Cout, ld] (column order (ki,kj,ci), zero padded) in dst_dtype, and the
 * inverse for the fp32 weight gradient (dw (+)= g re-ordered). */
int dvt_conv_weight_pack(const float* w, void* dst, int dst_dtype, int Cout, int Cin, int kh, int kw, int64_t ld,
                         dvt_stream_t stream);
int dvt_conv_weight_unpack_grad(const float* g, float* dw, int Cout, int Cin, int kh, int kw, int64_t ld,
                                int accumulate, dvt_stream_t stream);
/* nn.BatchNorm2d over the rows of x[rows, C] (custom_resnet.py:30,33,104).  Training: batch mean and
 * biased variance -> mean / invstd, running statistics updated with `momentum` (unbiased variance),
 * as torch does.  Eval: dvt_bn_eval_invstd from running_var.  workspace >= dvt_bn_workspace_bytes. */
size_t dvt_bn_workspace_bytes(int64_t rows, int C);
/* c_valid (every BatchNorm entry that takes it): the channels the parameter-side arrays really have -- gamma, beta, the
 * running statistics, dgamma, dbeta are [c_valid]; mean / invstd and the maps are [C] wide, C >= c_valid.  The layers of
 * R(2+1)D whose plane counts (45, 230, 460, 921) are zero-extended to multiples of 64 run at the padded width without
 * padded copies of their BatchNorm vectors: channels >= c_valid behave as gamma = beta = 0.  c_valid <= 0 means C. */
int dvt_bn_stats(const void* x, float* mean, float* invstd, float* running_mean, float* running_var,
                 void* workspace, int64_t rows, int C, int c_valid, float eps, float momentum, int dtype, dvt_stream_t stream);
/* Same result from the per-block partial sums a convolution left behind (dvt_conv_desc.stats_partial, a buffer of
 * dvt_conv2d_implicit_stats_bytes = parts + 64 rows of 2 * C floats: with more than 256 partial rows the call WRITES
 * the 64-row tail -- it folds the partial rows into it before the final sum, hence the non-const pointer). */
int dvt_bn_stats_from_partials(float* partial, int64_t parts, float* mean, float* invstd, float* running_mean,
                               float* running_var, int64_t rows, int C, int c_valid, float eps, float momentum,
                               dvt_stream_t stream);
int dvt_bn_eval_invstd(const float* running_var, float* invstd, int C, float eps, dvt_stream_t stream);
/* y = relu?((x-mean)*invstd*gamma + beta (+ residual)): `out += residual; relu` fused (custom_resnet.py:51-52).
 * relu_mask (optional, C % 8 == 0): uint8 [rows][C/8], bit k of byte (r, g) = y[r, 8g + k] > 0 -- the ReLU mask the backward
 * of a layer WITH a residual branch needs (its output cannot be recomputed from x alone); dvt_bn_bwd reads these bytes
 * (1/16 of the traffic) instead of the output y in both of its passes. */
int dvt_bn_apply_fwd(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                     const void* residual, void* y, void* relu_mask, int64_t rows, int C, int c_valid, int relu, int dtype,
                     dvt_stream_t stream);
/* dz = dy*(relu mask); dres = dz (if dres != NULL); dgamma/dbeta (+)=; dx by the batch-statistics formula (training) or
 * gamma*invstd*dz (eval).  The ReLU mask comes from relu_mask (dvt_bn_apply_fwd's bytes) when given, else from the
 * forward output y, else -- a ReLU layer without a residual branch -- it is recomputed from x as
 * (x - mean)*invstd*gamma + beta > 0 (beta required). */
int dvt_bn_bwd(const void* dy, const void* x, const void* y, const void* relu_mask, const float* mean, const float* invstd,
               const float* gamma, const float* beta, void* dx, void* dres, float* dgamma, float* dbeta, void* workspace,
               int64_t rows, int C, int c_valid, int relu, int training, int accumulate, int dtype, dvt_stream_t stream);
/* The ResNet stem's bn1 -> relu -> maxpool(3, 2, 1) (custom_resnet.py:100-105,138-142) without the normalised map in HBM:
 * _fwd reads the convolution output z once and writes the pooled map y[N*Ho*Wo, C] and the argmax taps idx (same values
 * and taps as dvt_bn_apply_fwd followed by dvt_maxpool_fwd); dvt_bn_bwd_pooled is dvt_bn_bwd whose incoming gradient
 * is gathered from the pooled gradient dy_pool and idx on the fly (the ReLU mask is recomputed from z), so neither the
 * max-pool's input gradient nor the BatchNorm output exist as tensors.  C % 8 == 0. */
int dvt_bn_relu_maxpool_fwd(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                            void* y, void* idx, int64_t N, int C, int H, int W, int relu, int dtype, dvt_stream_t stream);
int dvt_bn_bwd_pooled(const void* dy_pool, const void* idx, const void* x, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, void* dx, float* dgamma, float* dbeta, void* workspace, int64_t N,
                      int C, int H, int W, int relu, int training, int accumulate, int dtype, dvt_stream_t stream);
/* nn.MaxPool2d(k, stride, pad) on NHWC; idx: uint8 [N*Ho*Wo*C] window position of the (first) maximum. */
int dvt_maxpool_fwd(const void* x, void* y, void* idx, int64_t N, int C, int H, int W, int k, int stride, int pad,
                    int dtype, dvt_stream_t stream);
int dvt_maxpool_bwd(const void* dy, const void* idx, void* dx, int64_t N, int C, int H, int W, int k, int stride,
                    int pad, int dtype, dvt_stream_t stream);
/* [B, R, C] -> [B, C, R]  (NHWC <-> NCHW at the module boundary). */
int dvt_transpose_last2(const void* src, void* dst, int64_t B, int R, int Cc, int dtype, dvt_stream_t stream);

/* ---------------------------------------------------------------- losses
 * nn.BCEWithLogitsLoss() (mean): src/models/frame_transformer.py:89,263,268,273.
 * z: [n] dtype, target: [n] f32, loss: [1] f32. */
int dvt_bce_logits_fwd(const void* z, const float* target, float* loss, int64_t n, int dtype,
                       dvt_stream_t stream);
/* dz = gscale * (sigmoid(z) - target) / n */
int dvt_bce_logits_bwd(const void* z, const float* target, const float* gloss, void* dz,
                       int64_t n, int dtype, dvt_stream_t stream);
/* The classification head and its loss in ONE launch (vit.py:97-100 mlp_head = LayerNorm + Linear on the pooled CLS row,
 * :126-128; behind the temporal Transformer's final LayerNorm of :43, which commutes with the CLS pooling; the caller's
 * nn.BCEWithLogitsLoss() mean):  h1 = LN(x; g1, b1) (skipped when g1 == NULL), h2 = LN(h1; g2, b2), z = h2 W^T + c,
 * loss = mean BCE(z, target) -- eight rows at the metric shape, where the five launches this replaces (and the six of
 * their backward) cost ~5 us each and compute nothing to speak of.  All arithmetic fp32 on the fp32 parameters.
 * The same launch leaves d(loss)/d(everything) for an upstream gradient of 1 in `grads` (f32, dvt_head_bce_grads_elems
 * elements: [dx rows*d][dg1 d][db1 d][dg2 d][db2 d][dW classes*d][dc classes], then scratch); the backward pass is then
 * dvt_scaled_emit_group: scale by the incoming gradient of the loss and store / accumulate into the gradient buffers.
 * Limits: rows <= 32, d / 64 in {1, 2, 3, 4, 6, 8, 12, 16}, classes <= 64, classes * d <= 12288 (dvt_head_bce_supported). */
typedef struct dvt_head_bce_desc {
  const void* x;          /* [rows, d] f32 or 16-bit (x_dtype) */
  const float* g1;        /* first LayerNorm (NULL: none) */
  const float* b1;
  const float* g2;        /* second LayerNorm */
  const float* b2;
  const float* w;         /* [classes, d] */
  const float* c;         /* [classes] or NULL */
  const float* target;    /* [rows, classes] */
  float* logits;          /* [rows, classes] out */
  float* loss;            /* [1] out */
  float* grads;           /* out, see above */
  int32_t rows, d, classes, x_dtype;
  float eps1, eps2;
} dvt_head_bce_desc;
int dvt_head_bce_supported(int rows, int d, int classes);
int64_t dvt_head_bce_grads_elems(int rows, int d, int classes);
int dvt_head_bce_fwd(const dvt_head_bce_desc* desc, dvt_stream_t stream);
/* dst[i] = scale[0] * src[i] (+ dst[i] when accumulate), optionally also a 16-bit copy of the RESULT (dst_lp); dst may be
 * NULL when only the copy is wanted.  Up to 16 entries per launch; scale is a device scalar. */
typedef struct dvt_emit_entry {
  const float* src;
  float* dst;
  void* dst_lp;
  int64_t n;
  int32_t accumulate, lp_dtype;
} dvt_emit_entry;
int dvt_scaled_emit_group(const float* scale, const dvt_emit_entry* entries, int count, dvt_stream_t stream);
/* Hard-label distillation CE(student, argmax(teacher)), mean over rows:
 * src/models/frame_transformer.py:90,250.  student/teacher: [rows, C]. */
int dvt_ce_argmax_fwd(const void* student, const void* teacher, float* loss, int64_t rows,
                      int64_t C, int dtype, dvt_stream_t stream);
int dvt_ce_argmax_bwd(const void* student, const void* teacher, const float* gloss, void* dstudent,
                      int64_t rows, int64_t C, int dtype, dvt_stream_t stream);

/* ---------------------------------------------------------------- optimizer
 * torch.optim.AdamW step over a flat fp32 buffer (frame_transformer.py:127-129). */
int dvt_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                   dvt_stream_t stream);
/* Same update with the step counter in device memory (*step_dev is the number of steps
 * already taken; it is incremented by the call), so that the launch can be captured in
 * a hipGraph and replayed.
 * skip64 (nullable; here and in the three flat-buffer forms below): one byte per 64-element block of the flat
 * buffer; blocks whose byte is non-zero are left untouched -- parameter, moments and all.  torch's optimizers skip
 * parameters whose .grad is None (frame_transformer.py:123-134 hands every parameter to the optimizer, and the frozen
 * image encoder :59 / unused members never get a gradient): without the mask a flat-buffer step would still weight-
 * decay them.
 * The bias correction uses the ONE step counter of the flat buffer, where torch keeps a counter per parameter: the two
 * agree when the set of masked (gradient-less) parameters is the same in every step of a run -- a frozen encoder, an
 * unused member --, which is what the path's models do; a parameter that receives a gradient only in some steps would
 * see a larger step number here than under torch.optim.AdamW (same for dvt_adagrad_step's decayed learning rate). */
int dvt_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay,
                       int64_t* step_dev, const uint8_t* skip64, dvt_stream_t stream);
/* The whole flat-buffer optimizer step of the training loop in one launch: the dvt_adamw_step_dev update, the 16-bit
 * mirror of the updated weights that the next step's GEMMs read (mirror nullable; mirror_dtype DVT_BF16 / DVT_F16) and
 * the step counter.  step_dev2: int64[2] = {steps already taken, 0}: element 1 is a ticket the launch uses to let its
 * last workgroup store the increment (it is 0 again when the launch has finished). */
int dvt_adamw_step_fused(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int64_t* step_dev2,
                         const uint8_t* skip64, void* mirror, int mirror_dtype, dvt_stream_t stream);
/* AdamW under dynamic loss scaling (BASELINE configs[4]: fp16 + loss scaling; torch.cuda.amp.GradScaler rule).
 * grad holds the gradient of (scale * loss).  On the device, in stream order: found_inf |= any non-finite grad;
 * unless found_inf: the dvt_adamw_step_dev update with grad / scale and step_dev += 1; then
 * found_inf ? scale *= backoff : (every growth_interval clean steps: scale *= growth); found_inf = 0;
 * loss_grad[0] = scale * loss_grad_base (the seed of the next backward, e.g. base = 1 / world).  No host sync. */
int dvt_adamw_step_scaled(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int64_t* step_dev, float* scale,
                          int32_t* found_inf, int32_t* good_steps, int growth_interval, float growth, float backoff,
                          float* loss_grad, float loss_grad_base, const uint8_t* skip64, dvt_stream_t stream);
/* torch.optim.SGD(lr, momentum, weight_decay) (frame_transformer.py:124-126; config.yaml momentum 0.005):
 * d = g + wd*p; buf = momentum*buf + d; p -= lr*buf.  momentum_buf starts zeroed (may be NULL when momentum == 0). */
int dvt_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                 float weight_decay, const uint8_t* skip64, dvt_stream_t stream);
/* torch.optim.Adagrad(lr, weight_decay) (frame_transformer.py:130-132): d = g + wd*p; sum += d*d;
 * p -= lr/(1+(step-1)*lr_decay) * d / (sqrt(sum) + eps); step counts from 1. */
int dvt_adagrad_step(float* param, const float* grad, float* state_sum, int64_t n, float lr, float lr_decay,
                     float eps, float weight_decay, int64_t step, const uint8_t* skip64, dvt_stream_t stream);

/* ---------------------------------------------------------------- data-parallel gradient exchange
 * The reference is single-GPU (pl.Trainer(gpus=1), src/main.py:87); north_star partitions the clips of the global
 * batch over the 8 GPUs of a node, and the only exchange of the path is the SUM of the parameter gradients.  RCCL over
 * xGMI behind the ABI, one process per GPU:
 *   rank 0 calls dvt_comm_unique_id and hands the 128 bytes to every rank (any side channel: torch.distributed's
 *   store, MPI, a file); every rank calls dvt_comm_init (a collective: ncclCommInitRank on the calling thread's current
 *   device); dvt_comm_allreduce / dvt_comm_broadcast ENQUEUE an in-place collective on `stream` (no host sync: they may
 *   be captured in a hipGraph next to the backward kernels they overlap with); dvt_comm_destroy frees the communicator.
 * dtype: DVT_F32, or DVT_BF16 / DVT_F16 for half-width gradient buckets.  RCCL is bound at run time (the instance the
 * process has already loaded, e.g. PyTorch's, else the ROCm installation's); DVT_ERR_UNSUPPORTED when none is found. */
#define DVT_COMM_ID_BYTES 128
typedef void* dvt_comm_t;
int dvt_comm_unique_id(void* id_out /* DVT_COMM_ID_BYTES */);
int dvt_comm_init(dvt_comm_t* comm_out, const void* unique_id, int world, int rank);
int dvt_comm_allreduce(dvt_comm_t comm, void* buf, int64_t count, int dtype, dvt_stream_t stream);
int dvt_comm_broadcast(dvt_comm_t comm, void* buf, int64_t count, int dtype, int root, dvt_stream_t stream);
int dvt_comm_destroy(dvt_comm_t comm);

#ifdef __cplusplus
}
#endif
#endif /* DVT_HIP_H */

/*
 * mos_hip.h — C-ABI of libmos_hip.so: the MI355X (gfx950) kernels behind the
 * Mix-of-Show hot path (ED-LoRA attention, regional attention, gradient fusion).
 *
 * Conventions (all entry points):
 *   - return 0 on success, a negative mos_status on error; never throw, never exit.
 *     mos_last_error_string() returns a thread-local description of the last error.
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`.
 *   - the caller allocates every output and workspace and owns all memory.
 *   - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*;
 *     NULL = the legacy default stream). No entry point synchronises.
 *   - `dtype` selects the storage/MFMA input type of activations and packed weights:
 *     MOS_F16 or MOS_BF16. Accumulation is always fp32; statistics (lse, pcols, grads of
 *     LoRA factors, Gram matrices) are fp32 or fp64 as documented.
 *   - row strides (`ld*`) are in ELEMENTS and must be multiples of 8 (16-byte rows);
 *     base pointers must be 16-byte aligned.
 *
 * Each entry point cites the reference interface it replaces
 * (paths relative to TencentARC/Mix-of-Show).
 */
#ifndef MOS_HIP_H
#define MOS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { MOS_F16 = 0, MOS_BF16 = 1 } mos_dtype;

typedef enum {
    MOS_OK = 0,
    MOS_ERR_BAD_ARG = -1,      /* NULL pointer, bad size, misaligned stride */
    MOS_ERR_UNSUPPORTED = -2,  /* head dim / dtype / rank not compiled */
    MOS_ERR_LAUNCH = -3        /* hipGetLastError() after launch != hipSuccess */
} mos_status;

#define MOS_LORA_PAD 16   /* packed LoRA rank dimension (sum of ranks of fused sites <= 16) */
#define MOS_MAX_PCOLS 4   /* max probability columns exported by the training cross-attn */
#define MOS_MAX_SOURCES 9 /* context + up to 8 regions in one regional-attention launch */

int mos_version(void);
const char* mos_last_error_string(void);

/* Optional per-kernel timing (HIP events recorded on the launch stream around every kernel of this library).
 * mos_profile_begin() clears and enables; mos_profile_end() disables, waits for the recorded events and
 * returns the number of distinct (kernel, shape) records; mos_profile_get() reads record `idx`:
 * total milliseconds, number of launches and the summed ALGORITHMIC flops / bytes of those launches. */
int mos_profile_begin(void);
int mos_profile_end(void);
int mos_profile_get(int idx, char* name, int name_cap, double* total_ms, long long* calls,
                    double* flops, double* bytes);

/* ------------------------------------------------------------------------------------------
 * LoRA-augmented linear: replaces LoRALinearLayer.forward (mixofshow/models/edlora.py:244-246)
 *     y = orig(x) + alpha * lora_up(lora_down(x))
 * for one site or for several sites that share x (fused q/k/v: edlora.py:69-71,143-145).
 *
 * Packed operands (built by mos_lora_pack from the fp32 master parameters):
 *   A16  [16, K]   rows g*r..g*r+r-1 = lora_down.weight of site g, other rows 0
 *   A16T [K, 16]   its transpose
 *   Bp16 [N, 16]   row n of site g: cols g*r.. = alpha_g * lora_up.weight[n - n0_g, :], else 0
 *   BpT  [16, N]   its transpose
 * ------------------------------------------------------------------------------------------ */

/* Up to 4 fused sites; site g covers output rows [n_begin[g], n_begin[g]+n_rows[g]). */
typedef struct {
    int n_sites;
    int rank;                 /* r, same for all sites; n_sites*rank <= 16 */
    int K;                    /* in_features */
    int N;                    /* total out_features (sum over sites) */
    const float* down[4];     /* [r, K]    fp32 master lora_down.weight */
    const float* up[4];       /* [n_rows, r] fp32 master lora_up.weight */
    float alpha[4];
    int n_begin[4];
    int n_rows[4];
} mos_lora_sites;

int mos_lora_pack(const mos_lora_sites* sites_host, int dtype,
                  void* A16, void* A16T, void* Bp16, void* BpT, void* stream);

/* Every group of a model in ONE launch (one descriptor per fused-projection group, array in DEVICE memory): what a
 * training step needs once per optimiser update instead of one mos_lora_pack launch per projection call. */
typedef struct {
    mos_lora_sites s;
    void* A16; void* A16T; void* Bp16; void* BpT;   /* outputs of this group, `dtype` */
} mos_lora_group;
int mos_lora_pack_all(const mos_lora_group* groups_dev, int n_groups, int max_elems, int dtype, void* stream);

/* t[M,16] = x[M,K] . A16^T   (lora_down of all fused sites, one pass over x) */
int mos_lora_down(const void* x, int64_t ldx, const void* A16, void* t,
                  int M, int K, int dtype, void* stream);

/* y[M,N] = x[M,K] . W[N,K]^T (+ t[M,16] . Bp16[N,16]^T) (+ bias[N])
 * W is the frozen base weight in `dtype`; t/Bp16 may be NULL (plain linear).
 * bias is fp32 or NULL. */
int mos_lora_linear_fwd(const void* x, int64_t ldx, const void* W, int64_t ldw,
                        const void* t, const void* Bp16, const float* bias,
                        void* y, int64_t ldy, int M, int N, int K, int dtype, void* stream);

/* Same result with the down projection FUSED into the GEMM: t = x . A16^T is accumulated in the K loop next to the
 * base product (A16 rides as 16 extra weight rows), rounded to `dtype` and fed to the rank-16 epilogue from registers;
 * t_out [M,16] (may be NULL) receives it for the backward. One launch, no separate mos_lora_down pass over x. */
int mos_lora_linear_fused_fwd(const void* x, int64_t ldx, const void* W, int64_t ldw,
                              const void* A16, const void* Bp16, const float* bias,
                              void* y, int64_t ldy, void* t_out, int M, int N, int K, int dtype, void* stream);

/* Epilogue variants of the forward GEMM (round 4). The callers are the transformer block's feed-forward and the 1x1
 * `proj_out` of diffusers' Transformer2DModel, which the reference's processors sit in (edlora.py:49-75,124-141 return into
 * `hidden_states = attn(...) + hidden_states; hidden_states = ff(norm3(hidden_states)) + hidden_states`):
 *   residual != NULL : y = round(x.W^T (+ LoRA) + bias) + residual[M, Nout]   -- the rounding points of "GEMM, then an add
 *                      kernel" (bit-identical to that pair), without the add launch and its 3 passes over the activation
 * A16 / Bp16 NULL: plain GEMM; otherwise the fused LoRA form of mos_lora_linear_fused_fwd (t_out as there). */
typedef struct {
    const void* residual;     /* [M, Nout] in `dtype`, or NULL */
    int64_t ldr;              /* its row stride in elements */
} mos_gemm_epilogue;
int mos_lora_linear_fwd_ex(const void* x, int64_t ldx, const void* W, int64_t ldw,
                           const void* A16, const void* Bp16, const float* bias,
                           void* y, int64_t ldy, void* t_out, int M, int N, int K, int dtype,
                           const mos_gemm_epilogue* epilogue_host, void* stream);

/* Backward of the above w.r.t. x and the packed LoRA factors (W is frozen: no dW).
 *   dt[M,16]  = dy . BpT^T                      (written to dt)
 *   dx[M,K]   = dy[M,N] . Wt[K,N]^T + dt . A16T[K,16]^T   (Wt = W^T, cached by the caller)
 *   dA16[16,K] (fp32) = dt^T . x
 *   dBpT[16,N] (fp32) = t^T . dy               (caller scales by alpha and slices per site)
 * lora_cols: number of packed rank columns in use (n_sites * rank <= 16); the factor-gradient reduction only
 *            computes that many rows (rounded up to 4), the remaining rows of dA16/dBpT are zero.
 * ws: fp32 workspace of mos_lora_bwd_workspace_bytes(M,N,K) bytes.
 * dx may be NULL (skip), dA16/dBpT may be NULL (no LoRA / frozen LoRA). */
int64_t mos_lora_bwd_workspace_bytes(int M, int N, int K);
int mos_lora_linear_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx,
                        const void* Wt, int64_t ldwt, const void* t,
                        const void* A16T, const void* BpT,
                        void* dt, void* dx, int64_t lddx, float* dA16, float* dBpT,
                        void* ws, int M, int N, int K, int lora_cols, int dtype, void* stream);

/* Fused backward: launch 1 = dx (with dt = dy . BpT^T produced in the same kernel, written to dt); launches 2+3 = both
 * factor gradients, reduced over tokens (one launch for both) and summed in chunk order (deterministic) STRAIGHT into
 * the fp32 parameter gradients:
 *   down_grad[g][r, K]      (+)= (dt^T x)[g*r .. g*r+r-1, :]
 *   up_grad[g][n_rows, r]   (+)= alpha_g * (t^T dy)[g*r .. , n_begin .. n_begin+n_rows)^T
 * (NULL pointers are skipped; accumulate_* = 1 adds to what is there — gradient buckets, micro-batches).
 * ws: mos_lora_bwd_workspace_bytes(M,N,K) bytes. dx may be NULL; grads_host may be NULL (only dt / dx wanted). */
typedef struct {
    int n_sites, rank;
    float* down_grad[4];
    float* up_grad[4];
    float alpha[4];
    int n_begin[4];
    int n_rows[4];
    int accumulate_down[4];
    int accumulate_up[4];
} mos_lora_grad_out;
int mos_lora_linear_fused_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx,
                              const void* Wt, int64_t ldwt, const void* t,
                              const void* A16T, const void* BpT,
                              void* dt, void* dx, int64_t lddx,
                              const mos_lora_grad_out* grads_host, void* ws,
                              int M, int N, int K, int lora_cols, int dtype, void* stream);

/* Batched form of the last step of mos_lora_linear_fused_bwd. `_deferred` runs dx + dt and the token reduction of both
 * factor gradients but NOT the ordered final sum: it fills *rec_host with what that sum needs (pointers into `ws`, which the
 * caller keeps alive, and the gradient targets). After the backward pass the caller uploads the records of all groups
 * (block_begin = running sum of n_blocks) and ONE mos_lora_grad_final_all launch (total_blocks = sum of n_blocks) writes /
 * accumulates every LoRA factor gradient of the step (edlora.py:244-246 has ~100 such layers in SD-1.5 + CLIP): same
 * summation order as the per-group form, bit-identical results. */
typedef struct {
    const float* partial[2];
    int C[2], cb[2];
    int nchunk, nj, block_begin, n_blocks;
    mos_lora_grad_out out;
} mos_lora_final_rec;
int mos_lora_linear_fused_bwd_deferred(const void* dy, int64_t lddy, const void* x, int64_t ldx,
                                       const void* Wt, int64_t ldwt, const void* t,
                                       const void* A16T, const void* BpT,
                                       void* dt, void* dx, int64_t lddx,
                                       const mos_lora_grad_out* grads_host, void* ws,
                                       int M, int N, int K, int lora_cols, int dtype, void* stream,
                                       mos_lora_final_rec* rec_host);
int mos_lora_grad_final_all(const mos_lora_final_rec* recs_dev, int n_recs, int total_blocks, void* stream);

/* Round 5: the token reductions themselves batched. `_deferred_all` launches only the dx (+dt) GEMM and describes the group's
 * token reduction in `job_host` (pointers to dt / x / t / dy, which the caller keeps alive) and its final sum in `rec_host`;
 * mos_lora_grad_all then runs the reductions of EVERY group of one padded-rank class `nj` (4 / 8 / 12 / 16) of a backward pass in
 * ONE launch from a table in device memory (`block_begin` = running sum of `n_blocks` within the table, filled by the caller),
 * followed by mos_lora_grad_final_all. Per group the arithmetic
 * and its order are those of the per-group launch: bit-identical gradients. `flops` / `bytes`: sums of the jobs' figures (profiler). */
typedef struct {
    const void* P[2];         /* dt, t   [M,16] */
    const void* Z[2];         /* x [M,K], dy [M,N] */
    int64_t ldz[2];
    int C[2], cb[2];          /* K, N ; 64-column blocks of each */
    float* partial[2];        /* [nchunk][nj][C] fp32 partial sums inside the group's workspace */
    int M, rpc, nchunk, nj;
    int block_begin, n_blocks;
    double flops, bytes;
} mos_lora_grad_job;
int mos_lora_linear_fused_bwd_deferred_all(const void* dy, int64_t lddy, const void* x, int64_t ldx,
                                           const void* Wt, int64_t ldwt, const void* t,
                                           const void* A16T, const void* BpT,
                                           void* dt, void* dx, int64_t lddx,
                                           const mos_lora_grad_out* grads_host, void* ws,
                                           int M, int N, int K, int lora_cols, int dtype, void* stream,
                                           mos_lora_final_rec* rec_host, mos_lora_grad_job* job_host);
int mos_lora_grad_all(const mos_lora_grad_job* jobs_dev, int n_jobs, int total_blocks, int nj, int dtype, double flops, double bytes,
                      void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused attention core: softmax(scale * Q K^T) V with online softmax, no materialised P.
 * Replaces attn.get_attention_scores + torch.bmm / xformers.memory_efficient_attention in
 * EDLoRA_AttnProcessor / EDLoRA_Control_AttnProcessor (edlora.py:77-83,151-156), the default
 * self-attention of diffusers Attention, and RegionT2I_AttnProcessor base attention
 * (pipeline_regionally_t2iadapter.py:111-116).
 *
 * Layout: q is addressed as q[b*q_bs + n*q_rs + h*d + c]  (token-major "(B, N, H*d)" — the
 * head_to_batch_dim permute of the reference is folded into the addressing, so q/k/v may be
 * column slices of one fused projection output). Same for k, v, o.
 *   lse   [B, H, Nq] fp32: log-sum-exp of scaled scores (saved for backward), may be NULL.
 *   pcols [B, H, Nq, n_pcols] fp32 (optional, cross-attention training): softmax probability
 *         of key index tok_idx[b*n_pcols + t] — all that cal_attn_reg (trainer_edlora.py:263-313)
 *         consumes of the maps AttentionStore keeps (ptp_util.py:79-98).
 * Supported head dims: 40, 80, 160 (SD-1.5 UNet, 8 heads at C = 320/640/1280) and 64 (CLIP ViT-L/14 text tower, 12 heads,
 * causal, 77 tokens: CLIPAttention of the text encoder the ED-LoRA tune trains, trainer_edlora.py:97-115,224-232).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int B, H, Nq, Nkv, d;
    int64_t q_bs, q_rs;   /* batch / row strides in elements */
    int64_t k_bs, k_rs;
    int64_t v_bs, v_rs;
    int64_t o_bs, o_rs;
    float scale;
    int causal;           /* 1: keys with index > the query's index are masked (CLIP text tower; needs Nq == Nkv) */
} mos_attn_shape;

int mos_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                 const int32_t* tok_idx, int n_pcols, float* pcols,
                 const mos_attn_shape* shape_host, int dtype, void* stream);

/* Backward. dO has o's layout; dq/dk/dv have q/k/v's layouts (strides given separately so
 * they may be slices of one fused buffer). dpcols may be NULL.
 * ws: mos_attn_bwd_workspace_bytes() bytes (fp32 D vector, split-q partials for dK/dV). */
typedef struct {
    int64_t do_bs, do_rs;
    int64_t dq_bs, dq_rs;
    int64_t dk_bs, dk_rs;
    int64_t dv_bs, dv_rs;
} mos_attn_grad_strides;

int64_t mos_attn_bwd_workspace_bytes(const mos_attn_shape* shape_host);
int mos_attn_bwd(const void* q, const void* k, const void* v, const void* o, const float* lse,
                 const void* dO, const int32_t* tok_idx, int n_pcols, const float* pcols,
                 const float* dpcols, void* dq, void* dk, void* dv, void* ws,
                 const mos_attn_shape* shape_host, const mos_attn_grad_strides* gs_host,
                 int dtype, void* stream);

/* Names the reference-side binding uses (thin wrappers over mos_attn_fwd/bwd; SURVEY 8(b)'s proposed set). */
int mos_self_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                      const mos_attn_shape* shape_host, int dtype, void* stream);
int mos_cross_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                       const int32_t* tok_idx, int n_pcols, float* pcols,
                       const mos_attn_shape* shape_host, int dtype, void* stream);
int mos_self_attn_bwd(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dO,
                      void* dq, void* dk, void* dv, void* ws, const mos_attn_shape* shape_host,
                      const mos_attn_grad_strides* gs_host, int dtype, void* stream);
int mos_cross_attn_bwd(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dO,
                       const int32_t* tok_idx, int n_pcols, const float* pcols, const float* dpcols,
                       void* dq, void* dk, void* dv, void* ws, const mos_attn_shape* shape_host,
                       const mos_attn_grad_strides* gs_host, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Regional cross-attention mask-and-blend: replaces RegionT2I_AttnProcessor.region_rewrite
 * (pipeline_regionally_t2iadapter.py:32-86) fused with the base cross-attention (:115-116).
 *   out[q] = base(q)                                   if no region box covers q
 *          = sum_{r covers q} attn(q, K_r, V_r) / count(q)   otherwise (replace_ratio = 1.0)
 * Source 0 is the context prompt (base attention); sources 1..n_regions are regions with
 * integer feature-map boxes [h0,h1) x [w0,w1) (already ceil/floor-rounded per :38-39 by host).
 *   k_src / v_src: [n_src][B, Nkv, H*d] with strides src_stride (between sources), k_bs, k_rs;
 *   65 <= Nkv <= 96 (CLIP's 77-token context: three 32-key sub-tiles of one LDS residency).
 * Inference only (no backward in the reference: pipeline __call__ is @torch.no_grad, :301).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int n_regions;                 /* <= MOS_MAX_SOURCES-1 */
    int feat_h, feat_w;            /* Nq == feat_h*feat_w */
    int box[MOS_MAX_SOURCES - 1][4]; /* h0, w0, h1, w1 in feature cells */
    int64_t src_stride;            /* elements between consecutive sources in k_src/v_src */
} mos_region_desc;

int mos_region_cross_attn_fwd(const void* q, const void* k_src, const void* v_src, void* o,
                              const mos_attn_shape* shape_host, const mos_region_desc* reg_host,
                              int dtype, void* stream);
/* Region lists longer than MOS_MAX_SOURCES-1 (the reference loops over an unbounded `region_list`,
 * mixofshow/pipelines/pipeline_regionally_t2iadapter.py:60-83): walk the list in chunks of <= MOS_MAX_SOURCES-1 boxes.
 *   total_count : device pointer, Nq bytes: number of boxes of the WHOLE list covering each query (the reference's `count`
 *                 tensor, :56,80); every chunk divides by it. NULL = count this launch's boxes (one-launch case).
 *   accumulate  : 0 = first chunk: o = blend of this chunk's regions, queries with total_count 0 get the context prompt
 *                 (source 0); 1 = later chunk: o += blend of this chunk's regions, source 0 is not read (pass the pointer of
 *                 the source just before the chunk's first region so that sources 1.. are the chunk's regions).
 * One chunk == mos_region_cross_attn_fwd. */
int mos_region_cross_attn_fwd_chunk(const void* q, const void* k_src, const void* v_src, void* o,
                                    const mos_attn_shape* shape_host, const mos_region_desc* reg_host,
                                    const void* total_count, int accumulate, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Cross-attention with the probabilities MATERIALISED: the controller half of the processor boundary. The reference hands
 * the full (B*H, N, 77) tensor to any controller between softmax and P.V (mixofshow/models/edlora.py:81-83;
 * AttentionControl.__call__ / AttentionStore.forward, mixofshow/utils/ptp_util.py:37-53,79-82, store it or edit its conditional
 * half in place). Controllers that declare the key columns they read take mos_cross_attn_fwd's probability columns instead;
 * every other controller gets exactly the reference's split:
 *   mos_attn_probs: probs[(b*H + h), q, j] = softmax_j(scale * q_h[q] . k_h[j])   dense (B*H, Nq, Nkv) in `dtype`
 *   mos_attn_pv   : o[b, q, h*d + c]       = sum_j probs[(b*H + h), q, j] * v_h[j, c]
 * shape as for mos_attn_fwd (q/k strides for _probs, v/o strides for _pv); Nkv <= 96, d in {40, 80, 160}, no causal mask.
 * Backward (round 6; the reference gives the map WITH grad to any controller, e.g. its own AttentionStore(training=True) whose
 * stored maps cal_attn_reg differentiates, ptp_util.py:37-53,79-82 / trainer_edlora.py:263-313):
 *   mos_attn_pv_bwd   : dprobs[(b*H+h), q, j] = sum_c dO[b, q, h*d+c] v_h[j, c]   dense, `dtype`
 *                       dv[b, j, h*d+c]       = sum_q probs[(b*H+h), q, j] dO[b, q, h*d+c]
 *   mos_attn_probs_bwd: dS = probs o (dprobs - rowsum(probs o dprobs)); dq = scale * dS . k_h ; dk = scale * dS^T . q_h
 *   (probs = what mos_attn_probs returned for _probs_bwd, what the controller returned for _pv_bwd; dprobs of _probs_bwd = the
 *   SUM of the gradient arriving through mos_attn_pv and the one the controller's loss sends into the map: autograd adds them.)
 *   Strides: v in shape.v_*, q / k in shape.q_* / k_*; dO, dq, dk, dv in mos_attn_grad_strides. ws: mos_attn_probs_bwd_workspace_bytes
 *   (per-64-query-block fp32 partials of the key-side sum, combined in block order: deterministic). One workspace size for both.
 * ------------------------------------------------------------------------------------------ */
int mos_attn_probs(const void* q, const void* k, void* probs, const mos_attn_shape* shape_host, int dtype, void* stream);
int mos_attn_pv(const void* probs, const void* v, void* o, const mos_attn_shape* shape_host, int dtype, void* stream);
int64_t mos_attn_probs_bwd_workspace_bytes(const mos_attn_shape* shape_host);
int mos_attn_pv_bwd(const void* probs, const void* v, const void* dO, void* dprobs, void* dv, void* ws,
                    const mos_attn_shape* shape_host, const mos_attn_grad_strides* gs_host, int dtype, void* stream);
int mos_attn_probs_bwd(const void* q, const void* k, const void* probs, const void* dprobs, void* dq, void* dk, void* ws,
                       const mos_attn_shape* shape_host, const mos_attn_grad_strides* gs_host, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Gradient-fusion least squares: replaces chunk_compute_mse + the closure of
 * update_quasi_newton (gradient_fusion.py:22-35,62-76). The loss
 *     L(W) = mean((X W^T - Y)^2) = (tr(W G W^T) - 2 tr(W P^T) + c) / (n * Cout)
 * depends on the data only through G = X^T X, P = Y^T X, c = sum(Y^2), which are accumulated
 * ONCE per layer (streamed straight from the forward hooks, gradient_fusion.py:150-167)
 * instead of re-uploading X and Y on every closure evaluation.
 *   mos_gram_accumulate: G[Cin,Cin] += X^T X ; P[Cout,Cin] += Y^T X ; c += sum(Y^2)
 *       X [n,Cin], Y [n,Cout] in `dtype` (the hooks record fp16 activations);
 *       G, P, c are fp64 device accumulators (fp32 MFMA partials per row-chunk, fp64 combine).
 *       ws: mos_gram_workspace_bytes(n,Cin,Cout) bytes.
 *   mos_lsq_loss_grad_gram: loss (fp64 scalar on device) and grad[Cout,Cin] (fp64) of L at W (fp64);
 *       ws: mos_lsq_workspace_bytes(Cout,Cin) bytes (per-block partial sums, combined in fixed order).
 * ------------------------------------------------------------------------------------------ */
int64_t mos_gram_workspace_bytes(int64_t n, int Cin, int Cout);
int mos_gram_accumulate(const void* X, int64_t ldx, const void* Y, int64_t ldy, int64_t n,
                        int Cin, int Cout, int dtype, double* G, double* P, double* c,
                        void* ws, void* stream);
int64_t mos_lsq_workspace_bytes(int Cout, int Cin);

int mos_lsq_loss_grad_gram(const double* W, const double* G, const double* P, const double* c,
                           double n_times_cout, int Cout, int Cin, double* loss, double* grad,
                           void* ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused GroupNorm (+ SiLU) on half-precision NCHW activations — a CALLER of the hot path (SURVEY.md
 * §8(f) item 1: the ResnetBlock2D / Transformer2DModel / VAE norm->activation pairs that diffusers runs
 * as fp32 group_norm + silu under autocast). x, y, dy, dx: (B, C, HW) contiguous in `dtype`;
 * gamma/beta fp32 (frozen: no affine gradients); stats (B*G, 2) fp32 = mean, rstd (saved for backward);
 * HW must be a multiple of 8; ws: mos_groupnorm_workspace_bytes() bytes.
 * ------------------------------------------------------------------------------------------ */
int64_t mos_groupnorm_workspace_bytes(int B, int C, int HW, int G);
int mos_groupnorm_silu_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                           void* ws, int B, int C, int HW, int G, float eps, int silu, int dtype, void* stream);
int mos_groupnorm_silu_bwd(const void* dy, const void* x, const float* gamma, const float* beta,
                           const float* stats, void* dx, void* ws, int B, int C, int HW, int G, int silu,
                           int dtype, void* stream);

/* Same operator on channels-last activations: x, y, dy, dx are (B, HW, C) contiguous — the layout of the token-major
 * attention path and of MIOpen's fp16 NHWC implicit-GEMM convolutions, so a UNet kept in channels_last needs no
 * NCHW<->NHWC transposes around convolutions and no permute copies around the transformer blocks. C % 8 == 0,
 * C <= 4096, G <= 64. ws: mos_groupnorm_nhwc_workspace_bytes() bytes (per-slice partial sums + the per-group constants
 * one block per image folds them into). `silu` is a flag word: bit 0 = SiLU after the norm; bit 1 (MOS_GN_FORCE_SLICES) =
 * take the three-launch slice form even where the one-launch column kernel applies (parity tests and A/B runs: there is no
 * environment switch and no other process-global state behind these entry points). */
#define MOS_GN_FORCE_SLICES 2
int64_t mos_groupnorm_nhwc_workspace_bytes(int B, int C, int HW, int G);
int mos_groupnorm_silu_fwd_nhwc(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                                void* ws, int B, int C, int HW, int G, float eps, int silu, int dtype, void* stream);
int mos_groupnorm_nhwc_reads_twice(int B, int C, int HW, int G);
/* (`silu` of the _pre form: bit 0 = SiLU; MOS_GN_PRE_TWO_LAUNCHES / MOS_GN_PRE_ONE_LAUNCH pin the form the library otherwise picks
 * from the cost of re-adding the producer's tile sums in every workgroup -- one launch on the UNet's maps, a small finalize launch
 * + the streaming launch on the VAE's, whose 512 x 512 images carry 1024 tiles: parity tests and tools.) */
#define MOS_GN_PRE_TWO_LAUNCHES 4
#define MOS_GN_PRE_ONE_LAUNCH 8
int mos_groupnorm_silu_fwd_nhwc_pre(const void* x, const float* chan_part, int tiles_per_image, const float* gamma,
                                    const float* beta, void* y, float* stats, void* ws, int B, int C, int HW, int G, float eps,
                                    int silu, int dtype, void* stream);
int mos_groupnorm_silu_bwd_nhwc(const void* dy, const void* x, const float* gamma, const float* beta,
                                const float* stats, void* dx, void* ws, int B, int C, int HW, int G, int silu,
                                int dtype, void* stream);
/* The same with the gradient `ds` (x's shape / layout / dtype) of a residual connection that bypasses the norm:
 * dx = round(GN_bwd(dy)) + ds -- what autograd's accumulation of the two gradients yields, without the extra elementwise
 * launch (the inputs of ResnetBlock2D and Transformer2DModel feed both a GroupNorm and a skip path). */
int mos_groupnorm_silu_bwd_nhwc_res(const void* dy, const void* ds, const void* x, const float* gamma, const float* beta,
                                    const float* stats, void* dx, void* ws, int B, int C, int HW, int G, int silu, int dtype,
                                    void* stream);
/* ... with `ds` read in place from a channel slice of a wider channels-last tensor (the gradient of one input of a torch.cat along
 * the channels: the UNet's skip concatenations): ds_pixel_stride = elements between consecutive pixels of ds (>= C, % 8 == 0). */
int mos_groupnorm_silu_bwd_nhwc_res_ps(const void* dy, const void* ds, int64_t ds_pixel_stride, const void* x, const float* gamma,
                                       const float* beta, const float* stats, void* dx, void* ws, int B, int C, int HW, int G,
                                       int silu, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * 3x3 / stride 1 / pad 1 convolution on channels-last activations as an implicit GEMM (the ResnetBlock2D, Upsample2D
 * and VAE convolutions that surround the attention layers; frozen weights in ED-LoRA training, so forward and
 * backward-DATA only — the latter is this same entry point called on dy with the flipped, transposed weight).
 *   x        (B, H, W, Cin)   in `dtype` (with upsample2x: (B, H/2, W/2, Cin), read through a nearest 2x upsample)
 *   w        (Cout, 3, 3, Cin) in `dtype`  (= a PyTorch conv weight in channels_last memory format)
 *   bias     fp32 [Cout] or NULL; tbias (B, Cout) in `dtype` or NULL (per-sample bias: the ResNet time-embedding add);
 *   residual (B, H, W, Cout) in `dtype` or NULL (added after rounding the convolution, like `x + conv(h)`)
 *   y        (B, H, W, Cout)
 * Cin % 64 == 0, Cout % 8 == 0. No workspace.
 * `_ws` form (round 4): with a caller-allocated fp32 workspace of mos_conv3x3_nhwc_workspace_bytes() bytes (0 = this shape
 * does not use one) the low-resolution levels (16x16 / 8x8 maps: a quarter-full chip walking a 9*Cin-deep K loop while the
 * operator is bound by streaming 30-59 MB of weights) run split over K: per-range fp32 partial tiles + one ordered
 * (deterministic) reduction that applies the same epilogue. ws NULL: the unsplit kernel, as mos_conv3x3_nhwc.
 * ------------------------------------------------------------------------------------------ */
int mos_conv3x3_nhwc(const void* x, const void* w, const float* bias, const void* tbias, const void* residual,
                     void* y, int B, int H, int W, int Cin, int Cout, int upsample2x, int dtype, void* stream);
int64_t mos_conv3x3_nhwc_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int mos_conv3x3_nhwc_ws(const void* x, const void* w, const float* bias, const void* tbias, const void* residual,
                        void* y, int B, int H, int W, int Cin, int Cout, int upsample2x, int dtype, void* ws, void* stream);
/* 3x3 / STRIDE-2 convolution of the down-samplers (round 6; diffusers Downsample2D: UNet `padding=1`, VAE encoder
 * `F.pad(x, (0, 1, 0, 1))` + `padding=0`), forward only, same operand layout as mos_conv3x3_nhwc:
 *   x [B, Hin, Win, Cin] NHWC, w [Cout, 3, 3, Cin], bias fp32 or NULL, y [B, Hout, Wout, Cout]
 *   pad_mode 1: Hout = (Hin - 1) / 2 + 1 (padding 1);  pad_mode 2: Hout = (Hin - 2) / 2 + 1 (zero row / column appended at the
 *   bottom / right -- folded into the kernel's bounds, the padded copy is never materialised).
 *   ws: mos_conv3x3_nhwc_workspace_bytes(B, Hout, Wout, Cin, Cout) bytes or NULL (split-K form of the low-resolution levels). */
/* The same convolution, also leaving the GroupNorm statistics of its OUTPUT (round 6): gn_part [B][tiles][Cout][2] fp32 = per
 * (pixel tile, channel) sum and sum of squares of the stored values; tiles = mos_conv3x3_gn_tiles(B, H, W, Cin, Cout) per image
 * (H, W = output size), 0 = this shape's kernel form keeps no statistics (pass gn_part NULL). Consumer:
 * mos_groupnorm_silu_fwd_nhwc_pre. Replaces nothing in the reference -- it removes the statistics pass of the GroupNorm that
 * follows every ResnetBlock2D convolution (diffusers resnet.py; callers mixofshow/pipelines/trainer_edlora.py:237). */
int mos_conv3x3_gn_tiles(int B, int H, int W, int Cin, int Cout);
int mos_conv3x3_nhwc_gn(const void* x, const void* w, const float* bias, const void* tbias, const void* residual, void* y,
                        int B, int H, int W, int Cin, int Cout, int upsample2x, int dtype, void* ws, void* gn_part, void* stream);
/* ... with x read in place from a channel slice of a wider channels-last tensor: x_pixel_stride = elements between consecutive
 * pixels of x (>= Cin, % 8 == 0; mos_conv3x3_nhwc_gn is this with x_pixel_stride = Cin). The dX convolution of a layer whose
 * output went into a torch.cat reads its slice of the concatenation's gradient this way, without a contiguous copy. */
int mos_conv3x3_nhwc_px(const void* x, int64_t x_pixel_stride, const void* w, const float* bias, const void* tbias,
                        const void* residual, void* y, int B, int H, int W, int Cin, int Cout, int upsample2x, int dtype, void* ws,
                        void* gn_part, void* stream);
int mos_conv3x3_s2_nhwc(const void* x, const void* w, const float* bias, void* y, int B, int Hin, int Win, int Cin, int Cout,
                        int pad_mode, int dtype, void* ws, void* stream);


/* ------------------------------------------------------------------------------------------
 * Row-wise operators of the transformer blocks around the attention layers:
 *   LayerNorm: y = (x - mean) * rstd * gamma + beta over the last dim; x, y (rows, C) contiguous in `dtype`, gamma/beta
 *     fp32 (frozen: no affine gradients), stats (rows, 2) fp32 = mean, rstd (may be NULL forward-only). C % 8 == 0, C <= 2048.
 *     (BasicTransformerBlock.norm1/2/3 and CLIP's layer norms, which autocast runs as fp32 layer_norm between casts.)
 *   GEGLU: y[rows, F] = h[:, :F] * gelu(h[:, F:]) for h (rows, 2F) contiguous (exact erf GELU; diffusers GEGLU), and
 *     dh from dy.
 * ------------------------------------------------------------------------------------------ */
int mos_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int rows, int C,
                      float eps, int dtype, void* stream);
int mos_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* stats, void* dx, int rows, int C,
                      int dtype, void* stream);
/* Residual add fused into the LayerNorm that consumes it (transformer blocks are `x = x + f(LN(x))`; reference: the `+`
 * of diffusers BasicTransformerBlock / transformers CLIPEncoderLayer followed by nn.LayerNorm, each a separate kernel):
 *   forward   s = x + r (written to s_out),  y = LN(s) * gamma + beta;   r == NULL: y = LN(x), s_out ignored
 *   backward  dx = LN_bwd(dy; s, gamma, stats) + ds;   ds == NULL: no bypass gradient;
 *             dx_half (optional, only with stream_fp32): the same gradient rounded to `dtype` for the half branch r
 * x / s_out / ds / s / dx live in the RESIDUAL STREAM dtype: `dtype` itself (stream_fp32 = 0, the UNet: bit-identical to
 * add + layernorm as separate kernels) or fp32 (stream_fp32 = 1, the CLIP tower under autocast: statistics and gradient
 * from the fp32 sum, like the reference's fp32 layer_norm). r, y, dy, dx_half are `dtype`. Shapes as mos_layernorm_*. */
int mos_add_layernorm_fwd(const void* x, const void* r, const float* gamma, const float* beta, void* s_out, void* y,
                          float* stats, int rows, int C, float eps, int dtype, int stream_fp32, void* stream);
int mos_add_layernorm_bwd(const void* dy, const void* ds, const void* s, const float* gamma, const float* stats, void* dx,
                          void* dx_half, int rows, int C, int dtype, int stream_fp32, void* stream);
int mos_geglu_fwd(const void* h, void* y, int64_t rows, int F, int dtype, void* stream);
/* y[r, :] = softmax(scale * x[r, :]), x / y (rows, N) contiguous in `dtype` (may alias), N % 8 == 0, N <= 32768: the VAE
 * mid-block attention (single head, d = 512, N = 4096) as scores GEMM -> this -> values GEMM on the library's GEMM. */
int mos_softmax_rows(const void* x, void* y, int rows, int N, float scale, int dtype, void* stream);
int mos_geglu_bwd(const void* dy, const void* h, void* dh, int64_t rows, int F, int dtype, void* stream);
/* quick-GELU of the CLIP text tower MLP (transformers `quick_gelu`, the activation of SD-1.5's text encoder that
 * EDLoRATrainer.forward runs under LoRA, trainer_edlora.py:216-223): y = x * sigmoid(1.702 x) over n contiguous elements
 * (n % 8 == 0), and dx = dy * d/dx[x sigmoid(1.702 x)]. */
int mos_quick_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream);
int mos_quick_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOS_HIP_H */

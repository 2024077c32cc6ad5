// mos_probs.hip — cross-attention with the probabilities MATERIALISED, for the controller half of the plug-in boundary.
//
// The reference hands the full (B*H, N, 77) probability tensor to ANY controller object between softmax and P.V
// (mixofshow/models/edlora.py:81-83: get_attention_scores -> self.controller(attention_probs, is_cross, place) -> torch.bmm),
// and the prompt-to-prompt controllers of mixofshow/utils/ptp_util.py:37-53,79-82 store it, or edit the conditional half of the
// CFG batch in place. The fused kernels of mos_attn.hip never build that tensor (controllers that declare the key columns they
// consume get those columns only). For every OTHER controller these two kernels split the layer where the reference splits it:
//     mos_attn_probs : P[b*H + h, q, j] = softmax_j(scale * q_h . k_h[j])      (fp32 scores and softmax, P rounded once to `dtype`)
//     mos_attn_pv    : O[b, q, h*d + c] = sum_j P[b*H + h, q, j] * v_h[j, c]   (P as the controller returned it)
// Bound: HBM. The product IS the probability tensor: P is written once and read once (2 * B*H*N*Nkv*2 B: 20 MB at level 0 for a
// CFG pair) against 4 * B*H*N*Nkv*d flop (0.8 GFLOP there) -- AI = 20 flop/B, far left of the ridge, so the contraction runs on
// the VALU (fp32 FMA from LDS-resident K / V tiles of <= 96 keys) and the stores / loads of P are staged through LDS so that
// they are 16-byte, fully coalesced runs (a 64-query tile of P is one contiguous block of 64 * Nkv elements).
// Layout: q / k / v / o token-major (B, N, H*d) with explicit strides like mos_attn.hip (no head_to_batch_dim copies); P dense
// (B*H, Nq, Nkv), the reference's layout. Nkv <= 96 (text keys), d in {40, 80, 160}. Round 6: both kernels have a backward (below) so that a controller
// without `token_positions` also trains; the product's own training-time controller still takes the probability-column path of
// mos_attn.hip (no dense map at all).
#include <cstdio>
#include <type_traits>
#include "mos_common.h"

namespace {

constexpr int PQ = 64;          // queries per workgroup
constexpr int PKMAX = 96;       // keys (one LDS tile)
constexpr int PKT = PKMAX / 4;  // keys per thread: 4 threads share a query, thread kl owns keys kl, kl + 4, ...

struct ProbArgs {
    const void* q; const void* kv; void* p; void* o;
    int B, H, Nq, Nkv;
    int64_t q_bs, q_rs, kv_bs, kv_rs, o_bs, o_rs;
    float scale;
};

template <typename T>
__device__ __forceinline__ void unpack8(u32x4 raw, float* dst) {
    const typename MT<T>::v8 v = as_v8<T>(raw);
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = (float)v[e];
}

// rows of a token-major (n, H*d) slice -> fp32 LDS tile [rows][LD], zero rows past n
template <typename T, int D, int LD>
__device__ __forceinline__ void stage_rows(float* lds, const T* base, int64_t rs, int n_valid, int n_rows, int tid) {
    constexpr int CH = D / 8;
    for (int c = tid; c < n_rows * CH; c += 256) {
        const int row = c / CH, cc = (c - row * CH) * 8;
        float v[8];
        if (row < n_valid) unpack8<T>(ld16(base + (int64_t)row * rs + cc), v);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        *reinterpret_cast<f32x4*>(lds + row * LD + cc) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(lds + row * LD + cc + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
}

// ---- P = softmax(scale * Q K^T) ---------------------------------------------------------------------------------------------
// grid (ceil(Nq / 64), B * H); thread = (query tid >> 2, key lane tid & 3). K rows padded to D + 4 floats: the four key lanes of
// a wave instruction read four consecutive rows -> distinct banks for d = 40 / 80 / 160. The query row lives in registers.
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_probs_kernel(const ProbArgs a) {
    constexpr int LDK = D + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* Ks = reinterpret_cast<float*>(smem_raw);                 // [PKMAX][LDK]
    T* Ps = reinterpret_cast<T*>(Ks + PKMAX * LDK);                 // [PQ][Nkv] dense: the global image of this tile
    const int tid = threadIdx.x, ql = tid >> 2, kl = tid & 3;
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int q0 = blockIdx.x * PQ, qi = q0 + ql;
    const int Nkv = a.Nkv;
    stage_rows<T, D, LDK>(Ks, reinterpret_cast<const T*>(a.kv) + (int64_t)b * a.kv_bs + (int64_t)h * D, a.kv_rs, Nkv, PKMAX, tid);
    float qv[D];
    {
        const T* qp = reinterpret_cast<const T*>(a.q) + (int64_t)b * a.q_bs + (int64_t)min(qi, a.Nq - 1) * a.q_rs + (int64_t)h * D;
#pragma unroll
        for (int c = 0; c < D; c += 8) unpack8<T>(ld16(qp + c), qv + c);
    }
    __syncthreads();
    float s[PKT];
#pragma unroll
    for (int j = 0; j < PKT; ++j) {
        const float* kr = Ks + (kl + 4 * j) * LDK;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            const f32x4 kk = *reinterpret_cast<const f32x4*>(kr + c);
            acc = fmaf(qv[c], kk[0], acc); acc = fmaf(qv[c + 1], kk[1], acc);
            acc = fmaf(qv[c + 2], kk[2], acc); acc = fmaf(qv[c + 3], kk[3], acc);
        }
        s[j] = (kl + 4 * j < Nkv) ? acc * a.scale : -3.0e38f;
    }
    float m = s[0];
#pragma unroll
    for (int j = 1; j < PKT; ++j) m = fmaxf(m, s[j]);
    m = fmaxf(m, __shfl_xor(m, 1));
    m = fmaxf(m, __shfl_xor(m, 2));
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < PKT; ++j) {
        s[j] = (kl + 4 * j < Nkv) ? __expf(s[j] - m) : 0.f;
        l += s[j];
    }
    l += __shfl_xor(l, 1);
    l += __shfl_xor(l, 2);
    const float inv = 1.0f / l;
#pragma unroll
    for (int j = 0; j < PKT; ++j)
        if (kl + 4 * j < Nkv) Ps[ql * Nkv + kl + 4 * j] = (T)(s[j] * inv);
    __syncthreads();
    // the tile is one contiguous run of nq * Nkv elements of P
    const int nq = min(PQ, a.Nq - q0);
    const int64_t g0 = ((int64_t)bh * a.Nq + q0) * Nkv;
    T* P = reinterpret_cast<T*>(a.p) + g0;
    const int total = nq * Nkv;
    if (((g0 * (int64_t)sizeof(T)) & 15) == 0 && ((uint64_t)a.p & 15) == 0) {
        const int nv = total / 8;
        for (int c = tid; c < nv; c += 256) st16(P + c * 8, ld16(Ps + c * 8));
        for (int c = nv * 8 + tid; c < total; c += 256) P[c] = Ps[c];
    } else {
        for (int c = tid; c < total; c += 256) P[c] = Ps[c];
    }
}

// ---- O = P V ------------------------------------------------------------------------------------------------------------------
// grid (ceil(Nq / 64), B * H); thread = (query tid >> 2, column lane cg = tid & 3) owning the float4 column chunks cg, cg + 4, ...
// of its query's output row (the four lanes of a query read 64 contiguous bytes of a V row per step). P rows in LDS as fp32 with
// an odd stride: the 16 queries of a wave instruction hit distinct banks.
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_pv_kernel(const ProbArgs a) {
    constexpr int NCH = D / 4, CPT = (NCH + 3) / 4;     // float4 chunks per row / per thread
    constexpr int LDP = PKMAX + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* Vs = reinterpret_cast<float*>(smem_raw);                 // [PKMAX][D]
    float* Pf = Vs + PKMAX * D;                                     // [PQ][LDP]
    const int tid = threadIdx.x, ql = tid >> 2, cg = tid & 3;
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int q0 = blockIdx.x * PQ, qi = q0 + ql;
    const int Nkv = a.Nkv;
    stage_rows<T, D, D>(Vs, reinterpret_cast<const T*>(a.kv) + (int64_t)b * a.kv_bs + (int64_t)h * D, a.kv_rs, Nkv, Nkv, tid);
    const int nq = min(PQ, a.Nq - q0);
    const T* P = reinterpret_cast<const T*>(a.p) + ((int64_t)bh * a.Nq + q0) * Nkv;
    for (int c = tid; c < nq * Nkv; c += 256) {                     // consecutive lanes, consecutive elements: coalesced
        const int r = c / Nkv;
        Pf[r * LDP + (c - r * Nkv)] = (float)P[c];
    }
    __syncthreads();
    if (qi >= a.Nq) return;
    f32x4 acc[CPT];
#pragma unroll
    for (int t = 0; t < CPT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* pr = Pf + ql * LDP;
    for (int j = 0; j < Nkv; ++j) {
        const float p = pr[j];
        const float* vr = Vs + j * D;
#pragma unroll
        for (int t = 0; t < CPT; ++t) {
            const int ch = cg + 4 * t;
            if (ch < NCH) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(vr + ch * 4);
                acc[t][0] = fmaf(p, v[0], acc[t][0]); acc[t][1] = fmaf(p, v[1], acc[t][1]);
                acc[t][2] = fmaf(p, v[2], acc[t][2]); acc[t][3] = fmaf(p, v[3], acc[t][3]);
            }
        }
    }
    T* op = reinterpret_cast<T*>(a.o) + (int64_t)b * a.o_bs + (int64_t)qi * a.o_rs + (int64_t)h * D;
#pragma unroll
    for (int t = 0; t < CPT; ++t) {
        const int ch = cg + 4 * t;
        if (ch < NCH) st8(op + ch * 4, pack4<T>(acc[t][0], acc[t][1], acc[t][2], acc[t][3]));
    }
}

// ---- backward of the split (round 6): a controller that takes the full map under autograd ---------------------------------
// The reference hands (B*H, N, 77) WITH grad to any controller (edlora.py:81-83; its own AttentionStore(training=True) keeps the
// maps and cal_attn_reg differentiates through them, ptp_util.py:37-53,79-82). Same split as the forward, same tiling
// (64 queries per workgroup, the <= 96-key side of the layer resident in LDS as fp32, VALU FMAs: AI ~ 20 flop/B):
//   attn_pv_bwd    : dP'[bh, q, j] = sum_c dO[q, c] V[j, c]            dense, `dtype` (what torch.bmm's backward gives the map)
//                    dV[j, c]      = sum_q P'[bh, q, j] dO[q, c]       per-workgroup fp32 partials -> ordered reduction
//   attn_probs_bwd : dS = P o (dP - rowsum(P o dP)) (softmax Jacobian, fp32), dQ[q, :] = scale * dS[q, :] K,
//                    dK[j, :] = scale * sum_q dS[q, j] Q[q, :]          per-workgroup fp32 partials -> ordered reduction
// The key-side sums over all queries are deterministic: every workgroup writes its 64-query partial [Nkv][D] to the caller's
// workspace and kv_partial_reduce_kernel adds them in block order.
struct ProbBwdArgs {
    const void* p; const void* dp_in; const void* x; const void* kv; void* dp_out; void* dx; float* ws;
    int B, H, Nq, Nkv;
    int64_t x_bs, x_rs, kv_bs, kv_rs, dx_bs, dx_rs;
    float scale;
};

// dense P tile (nq x Nkv contiguous) -> fp32 LDS [PQ][LDP], rows past nq zero (they enter the key-side sums)
template <typename T, int LDP>
__device__ __forceinline__ void stage_probs(float* Pf, const T* P, int nq, int Nkv, int tid) {
    for (int c = tid; c < PQ * Nkv; c += 256) {
        const int r = c / Nkv;
        Pf[r * LDP + (c - r * Nkv)] = (r < nq) ? (float)P[c] : 0.f;
    }
}

// key-side partial of this workgroup: out[j][c] = mul * sum_q W[q][j] * X[q][c]  (W [PQ][LDP], X [PQ][D] in LDS)
template <int D, int LDP>
__device__ __forceinline__ void key_side_partial(float* out, const float* W, const float* X, int Nkv, float mul, int tid) {
    constexpr int NCH = D / 4;
    for (int idx = tid; idx < Nkv * NCH; idx += 256) {
        const int j = idx / NCH, ch = idx - j * NCH;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int qq = 0; qq < PQ; ++qq) {
            const float w = W[qq * LDP + j];
            const f32x4 x = *reinterpret_cast<const f32x4*>(X + qq * D + ch * 4);
            acc[0] = fmaf(w, x[0], acc[0]); acc[1] = fmaf(w, x[1], acc[1]);
            acc[2] = fmaf(w, x[2], acc[2]); acc[3] = fmaf(w, x[3], acc[3]);
        }
        *reinterpret_cast<f32x4*>(out + (int64_t)j * D + ch * 4) = f32x4{acc[0] * mul, acc[1] * mul, acc[2] * mul, acc[3] * mul};
    }
}

template <typename T, int D>
__global__ __launch_bounds__(256) void attn_pv_bwd_kernel(const ProbBwdArgs a) {
    constexpr int LDK = D + 4, LDP = PKMAX + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* Vs = reinterpret_cast<float*>(smem_raw);                 // [PKMAX][LDK]
    float* dOs = Vs + PKMAX * LDK;                                  // [PQ][D]
    float* Pf = dOs + PQ * D;                                       // [PQ][LDP]
    T* Ps = reinterpret_cast<T*>(Pf + PQ * LDP);                    // [PQ][Nkv] dense: the global image of the dP tile
    const int tid = threadIdx.x, ql = tid >> 2, kl = tid & 3;
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int q0 = blockIdx.x * PQ, qi = q0 + ql;
    const int Nkv = a.Nkv, nq = min(PQ, a.Nq - q0);
    const T* dOb = reinterpret_cast<const T*>(a.x) + (int64_t)b * a.x_bs + (int64_t)h * D;
    stage_rows<T, D, LDK>(Vs, reinterpret_cast<const T*>(a.kv) + (int64_t)b * a.kv_bs + (int64_t)h * D, a.kv_rs, Nkv, PKMAX, tid);
    stage_rows<T, D, D>(dOs, dOb + (int64_t)q0 * a.x_rs, a.x_rs, nq, PQ, tid);
    const int64_t g0 = ((int64_t)bh * a.Nq + q0) * Nkv;
    stage_probs<T, LDP>(Pf, reinterpret_cast<const T*>(a.p) + g0, nq, Nkv, tid);
    float dov[D];
    {
        const T* dp = dOb + (int64_t)min(qi, a.Nq - 1) * a.x_rs;
#pragma unroll
        for (int c = 0; c < D; c += 8) unpack8<T>(ld16(dp + c), dov + c);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PKT; ++j) {
        const float* vr = Vs + (kl + 4 * j) * LDK;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            const f32x4 vv = *reinterpret_cast<const f32x4*>(vr + c);
            acc = fmaf(dov[c], vv[0], acc); acc = fmaf(dov[c + 1], vv[1], acc);
            acc = fmaf(dov[c + 2], vv[2], acc); acc = fmaf(dov[c + 3], vv[3], acc);
        }
        if (kl + 4 * j < Nkv) Ps[ql * Nkv + kl + 4 * j] = (T)acc;
    }
    __syncthreads();
    {
        T* P = reinterpret_cast<T*>(a.dp_out) + g0;
        const int total = nq * Nkv;
        if (((g0 * (int64_t)sizeof(T)) & 15) == 0 && ((uint64_t)a.dp_out & 15) == 0) {
            const int nv = total / 8;
            for (int c = tid; c < nv; c += 256) st16(P + c * 8, ld16(Ps + c * 8));
            for (int c = nv * 8 + tid; c < total; c += 256) P[c] = Ps[c];
        } else {
            for (int c = tid; c < total; c += 256) P[c] = Ps[c];
        }
    }
    key_side_partial<D, LDP>(a.ws + ((int64_t)blockIdx.x * gridDim.y + bh) * Nkv * D, Pf, dOs, Nkv, 1.0f, tid);
}

template <typename T, int D>
__global__ __launch_bounds__(256) void attn_probs_bwd_kernel(const ProbBwdArgs a) {
    constexpr int LDK = D + 4, LDP = PKMAX + 1, NCH = D / 4, CPT = (NCH + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* Ks = reinterpret_cast<float*>(smem_raw);                 // [PKMAX][LDK]
    float* Qs = Ks + PKMAX * LDK;                                   // [PQ][D]
    float* Pf = Qs + PQ * D;                                        // [PQ][LDP]  P
    float* dSf = Pf + PQ * LDP;                                     // [PQ][LDP]  dP, then dS (unscaled)
    const int tid = threadIdx.x, ql = tid >> 2, kl = tid & 3;
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int q0 = blockIdx.x * PQ, qi = q0 + ql;
    const int Nkv = a.Nkv, nq = min(PQ, a.Nq - q0);
    stage_rows<T, D, LDK>(Ks, reinterpret_cast<const T*>(a.kv) + (int64_t)b * a.kv_bs + (int64_t)h * D, a.kv_rs, Nkv, PKMAX, tid);
    stage_rows<T, D, D>(Qs, reinterpret_cast<const T*>(a.x) + (int64_t)b * a.x_bs + (int64_t)h * D + (int64_t)q0 * a.x_rs, a.x_rs, nq, PQ, tid);
    const int64_t g0 = ((int64_t)bh * a.Nq + q0) * Nkv;
    stage_probs<T, LDP>(Pf, reinterpret_cast<const T*>(a.p) + g0, nq, Nkv, tid);
    stage_probs<T, LDP>(dSf, reinterpret_cast<const T*>(a.dp_in) + g0, nq, Nkv, tid);
    __syncthreads();
    {
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < PKT; ++j) {
            const int key = kl + 4 * j;
            if (key < Nkv) dot = fmaf(Pf[ql * LDP + key], dSf[ql * LDP + key], dot);
        }
        dot += __shfl_xor(dot, 1);
        dot += __shfl_xor(dot, 2);
#pragma unroll
        for (int j = 0; j < PKT; ++j) {
            const int key = kl + 4 * j;
            if (key < Nkv) dSf[ql * LDP + key] = Pf[ql * LDP + key] * (dSf[ql * LDP + key] - dot);   // own elements only
        }
    }
    __syncthreads();
    if (qi < a.Nq) {           // dQ row: thread = (query, column lane), as attn_pv_kernel
        const int cg = kl;
        f32x4 acc[CPT];
#pragma unroll
        for (int t = 0; t < CPT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* dr = dSf + ql * LDP;
        for (int j = 0; j < Nkv; ++j) {
            const float w = dr[j];
            const float* kr = Ks + j * LDK;
#pragma unroll
            for (int t = 0; t < CPT; ++t) {
                const int ch = cg + 4 * t;
                if (ch < NCH) {
                    const f32x4 kk = *reinterpret_cast<const f32x4*>(kr + ch * 4);
                    acc[t][0] = fmaf(w, kk[0], acc[t][0]); acc[t][1] = fmaf(w, kk[1], acc[t][1]);
                    acc[t][2] = fmaf(w, kk[2], acc[t][2]); acc[t][3] = fmaf(w, kk[3], acc[t][3]);
                }
            }
        }
        T* dq = reinterpret_cast<T*>(a.dx) + (int64_t)b * a.dx_bs + (int64_t)qi * a.dx_rs + (int64_t)h * D;
#pragma unroll
        for (int t = 0; t < CPT; ++t) {
            const int ch = cg + 4 * t;
            if (ch < NCH) st8(dq + ch * 4, pack4<T>(acc[t][0] * a.scale, acc[t][1] * a.scale, acc[t][2] * a.scale, acc[t][3] * a.scale));
        }
    }
    key_side_partial<D, LDP>(a.ws + ((int64_t)blockIdx.x * gridDim.y + bh) * Nkv * D, dSf, Qs, Nkv, a.scale, tid);
}

// out[b, j, h*D + c] = sum over query blocks (in block order) of ws[blk][b*H + h][j][c]
template <typename T>
__global__ void kv_partial_reduce_kernel(const float* __restrict__ ws, int nblk, int BH, int H, int Nkv, int D, T* __restrict__ out,
                                         int64_t o_bs, int64_t o_rs) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // (bh, j, c4)
    const int nch = D / 4;
    if (idx >= (int64_t)BH * Nkv * nch) return;
    const int ch = idx % nch;
    const int64_t r = idx / nch;
    const int j = r % Nkv, bh = r / Nkv;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int blk = 0; blk < nblk; ++blk) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ws + (((int64_t)blk * BH + bh) * Nkv + j) * D + ch * 4);
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    const int b = bh / H, h = bh - b * H;
    st8(out + (int64_t)b * o_bs + (int64_t)j * o_rs + (int64_t)h * D + ch * 4, pack4<T>(acc[0], acc[1], acc[2], acc[3]));
}

template <typename T, int D>
int launch_pv_bwd(const ProbBwdArgs& a, T* dv, int64_t dv_bs, int64_t dv_rs, hipStream_t st) {
    const size_t lds = ((size_t)PKMAX * (D + 4) + (size_t)PQ * D + (size_t)PQ * (PKMAX + 1)) * sizeof(float) + (size_t)PQ * PKMAX * sizeof(T);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_pv_bwd_kernel<T, D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const unsigned nqb = (unsigned)((a.Nq + PQ - 1) / PQ);
    hipLaunchKernelGGL((attn_pv_bwd_kernel<T, D>), dim3(nqb, (unsigned)(a.B * a.H)), dim3(256), lds, st, a);
    const int64_t n = (int64_t)a.B * a.H * a.Nkv * (D / 4);
    hipLaunchKernelGGL((kv_partial_reduce_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.ws, (int)nqb, a.B * a.H, a.H,
                       a.Nkv, D, dv, dv_bs, dv_rs);
    return mos_check_launch("attn_pv_bwd");
}
template <typename T, int D>
int launch_probs_bwd(const ProbBwdArgs& a, T* dk, int64_t dk_bs, int64_t dk_rs, hipStream_t st) {
    const size_t lds = ((size_t)PKMAX * (D + 4) + (size_t)PQ * D + 2 * (size_t)PQ * (PKMAX + 1)) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_probs_bwd_kernel<T, D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const unsigned nqb = (unsigned)((a.Nq + PQ - 1) / PQ);
    hipLaunchKernelGGL((attn_probs_bwd_kernel<T, D>), dim3(nqb, (unsigned)(a.B * a.H)), dim3(256), lds, st, a);
    const int64_t n = (int64_t)a.B * a.H * a.Nkv * (D / 4);
    hipLaunchKernelGGL((kv_partial_reduce_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.ws, (int)nqb, a.B * a.H, a.H,
                       a.Nkv, D, dk, dk_bs, dk_rs);
    return mos_check_launch("attn_probs_bwd");
}
template <typename T>
int dispatch_bwd(bool pv, const ProbBwdArgs& a, void* dkv, int64_t bs, int64_t rs, int d, hipStream_t st) {
    T* out = reinterpret_cast<T*>(dkv);
    if (d == 40) return pv ? launch_pv_bwd<T, 40>(a, out, bs, rs, st) : launch_probs_bwd<T, 40>(a, out, bs, rs, st);
    if (d == 80) return pv ? launch_pv_bwd<T, 80>(a, out, bs, rs, st) : launch_probs_bwd<T, 80>(a, out, bs, rs, st);
    if (d == 160) return pv ? launch_pv_bwd<T, 160>(a, out, bs, rs, st) : launch_probs_bwd<T, 160>(a, out, bs, rs, st);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_attn_probs_bwd / mos_attn_pv_bwd: head dim %d (supported: 40, 80, 160)", d);
}

template <typename T, int D>
int launch_probs(const ProbArgs& a, hipStream_t st) {
    const size_t lds = (size_t)PKMAX * (D + 4) * sizeof(float) + (size_t)PQ * PKMAX * sizeof(T);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_probs_kernel<T, D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attn_probs_kernel<T, D>), dim3((unsigned)((a.Nq + PQ - 1) / PQ), (unsigned)(a.B * a.H)), dim3(256), lds, st, a);
    return mos_check_launch("attn_probs");
}
template <typename T, int D>
int launch_pv(const ProbArgs& a, hipStream_t st) {
    const size_t lds = (size_t)PKMAX * D * sizeof(float) + (size_t)PQ * (PKMAX + 1) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_pv_kernel<T, D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attn_pv_kernel<T, D>), dim3((unsigned)((a.Nq + PQ - 1) / PQ), (unsigned)(a.B * a.H)), dim3(256), lds, st, a);
    return mos_check_launch("attn_pv");
}

template <typename T>
int dispatch(bool pv, const ProbArgs& a, int d, hipStream_t st) {
    if (d == 40) return pv ? launch_pv<T, 40>(a, st) : launch_probs<T, 40>(a, st);
    if (d == 80) return pv ? launch_pv<T, 80>(a, st) : launch_probs<T, 80>(a, st);
    if (d == 160) return pv ? launch_pv<T, 160>(a, st) : launch_probs<T, 160>(a, st);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_attn_probs / mos_attn_pv: head dim %d (supported: 40, 80, 160)", d);
}

int check_shape(const char* who, const mos_attn_shape* s) {
    MOS_REQUIRE(s != nullptr, "%s: NULL shape", who);
    MOS_REQUIRE(s->B > 0 && s->H > 0 && s->Nq > 0 && s->Nkv > 0 && s->Nkv <= PKMAX, "%s: B=%d H=%d Nq=%d Nkv=%d (1 <= Nkv <= %d)", who,
                s->B, s->H, s->Nq, s->Nkv, PKMAX);
    MOS_REQUIRE(!s->causal, "%s: no causal mask on this path", who);
    MOS_REQUIRE((int64_t)s->B * s->H <= 65535, "%s: B*H = %lld exceeds the grid's y range", who, (long long)s->B * s->H);
    return MOS_OK;
}

}  // namespace

extern "C" {

int mos_attn_probs(const void* q, const void* k, void* probs, const mos_attn_shape* s, int dtype, void* stream) {
    MOS_REQUIRE(q && k && probs, "mos_attn_probs: NULL argument");
    int rc = check_shape("mos_attn_probs", s);
    if (rc) return rc;
    MOS_REQUIRE(s->q_rs % 8 == 0 && s->q_bs % 8 == 0 && s->k_rs % 8 == 0 && s->k_bs % 8 == 0 && ((uint64_t)q & 15) == 0 && ((uint64_t)k & 15) == 0,
                "mos_attn_probs: q / k rows must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char key[96];
    snprintf(key, sizeof(key), "%s d%d B%d H%d Nq%d Nkv%d", dtype == MOS_F16 ? "f16" : "bf16", s->d, s->B, s->H, s->Nq, s->Nkv);
    const double bhn = (double)s->B * s->H * s->Nq;
    MosProfScope prof(st, "attn_probs", key, 2.0 * bhn * s->Nkv * s->d, 2.0 * (bhn * s->d + (double)s->B * s->H * s->Nkv * s->d + bhn * s->Nkv));
    ProbArgs a;
    a.q = q; a.kv = k; a.p = probs; a.o = nullptr;
    a.B = s->B; a.H = s->H; a.Nq = s->Nq; a.Nkv = s->Nkv;
    a.q_bs = s->q_bs; a.q_rs = s->q_rs; a.kv_bs = s->k_bs; a.kv_rs = s->k_rs; a.o_bs = a.o_rs = 0;
    a.scale = s->scale;
    if (dtype == MOS_F16) return dispatch<f16_t>(false, a, s->d, st);
    if (dtype == MOS_BF16) return dispatch<bf16_t>(false, a, s->d, st);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_attn_probs: dtype %d", dtype);
}

int mos_attn_pv(const void* probs, const void* v, void* o, const mos_attn_shape* s, int dtype, void* stream) {
    MOS_REQUIRE(probs && v && o, "mos_attn_pv: NULL argument");
    int rc = check_shape("mos_attn_pv", s);
    if (rc) return rc;
    MOS_REQUIRE(s->v_rs % 8 == 0 && s->v_bs % 8 == 0 && s->o_rs % 4 == 0 && s->o_bs % 4 == 0 && ((uint64_t)v & 15) == 0 && ((uint64_t)o & 7) == 0,
                "mos_attn_pv: v rows must be 16-byte aligned, o rows 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char key[96];
    snprintf(key, sizeof(key), "%s d%d B%d H%d Nq%d Nkv%d", dtype == MOS_F16 ? "f16" : "bf16", s->d, s->B, s->H, s->Nq, s->Nkv);
    const double bhn = (double)s->B * s->H * s->Nq;
    MosProfScope prof(st, "attn_pv", key, 2.0 * bhn * s->Nkv * s->d, 2.0 * (bhn * s->d + (double)s->B * s->H * s->Nkv * s->d + bhn * s->Nkv));
    ProbArgs a;
    a.q = nullptr; a.kv = v; a.p = const_cast<void*>(probs); a.o = o;
    a.B = s->B; a.H = s->H; a.Nq = s->Nq; a.Nkv = s->Nkv;
    a.q_bs = a.q_rs = 0; a.kv_bs = s->v_bs; a.kv_rs = s->v_rs; a.o_bs = s->o_bs; a.o_rs = s->o_rs;
    a.scale = s->scale;
    if (dtype == MOS_F16) return dispatch<f16_t>(true, a, s->d, st);
    if (dtype == MOS_BF16) return dispatch<bf16_t>(true, a, s->d, st);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_attn_pv: dtype %d", dtype);
}

int64_t mos_attn_probs_bwd_workspace_bytes(const mos_attn_shape* s) {
    if (!s || s->Nq <= 0 || s->Nkv <= 0 || s->B <= 0 || s->H <= 0 || s->d <= 0) return 0;
    return (int64_t)((s->Nq + PQ - 1) / PQ) * s->B * s->H * s->Nkv * s->d * (int64_t)sizeof(float);
}

int mos_attn_pv_bwd(const void* probs, const void* v, const void* dO, void* dprobs, void* dv, void* ws, const mos_attn_shape* s,
                    const mos_attn_grad_strides* g, int dtype, void* stream) {
    MOS_REQUIRE(probs && v && dO && dprobs && dv && ws && g, "mos_attn_pv_bwd: NULL argument");
    int rc = check_shape("mos_attn_pv_bwd", s);
    if (rc) return rc;
    MOS_REQUIRE(s->v_rs % 8 == 0 && s->v_bs % 8 == 0 && g->do_rs % 8 == 0 && g->do_bs % 8 == 0 && g->dv_rs % 4 == 0 && g->dv_bs % 4 == 0 &&
                    ((uint64_t)v & 15) == 0 && ((uint64_t)dO & 15) == 0 && ((uint64_t)dv & 7) == 0,
                "mos_attn_pv_bwd: v / dO rows must be 16-byte aligned, dv rows 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char key[96];
    snprintf(key, sizeof(key), "%s d%d B%d H%d Nq%d Nkv%d", dtype == MOS_F16 ? "f16" : "bf16", s->d, s->B, s->H, s->Nq, s->Nkv);
    const double bhn = (double)s->B * s->H * s->Nq;
    MosProfScope prof(st, "attn_pv_bwd", key, 4.0 * bhn * s->Nkv * s->d, 2.0 * (bhn * s->d + 2.0 * bhn * s->Nkv));
    ProbBwdArgs a;
    a.p = probs; a.dp_in = nullptr; a.x = dO; a.kv = v; a.dp_out = dprobs; a.dx = nullptr; a.ws = reinterpret_cast<float*>(ws);
    a.B = s->B; a.H = s->H; a.Nq = s->Nq; a.Nkv = s->Nkv;
    a.x_bs = g->do_bs; a.x_rs = g->do_rs; a.kv_bs = s->v_bs; a.kv_rs = s->v_rs; a.dx_bs = a.dx_rs = 0;
    a.scale = 1.0f;
    if (dtype == MOS_F16) return dispatch_bwd<f16_t>(true, a, dv, g->dv_bs, g->dv_rs, s->d, st);
    if (dtype == MOS_BF16) return dispatch_bwd<bf16_t>(true, a, dv, g->dv_bs, g->dv_rs, s->d, st);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_attn_pv_bwd: dtype %d", dtype);
}

int mos_attn_probs_bwd(const void* q, const void* k, const void* probs, const void* dprobs, void* dq, void* dk, void* ws,
                       const mos_attn_shape* s, const mos_attn_grad_strides* g, int dtype, void* stream) {
    MOS_REQUIRE(q && k && probs && dprobs && dq && dk && ws && g, "mos_attn_probs_bwd: NULL argument");
    int rc = check_shape("mos_attn_probs_bwd", s);
    if (rc) return rc;
    MOS_REQUIRE(s->q_rs % 8 == 0 && s->q_bs % 8 == 0 && s->k_rs % 8 == 0 && s->k_bs % 8 == 0 && g->dq_rs % 4 == 0 && g->dq_bs % 4 == 0 &&
                    g->dk_rs % 4 == 0 && g->dk_bs % 4 == 0 && ((uint64_t)q & 15) == 0 && ((uint64_t)k & 15) == 0 && ((uint64_t)dq & 7) == 0 &&
                    ((uint64_t)dk & 7) == 0,
                "mos_attn_probs_bwd: q / k rows must be 16-byte aligned, dq / dk rows 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char key[96];
    snprintf(key, sizeof(key), "%s d%d B%d H%d Nq%d Nkv%d", dtype == MOS_F16 ? "f16" : "bf16", s->d, s->B, s->H, s->Nq, s->Nkv);
    const double bhn = (double)s->B * s->H * s->Nq;
    MosProfScope prof(st, "attn_probs_bwd", key, 4.0 * bhn * s->Nkv * s->d, 2.0 * (2.0 * bhn * s->d + 2.0 * bhn * s->Nkv));
    ProbBwdArgs a;
    a.p = probs; a.dp_in = dprobs; a.x = q; a.kv = k; a.dp_out = nullptr; a.dx = dq; a.ws = reinterpret_cast<float*>(ws);
    a.B = s->B; a.H = s->H; a.Nq = s->Nq; a.Nkv = s->Nkv;
    a.x_bs = s->q_bs; a.x_rs = s->q_rs; a.kv_bs = s->k_bs; a.kv_rs = s->k_rs; a.dx_bs = g->dq_bs; a.dx_rs = g->dq_rs;
    a.scale = s->scale;
    if (dtype == MOS_F16) return dispatch_bwd<f16_t>(false, a, dk, g->dk_bs, g->dk_rs, s->d, st);
    if (dtype == MOS_BF16) return dispatch_bwd<bf16_t>(false, a, dk, g->dk_bs, g->dk_rs, s->d, st);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_attn_probs_bwd: dtype %d", dtype);
}

}  // extern "C"

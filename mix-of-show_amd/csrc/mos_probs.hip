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
// (B*H, Nq, Nkv), the reference's layout. Nkv <= 96 (text keys), d in {40, 80, 160}; inference only (no backward: the
// training-time controller, AttentionStore(training=True), takes the probability-column path of mos_attn.hip).
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

}  // extern "C"

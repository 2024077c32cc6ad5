// mos_elem.hip — fused row-wise operators of the transformer blocks that CALL the attention path:
//   LayerNorm  (BasicTransformerBlock.norm1/2/3, CLIP layer_norm1/2/final): half in, half out, fp32 statistics. Under
//              autocast the reference stack runs it as cast-to-fp32 + fp32 layer_norm + cast back at the next GEMM
//              (3 passes forward, 4 backward); here one pass each way. The affine parameters are frozen in ED-LoRA
//              training: no dgamma / dbeta.
//   GEGLU      (FeedForward.net[0]): y = h[:, :F] * gelu(h[:, F:]) (exact erf GELU) and its backward, one pass each
//              instead of chunk + gelu + mul (+ 4 kernels backward).
// HBM-bound by construction: bytes = 2 x rows x C x 2 B forward (LayerNorm), 3 x backward.
#include <cstdio>
#include <type_traits>
#include "mos_common.h"

namespace {

constexpr int LN_MAX_CHUNKS = 4;     // 16 B chunks per lane: C <= 64 * 8 * 4 = 2048

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// one wave per row; the row stays in registers between the statistics and the normalisation
template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ stats, int rows, int C, float eps) {
    typedef typename MT<T>::v8 v8;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = C >> 3;
    const T* xr = x + (int64_t)row * C;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + 64 * k;
        if (ch < nch) {
            const v8 t = as_v8<T>(ld16(xr + ch * 8));
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[k][i] = (float)t[i]; s += v[k][i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[k][i] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        if (lane + 64 * k < nch) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = v[k][i] - mean; ss += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
    if (stats != nullptr && lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
    T* yr = y + (int64_t)row * C;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + 64 * k;
        if (ch < nch) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + ch * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + ch * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + ch * 8), b1 = *reinterpret_cast<const f32x4*>(beta + ch * 8 + 4);
            v8 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] = (T)((v[k][i] - mean) * rstd * g0[i] + b0[i]);
                o[i + 4] = (T)((v[k][i + 4] - mean) * rstd * g1[i] + b1[i]);
            }
            st16(yr + ch * 8, from_v8<T>(o));
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ stats,
                                                            T* __restrict__ dx, int rows, int C) {
    typedef typename MT<T>::v8 v8;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = C >> 3;
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    const T* xr = x + (int64_t)row * C;
    const T* dr = dy + (int64_t)row * C;
    float g[NCH][8], xh[NCH][8];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + 64 * k;
        if (ch < nch) {
            const v8 xv = as_v8<T>(ld16(xr + ch * 8)), dv = as_v8<T>(ld16(dr + ch * 8));
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + ch * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + ch * 8 + 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xh[k][i] = ((float)xv[i] - mean) * rstd;
                g[k][i] = (float)dv[i] * (i < 4 ? g0[i & 3] : g1[i & 3]);
                sg += g[k][i];
                sgx += g[k][i] * xh[k][i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) { g[k][i] = 0.f; xh[k][i] = 0.f; }
        }
    }
    const float mg = wave_sum(sg) / (float)C, mgx = wave_sum(sgx) / (float)C;
    T* or_ = dx + (int64_t)row * C;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + 64 * k;
        if (ch < nch) {
            v8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (T)(rstd * (g[k][i] - mg - xh[k][i] * mgx));
            st16(or_ + ch * 8, from_v8<T>(o));
        }
    }
}

// ---- residual add fused into LayerNorm ------------------------------------------------------------------------------
// A transformer block is `x = x + f(LN(x))` three times over: the residual sum is an input of the NEXT LayerNorm only, and in
// backward the LayerNorm's input gradient is added to the gradient that bypasses it. As separate kernels that is one ATen add
// per residual forward and one per LayerNorm backward (2 x 48 launches per SD-1.5 UNet step, 4 x 24 + casts in the CLIP tower
// whose residual stream is fp32 under autocast). Here:
//     forward   s = x + r  (written once, stream dtype S),   y = LN(s)          (RES; without RES: y = LN(x), nothing else)
//     backward  dx = LN_bwd(dy) + ds  (stream dtype S)  [+ a half copy of dx for the half branch of an fp32 stream]
// S == T (half stream, UNet): s and dx are rounded exactly where the unfused kernels round (s = half(x + r); dx =
// half(half(LN_bwd) + ds)), so results are bit-identical to add + layernorm. S == float (CLIP): statistics and gradient
// are taken from the fp32 sum itself, as the reference's fp32 layer_norm under autocast does.
template <typename S> struct Row8;
template <> struct Row8<float> {
    static __device__ __forceinline__ void load(const float* p, float* v) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[i + 4] = b[i]; }
    }
    static __device__ __forceinline__ void store(float* p, const float* v) {
        f32x4 a, b;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[i + 4]; }
        *reinterpret_cast<f32x4*>(p) = a;
        *reinterpret_cast<f32x4*>(p + 4) = b;
    }
    static __device__ __forceinline__ float round(float v) { return v; }
};
template <typename T> struct Row8 {
    typedef typename MT<T>::v8 v8;
    static __device__ __forceinline__ void load(const T* p, float* v) {
        const v8 t = as_v8<T>(ld16(p));
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
    }
    static __device__ __forceinline__ void store(T* p, const float* v) {
        v8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (T)v[i];
        st16(p, from_v8<T>(o));
    }
    static __device__ __forceinline__ float round(float v) { return (float)(T)v; }
};

template <typename T, typename S, int NCH, bool RES>
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const S* __restrict__ x, const T* __restrict__ r,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                S* __restrict__ s_out, T* __restrict__ y,
                                                                float* __restrict__ stats, int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = C >> 3;
    const S* xr = x + (int64_t)row * C;
    float v[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + 64 * k;
        if (ch < nch) {
            Row8<S>::load(xr + ch * 8, v[k]);
            if constexpr (RES) {
                float rv[8];
                Row8<T>::load(r + (int64_t)row * C + ch * 8, rv);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[k][i] = Row8<S>::round(v[k][i] + rv[i]);
                Row8<S>::store(s_out + (int64_t)row * C + ch * 8, v[k]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += v[k][i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[k][i] = 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        if (lane + 64 * k < nch) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = v[k][i] - mean; ss += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
    if (stats != nullptr && lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
    T* yr = y + (int64_t)row * C;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + 64 * k;
        if (ch < nch) {
            float g[8], b[8], o[8];
            Row8<float>::load(gamma + ch * 8, g);
            Row8<float>::load(beta + ch * 8, b);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (v[k][i] - mean) * rstd * g[i] + b[i];
            Row8<T>::store(yr + ch * 8, o);
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) (+ ds),  g = dy * gamma, xhat from the saved sum s and (mean, rstd)
template <typename T, typename S, int NCH, bool HAS_DS, bool HALF_COPY>
__global__ __launch_bounds__(256) void add_layernorm_bwd_kernel(const T* __restrict__ dy, const S* __restrict__ ds,
                                                                const S* __restrict__ s, const float* __restrict__ gamma,
                                                                const float* __restrict__ stats, S* __restrict__ dx,
                                                                T* __restrict__ dx_half, int rows, int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = C >> 3;
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    const int64_t base = (int64_t)row * C;
    float g[NCH][8], xh[NCH][8];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + 64 * k;
        if (ch < nch) {
            float sv[8], dv[8], gm[8];
            Row8<S>::load(s + base + ch * 8, sv);
            Row8<T>::load(dy + base + ch * 8, dv);
            Row8<float>::load(gamma + ch * 8, gm);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xh[k][i] = (sv[i] - mean) * rstd;
                g[k][i] = dv[i] * gm[i];
                sg += g[k][i];
                sgx += g[k][i] * xh[k][i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) { g[k][i] = 0.f; xh[k][i] = 0.f; }
        }
    }
    const float mg = wave_sum(sg) / (float)C, mgx = wave_sum(sgx) / (float)C;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + 64 * k;
        if (ch < nch) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = rstd * (g[k][i] - mg - xh[k][i] * mgx);
            if constexpr (HAS_DS) {
                float dsv[8];
                Row8<S>::load(ds + base + ch * 8, dsv);
                // half stream: the unfused path rounds LN_bwd to half before autograd adds the bypass gradient
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = Row8<S>::round(o[i]) + dsv[i];
            }
            Row8<S>::store(dx + base + ch * 8, o);
            if constexpr (HALF_COPY) Row8<T>::store(dx_half + base + ch * 8, o);
        }
    }
}

__device__ __forceinline__ float gelu_f(float z) { return 0.5f * z * (1.f + erff(z * 0.70710678118654752f)); }
// d/dz [z * Phi(z)] = Phi(z) + z * phi(z)
__device__ __forceinline__ float gelu_grad_f(float z) {
    const float cdf = 0.5f * (1.f + erff(z * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * z * z);
    return cdf + z * pdf;
}

// h (rows, 2F): value | gate. One thread = 8 outputs.
template <typename T>
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const T* __restrict__ h, T* __restrict__ y, int64_t nvec, int F) {
    typedef typename MT<T>::v8 v8;
    const int fv = F >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / fv;
        const int c = (int)(i - row * fv) * 8;
        const T* hr = h + row * 2 * (int64_t)F;
        const v8 a = as_v8<T>(ld16(hr + c)), gt = as_v8<T>(ld16(hr + F + c));
        v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (T)((float)a[e] * gelu_f((float)gt[e]));
        st16(y + row * (int64_t)F + c, from_v8<T>(o));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ h, T* __restrict__ dh,
                                                        int64_t nvec, int F) {
    typedef typename MT<T>::v8 v8;
    const int fv = F >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / fv;
        const int c = (int)(i - row * fv) * 8;
        const T* hr = h + row * 2 * (int64_t)F;
        const v8 a = as_v8<T>(ld16(hr + c)), gt = as_v8<T>(ld16(hr + F + c)), d = as_v8<T>(ld16(dy + row * (int64_t)F + c));
        v8 da, dg;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float z = (float)gt[e], dd = (float)d[e];
            da[e] = (T)(dd * gelu_f(z));
            dg[e] = (T)(dd * (float)a[e] * gelu_grad_f(z));
        }
        T* dr = dh + row * 2 * (int64_t)F;
        st16(dr + c, from_v8<T>(da));
        st16(dr + F + c, from_v8<T>(dg));
    }
}

// y[r, :] = softmax(scale * x[r, :]) for half rows of up to 32768 elements; one 256-thread block per row, the row stays
// in registers (VAE mid-block attention: 4096 x 4096 single-head scores between two library GEMMs at 512 x 512; 32768 keys for
// the 1024 x 2048 images of the reference's regionally_sample.sh: 16 vectors per thread).
template <typename T, int NCH>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int N, float scale_log2e) {
    typedef typename MT<T>::v8 v8;
    __shared__ float red[8];
    const int tid = threadIdx.x;
    const int nch = N >> 3;
    const T* xr = x + (int64_t)blockIdx.x * N;
    T* yr = y + (int64_t)blockIdx.x * N;
    float v[NCH][8];
    float mx = -1.0e30f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = tid + 256 * k;
        if (ch < nch) {
            const v8 t = as_v8<T>(ld16(xr + ch * 8));
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[k][i] = (float)t[i] * scale_log2e; mx = fmaxf(mx, v[k][i]); }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[k][i] = -1.0e30f;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[k][i] = __builtin_amdgcn_exp2f(v[k][i] - mx); sum += v[k][i]; }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = tid + 256 * k;
        if (ch < nch) {
            v8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (T)(v[k][i] * inv);
            st16(yr + ch * 8, from_v8<T>(o));
        }
    }
}

template <typename T>
int softmax_rows(const void* x, void* y, int rows, int N, float scale, hipStream_t st) {
    const int nch = (N / 8 + 255) / 256;
    char key[64];
    snprintf(key, sizeof(key), "rows%d N%d", rows, N);
    MosProfScope prof(st, "softmax_rows", key, 5.0 * rows * N, 4.0 * rows * (double)N);
    const float sl = scale * 1.4426950408889634f;
#define SM_K(NC) hipLaunchKernelGGL((softmax_rows_kernel<T, NC>), dim3(rows), dim3(256), 0, st, (const T*)x, (T*)y, N, sl)
    if (nch <= 4) { switch (nch) { case 1: SM_K(1); break; case 2: SM_K(2); break; case 3: SM_K(3); break; default: SM_K(4); break; } }
    else if (nch <= 8) SM_K(8);
    else SM_K(16);
#undef SM_K
    return mos_check_launch("softmax_rows");
}

template <typename T>
int ln_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int rows, int C, float eps,
           hipStream_t st) {
    const int nch = (C / 8 + 63) / 64;
    const dim3 grid((rows + 3) / 4);
    char key[64];
    snprintf(key, sizeof(key), "rows%d C%d", rows, C);
    MosProfScope prof(st, "layernorm_fwd", key, 8.0 * rows * C, 4.0 * rows * (double)C);
#define LN_F(N) hipLaunchKernelGGL((layernorm_fwd_kernel<T, N>), grid, dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, stats, rows, C, eps)
    switch (nch) { case 1: LN_F(1); break; case 2: LN_F(2); break; case 3: LN_F(3); break; default: LN_F(4); break; }
#undef LN_F
    return mos_check_launch("layernorm_fwd");
}

template <typename T>
int ln_bwd(const void* dy, const void* x, const float* gamma, const float* stats, void* dx, int rows, int C, hipStream_t st) {
    const int nch = (C / 8 + 63) / 64;
    const dim3 grid((rows + 3) / 4);
    char key[64];
    snprintf(key, sizeof(key), "rows%d C%d", rows, C);
    MosProfScope prof(st, "layernorm_bwd", key, 12.0 * rows * C, 6.0 * rows * (double)C);
#define LN_B(N) hipLaunchKernelGGL((layernorm_bwd_kernel<T, N>), grid, dim3(256), 0, st, (const T*)dy, (const T*)x, gamma, stats, (T*)dx, rows, C)
    switch (nch) { case 1: LN_B(1); break; case 2: LN_B(2); break; case 3: LN_B(3); break; default: LN_B(4); break; }
#undef LN_B
    return mos_check_launch("layernorm_bwd");
}

template <typename T, typename S>
int add_ln_fwd(const void* x, const void* r, const float* gamma, const float* beta, void* s_out, void* y, float* stats, int rows,
               int C, float eps, hipStream_t st) {
    const int nch = (C / 8 + 63) / 64;
    const dim3 grid((rows + 3) / 4);
    char key[64];
    snprintf(key, sizeof(key), "rows%d C%d s%d", rows, C, (int)sizeof(S));
    const double sb = (double)sizeof(S), tb = (double)sizeof(T);
    MosProfScope prof(st, "add_layernorm_fwd", key, 9.0 * rows * C, rows * (double)C * (r ? 2 * sb + 2 * tb : sb + tb));
#define ALN_F(N, R) hipLaunchKernelGGL((add_layernorm_fwd_kernel<T, S, N, R>), grid, dim3(256), 0, st, (const S*)x, (const T*)r, \
                                       gamma, beta, (S*)s_out, (T*)y, stats, rows, C, eps)
#define ALN_FN(N) do { if (r != nullptr) ALN_F(N, true); else ALN_F(N, false); } while (0)
    switch (nch) { case 1: ALN_FN(1); break; case 2: ALN_FN(2); break; case 3: ALN_FN(3); break; default: ALN_FN(4); break; }
#undef ALN_FN
#undef ALN_F
    return mos_check_launch("add_layernorm_fwd");
}

template <typename T, typename S>
int add_ln_bwd(const void* dy, const void* ds, const void* s, const float* gamma, const float* stats, void* dx, void* dx_half,
               int rows, int C, hipStream_t st) {
    const int nch = (C / 8 + 63) / 64;
    const dim3 grid((rows + 3) / 4);
    char key[64];
    snprintf(key, sizeof(key), "rows%d C%d s%d", rows, C, (int)sizeof(S));
    const double sb = (double)sizeof(S), tb = (double)sizeof(T);
    MosProfScope prof(st, "add_layernorm_bwd", key, 13.0 * rows * C,
                      rows * (double)C * (tb + 2 * sb + (ds ? sb : 0.0) + (dx_half ? tb : 0.0)));
#define ALN_B(N, D, H) hipLaunchKernelGGL((add_layernorm_bwd_kernel<T, S, N, D, H>), grid, dim3(256), 0, st, (const T*)dy, \
                                          (const S*)ds, (const S*)s, gamma, stats, (S*)dx, (T*)dx_half, rows, C)
#define ALN_BN(N) do { \
        if (ds != nullptr && dx_half != nullptr) ALN_B(N, true, true); \
        else if (ds != nullptr) ALN_B(N, true, false); \
        else if (dx_half != nullptr) ALN_B(N, false, true); \
        else ALN_B(N, false, false); } while (0)
    switch (nch) { case 1: ALN_BN(1); break; case 2: ALN_BN(2); break; case 3: ALN_BN(3); break; default: ALN_BN(4); break; }
#undef ALN_BN
#undef ALN_B
    return mos_check_launch("add_layernorm_bwd");
}

// quick-GELU of the CLIP text tower MLP (transformers `quick_gelu`: x * sigmoid(1.702 x)); the reference runs it as three
// elementwise torch kernels forward and ~five backward per layer. One thread = 8 elements.
__device__ __forceinline__ float sigmoid_f(float z) { return 1.f / (1.f + __expf(-z)); }
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void quick_gelu_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out,
                                                         int64_t nvec) {
    typedef typename MT<T>::v8 v8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const v8 a = as_v8<T>(ld16(x + i * 8));
        v8 o;
        if constexpr (BWD) {
            const v8 d = as_v8<T>(ld16(dy + i * 8));
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float z = (float)a[e], sg = sigmoid_f(1.702f * z);
                o[e] = (T)((float)d[e] * sg * (1.f + 1.702f * z * (1.f - sg)));     // d/dz [z sigmoid(1.702 z)]
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float z = (float)a[e];
                o[e] = (T)(z * sigmoid_f(1.702f * z));
            }
        }
        st16(out + i * 8, from_v8<T>(o));
    }
}

template <bool BWD>
int quick_gelu_launch(const void* x, const void* dy, void* out, int64_t n, int dtype, hipStream_t st) {
    const int64_t nvec = n / 8;
    const int blocks = (int)((nvec + 255) / 256 > 16384 ? 16384 : (nvec + 255) / 256);
    char key[64];
    snprintf(key, sizeof(key), "n%lld", (long long)n);
    MosProfScope prof(st, BWD ? "quick_gelu_bwd" : "quick_gelu_fwd", key, (BWD ? 12.0 : 8.0) * n, (BWD ? 6.0 : 4.0) * (double)n);
    if (dtype == MOS_F16)
        hipLaunchKernelGGL((quick_gelu_kernel<f16_t, BWD>), dim3(blocks), dim3(256), 0, st, (const f16_t*)x, (const f16_t*)dy, (f16_t*)out, nvec);
    else if (dtype == MOS_BF16)
        hipLaunchKernelGGL((quick_gelu_kernel<bf16_t, BWD>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)out, nvec);
    else return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_quick_gelu: dtype %d", dtype);
    return mos_check_launch(BWD ? "quick_gelu_bwd" : "quick_gelu_fwd");
}

}  // namespace

extern "C" {

int mos_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int rows, int C,
                      float eps, int dtype, void* stream) {
    MOS_REQUIRE(x && gamma && beta && y, "mos_layernorm_fwd: NULL argument");
    MOS_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * 8 * LN_MAX_CHUNKS, "mos_layernorm_fwd: rows=%d C=%d (C %% 8 == 0, C <= %d)",
                rows, C, 64 * 8 * LN_MAX_CHUNKS);
    if (dtype == MOS_F16) return ln_fwd<f16_t>(x, gamma, beta, y, stats, rows, C, eps, (hipStream_t)stream);
    if (dtype == MOS_BF16) return ln_fwd<bf16_t>(x, gamma, beta, y, stats, rows, C, eps, (hipStream_t)stream);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_layernorm_fwd: dtype %d", dtype);
}

int mos_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* stats, void* dx, int rows, int C,
                      int dtype, void* stream) {
    MOS_REQUIRE(dy && x && gamma && stats && dx, "mos_layernorm_bwd: NULL argument");
    MOS_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * 8 * LN_MAX_CHUNKS, "mos_layernorm_bwd: rows=%d C=%d", rows, C);
    if (dtype == MOS_F16) return ln_bwd<f16_t>(dy, x, gamma, stats, dx, rows, C, (hipStream_t)stream);
    if (dtype == MOS_BF16) return ln_bwd<bf16_t>(dy, x, gamma, stats, dx, rows, C, (hipStream_t)stream);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_layernorm_bwd: dtype %d", dtype);
}

int mos_add_layernorm_fwd(const void* x, const void* r, const float* gamma, const float* beta, void* s_out, void* y,
                          float* stats, int rows, int C, float eps, int dtype, int stream_fp32, void* stream) {
    MOS_REQUIRE(x && gamma && beta && y && (r == nullptr || s_out != nullptr), "mos_add_layernorm_fwd: NULL argument");
    MOS_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * 8 * LN_MAX_CHUNKS, "mos_add_layernorm_fwd: rows=%d C=%d", rows, C);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MOS_F16)
        return stream_fp32 ? add_ln_fwd<f16_t, float>(x, r, gamma, beta, s_out, y, stats, rows, C, eps, st)
                           : add_ln_fwd<f16_t, f16_t>(x, r, gamma, beta, s_out, y, stats, rows, C, eps, st);
    if (dtype == MOS_BF16)
        return stream_fp32 ? add_ln_fwd<bf16_t, float>(x, r, gamma, beta, s_out, y, stats, rows, C, eps, st)
                           : add_ln_fwd<bf16_t, bf16_t>(x, r, gamma, beta, s_out, y, stats, rows, C, eps, st);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_add_layernorm_fwd: dtype %d", dtype);
}

int mos_add_layernorm_bwd(const void* dy, const void* ds, const void* s, const float* gamma, const float* stats, void* dx,
                          void* dx_half, int rows, int C, int dtype, int stream_fp32, void* stream) {
    MOS_REQUIRE(dy && s && gamma && stats && dx, "mos_add_layernorm_bwd: NULL argument");
    MOS_REQUIRE(dx_half == nullptr || stream_fp32, "mos_add_layernorm_bwd: dx_half only with an fp32 residual stream");
    MOS_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * 8 * LN_MAX_CHUNKS, "mos_add_layernorm_bwd: rows=%d C=%d", rows, C);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MOS_F16)
        return stream_fp32 ? add_ln_bwd<f16_t, float>(dy, ds, s, gamma, stats, dx, dx_half, rows, C, st)
                           : add_ln_bwd<f16_t, f16_t>(dy, ds, s, gamma, stats, dx, dx_half, rows, C, st);
    if (dtype == MOS_BF16)
        return stream_fp32 ? add_ln_bwd<bf16_t, float>(dy, ds, s, gamma, stats, dx, dx_half, rows, C, st)
                           : add_ln_bwd<bf16_t, bf16_t>(dy, ds, s, gamma, stats, dx, dx_half, rows, C, st);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_add_layernorm_bwd: dtype %d", dtype);
}

int mos_quick_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream) {
    MOS_REQUIRE(x && y && n > 0 && n % 8 == 0, "mos_quick_gelu_fwd: n=%lld (n %% 8 == 0)", (long long)n);
    return quick_gelu_launch<false>(x, nullptr, y, n, dtype, (hipStream_t)stream);
}

int mos_quick_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype, void* stream) {
    MOS_REQUIRE(dy && x && dx && n > 0 && n % 8 == 0, "mos_quick_gelu_bwd: n=%lld (n %% 8 == 0)", (long long)n);
    return quick_gelu_launch<true>(x, dy, dx, n, dtype, (hipStream_t)stream);
}

int mos_softmax_rows(const void* x, void* y, int rows, int N, float scale, int dtype, void* stream) {
    MOS_REQUIRE(x && y && rows > 0 && N > 0 && N % 8 == 0 && N <= 32768, "mos_softmax_rows: rows=%d N=%d (N %% 8 == 0, N <= 32768)", rows, N);
    if (dtype == MOS_F16) return softmax_rows<f16_t>(x, y, rows, N, scale, (hipStream_t)stream);
    if (dtype == MOS_BF16) return softmax_rows<bf16_t>(x, y, rows, N, scale, (hipStream_t)stream);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_softmax_rows: dtype %d", dtype);
}

int mos_geglu_fwd(const void* h, void* y, int64_t rows, int F, int dtype, void* stream) {
    MOS_REQUIRE(h && y && rows > 0 && F > 0 && F % 8 == 0, "mos_geglu_fwd: rows=%lld F=%d", (long long)rows, F);
    hipStream_t st = (hipStream_t)stream;
    const int64_t nvec = rows * (F / 8);
    const int blocks = (int)((nvec + 255) / 256 > 16384 ? 16384 : (nvec + 255) / 256);
    char key[64];
    snprintf(key, sizeof(key), "rows%lld F%d", (long long)rows, F);
    MosProfScope prof(st, "geglu_fwd", key, 20.0 * rows * F, 6.0 * rows * (double)F);
    if (dtype == MOS_F16) hipLaunchKernelGGL((geglu_fwd_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, (const f16_t*)h, (f16_t*)y, nvec, F);
    else if (dtype == MOS_BF16) hipLaunchKernelGGL((geglu_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)h, (bf16_t*)y, nvec, F);
    else return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_geglu_fwd: dtype %d", dtype);
    return mos_check_launch("geglu_fwd");
}

int mos_geglu_bwd(const void* dy, const void* h, void* dh, int64_t rows, int F, int dtype, void* stream) {
    MOS_REQUIRE(dy && h && dh && rows > 0 && F > 0 && F % 8 == 0, "mos_geglu_bwd: rows=%lld F=%d", (long long)rows, F);
    hipStream_t st = (hipStream_t)stream;
    const int64_t nvec = rows * (F / 8);
    const int blocks = (int)((nvec + 255) / 256 > 16384 ? 16384 : (nvec + 255) / 256);
    char key[64];
    snprintf(key, sizeof(key), "rows%lld F%d", (long long)rows, F);
    MosProfScope prof(st, "geglu_bwd", key, 40.0 * rows * F, 10.0 * rows * (double)F);
    if (dtype == MOS_F16) hipLaunchKernelGGL((geglu_bwd_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, (const f16_t*)dy, (const f16_t*)h, (f16_t*)dh, nvec, F);
    else if (dtype == MOS_BF16) hipLaunchKernelGGL((geglu_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)h, (bf16_t*)dh, nvec, F);
    else return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_geglu_bwd: dtype %d", dtype);
    return mos_check_launch("geglu_bwd");
}

}  // extern "C"

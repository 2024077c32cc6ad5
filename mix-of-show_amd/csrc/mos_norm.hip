// mos_norm.hip — fused GroupNorm (+ SiLU) forward / backward on half-precision NCHW and NHWC activations (gfx950).
//
// Not one of the attention-path kernels of SURVEY.md §8(a); it is the first item of §8(f) "next": the ResnetBlock2D /
// Transformer2DModel / VAE GroupNorm->SiLU pairs that CALL the attention path. Under autocast the reference stack runs
// GroupNorm in fp32 (cast in, fp32 normalise, fp32 SiLU, cast out at the next conv): 4-6 elementwise passes over
// activations as large as (4,128,512,512). Here: statistics in fp32/fp64 from the half tensor, one fused
// normalise+affine+SiLU pass writing half — 2 reads + 1 write of the half tensor. HBM-bound by construction.
// gamma/beta are frozen in ED-LoRA training (no affine gradients are produced).
#include <cstdio>
#include <type_traits>
#include <cstdlib>
#include "mos_common.h"

#ifndef MOS_GN_PRE_MAX_STEPS
#define MOS_GN_PRE_MAX_STEPS 32   // prologue steps per wave up to which GroupNorm-from-producer-statistics is ONE launch (see gn_nhwc_run_pre)
#endif

namespace {

constexpr int GN_SLICE = 16384;   // elements per workgroup slice (256 threads x 8 elements x 8 iterations)
constexpr int GN_MAX_SPLIT = 64;

struct GnArgs {
    const void* x; const void* dy; void* out;      // fwd: x -> out ; bwd: (x, dy) -> out (= dx)
    const float* gamma; const float* beta;
    float* stats;                                  // [B*G][2] = mean, rstd
    float* partial;                                // [B*G][nsplit][2]
    int B, C, HW, G, cpg, nsplit;
    int64_t group_elems;
    float eps;
};

__device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }

__device__ __forceinline__ void block_reduce2(float& a, float& b, float* red /*[8]*/) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave * 2] = a; red[wave * 2 + 1] = b; }
    __syncthreads();
    a = red[0] + red[2] + red[4] + red[6];
    b = red[1] + red[3] + red[5] + red[7];
}

// slice [e0, e1) of group bg, in units of elements; every slice boundary is a multiple of 8
__device__ __forceinline__ void slice_range(const GnArgs& a, int64_t& e0, int64_t& e1) {
    const int64_t per = ((a.group_elems / 8 + a.nsplit - 1) / a.nsplit) * 8;
    e0 = (int64_t)blockIdx.y * per;
    e1 = min(e0 + per, a.group_elems);
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(GnArgs a) {
    typedef typename MT<T>::v8 v8;
    __shared__ float red[8];
    const int bg = blockIdx.x;
    int64_t e0, e1;
    slice_range(a, e0, e1);
    const T* xg = (const T*)a.x + (int64_t)bg * a.group_elems;
    float s = 0.f, ss = 0.f;
    for (int64_t e = e0 + (int64_t)threadIdx.x * 8; e < e1; e += 256 * 8) {
        const v8 v = as_v8<T>(ld16(xg + e));
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float f = (float)v[i]; s += f; ss += f * f; }
    }
    block_reduce2(s, ss, red);
    if (threadIdx.x == 0) {
        float* p = a.partial + ((int64_t)bg * a.nsplit + blockIdx.y) * 2;
        p[0] = s; p[1] = ss;
    }
}

// mean / rstd of group bg from the slice partials (combined in double: E[x^2]-mean^2 is cancellation-prone in fp32)
__device__ __forceinline__ void group_stats(const GnArgs& a, int bg, float& mean, float& rstd) {
    double s = 0.0, ss = 0.0;
    for (int i = 0; i < a.nsplit; ++i) {
        s += (double)a.partial[((int64_t)bg * a.nsplit + i) * 2];
        ss += (double)a.partial[((int64_t)bg * a.nsplit + i) * 2 + 1];
    }
    const double n = (double)a.group_elems;
    const double m = s / n;
    double var = ss / n - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)a.eps));
}

template <typename T, bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnArgs a) {
    typedef typename MT<T>::v8 v8;
    const int bg = blockIdx.x;
    const int g = bg % a.G;
    float mean, rstd;
    group_stats(a, bg, mean, rstd);
    if (blockIdx.y == 0 && threadIdx.x == 0 && a.stats != nullptr) { a.stats[bg * 2] = mean; a.stats[bg * 2 + 1] = rstd; }
    int64_t e0, e1;
    slice_range(a, e0, e1);
    const T* xg = (const T*)a.x + (int64_t)bg * a.group_elems;
    T* yg = (T*)a.out + (int64_t)bg * a.group_elems;
    for (int64_t e = e0 + (int64_t)threadIdx.x * 8; e < e1; e += 256 * 8) {
        const int c = g * a.cpg + (int)(e / a.HW);        // HW % 8 == 0: the 8 elements share one channel
        const float ga = a.gamma[c] * rstd, be = a.beta[c] - mean * a.gamma[c] * rstd;
        const v8 v = as_v8<T>(ld16(xg + e));
        v8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float z = (float)v[i] * ga + be;
            if (SILU) z = silu_f(z);
            o[i] = (T)z;
        }
        st16(yg + e, from_v8<T>(o));
    }
}

// backward pass 1: per group sum(g) and sum(g * xhat), g = dL/dz * gamma, z = xhat*gamma + beta
template <typename T, bool SILU>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(GnArgs a) {
    typedef typename MT<T>::v8 v8;
    __shared__ float red[8];
    const int bg = blockIdx.x;
    const int g = bg % a.G;
    const float mean = a.stats[bg * 2], rstd = a.stats[bg * 2 + 1];
    int64_t e0, e1;
    slice_range(a, e0, e1);
    const T* xg = (const T*)a.x + (int64_t)bg * a.group_elems;
    const T* dg = (const T*)a.dy + (int64_t)bg * a.group_elems;
    float sg = 0.f, sgx = 0.f;
    for (int64_t e = e0 + (int64_t)threadIdx.x * 8; e < e1; e += 256 * 8) {
        const int c = g * a.cpg + (int)(e / a.HW);
        const float gam = a.gamma[c], bet = a.beta[c];
        const v8 v = as_v8<T>(ld16(xg + e)), d = as_v8<T>(ld16(dg + e));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xh = ((float)v[i] - mean) * rstd;
            float dz = (float)d[i];
            if (SILU) {
                const float z = xh * gam + bet;
                const float sig = 1.f / (1.f + __expf(-z));
                dz *= sig * (1.f + z * (1.f - sig));
            }
            const float gg = dz * gam;
            sg += gg; sgx += gg * xh;
        }
    }
    block_reduce2(sg, sgx, red);
    if (threadIdx.x == 0) {
        float* p = a.partial + ((int64_t)bg * a.nsplit + blockIdx.y) * 2;
        p[0] = sg; p[1] = sgx;
    }
}

// backward pass 2: dx = rstd * (g - mean(g) - xhat * mean(g*xhat))
template <typename T, bool SILU>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GnArgs a) {
    typedef typename MT<T>::v8 v8;
    const int bg = blockIdx.x;
    const int g = bg % a.G;
    const float mean = a.stats[bg * 2], rstd = a.stats[bg * 2 + 1];
    double s0 = 0.0, s1 = 0.0;
    for (int i = 0; i < a.nsplit; ++i) {
        s0 += (double)a.partial[((int64_t)bg * a.nsplit + i) * 2];
        s1 += (double)a.partial[((int64_t)bg * a.nsplit + i) * 2 + 1];
    }
    const float mg = (float)(s0 / (double)a.group_elems), mgx = (float)(s1 / (double)a.group_elems);
    int64_t e0, e1;
    slice_range(a, e0, e1);
    const T* xg = (const T*)a.x + (int64_t)bg * a.group_elems;
    const T* dg = (const T*)a.dy + (int64_t)bg * a.group_elems;
    T* og = (T*)a.out + (int64_t)bg * a.group_elems;
    for (int64_t e = e0 + (int64_t)threadIdx.x * 8; e < e1; e += 256 * 8) {
        const int c = g * a.cpg + (int)(e / a.HW);
        const float gam = a.gamma[c], bet = a.beta[c];
        const v8 v = as_v8<T>(ld16(xg + e)), d = as_v8<T>(ld16(dg + e));
        v8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xh = ((float)v[i] - mean) * rstd;
            float dz = (float)d[i];
            if (SILU) {
                const float z = xh * gam + bet;
                const float sig = 1.f / (1.f + __expf(-z));
                dz *= sig * (1.f + z * (1.f - sig));
            }
            o[i] = (T)(rstd * (dz * gam - mg - xh * mgx));
        }
        st16(og + e, from_v8<T>(o));
    }
}

int gn_nsplit(int64_t group_elems) {
    int64_t n = (group_elems + GN_SLICE - 1) / GN_SLICE;
    if (n < 1) n = 1;
    if (n > GN_MAX_SPLIT) n = GN_MAX_SPLIT;
    return (int)n;
}

int gn_check(const void* x, const void* out, const float* gamma, const float* beta, void* ws, int B, int C, int HW, int G,
             const char* who) {
    if (!x || !out || !gamma || !beta || !ws) return mos_set_error(MOS_ERR_BAD_ARG, "%s: NULL argument", who);
    if (B <= 0 || C <= 0 || HW <= 0 || G <= 0 || C % G != 0 || HW % 8 != 0)
        return mos_set_error(MOS_ERR_BAD_ARG, "%s: B=%d C=%d HW=%d G=%d (need C %% G == 0, HW %% 8 == 0)", who, B, C, HW, G);
    return MOS_OK;
}

GnArgs gn_args(const void* x, const void* dy, void* out, const float* gamma, const float* beta, float* stats, void* ws,
               int B, int C, int HW, int G, float eps) {
    GnArgs a;
    a.x = x; a.dy = dy; a.out = out; a.gamma = gamma; a.beta = beta; a.stats = stats; a.partial = (float*)ws;
    a.B = B; a.C = C; a.HW = HW; a.G = G; a.cpg = C / G; a.group_elems = (int64_t)(C / G) * HW;
    a.nsplit = gn_nsplit(a.group_elems); a.eps = eps;
    return a;
}

template <typename T>
int gn_fwd(GnArgs a, int silu, hipStream_t st) {
    const dim3 grid(a.B * a.G, a.nsplit);
    char key[96];
    snprintf(key, sizeof(key), "B%d C%d HW%d%s", a.B, a.C, a.HW, silu ? " +silu" : "");
    const double n = (double)a.B * a.C * a.HW;
    {
        MosProfScope prof(st, "groupnorm_stats", key, 3.0 * n, 2.0 * n);
        hipLaunchKernelGGL((gn_stats_kernel<T>), grid, dim3(256), 0, st, a);
    }
    int rc = mos_check_launch("gn_stats");
    if (rc) return rc;
    MosProfScope prof(st, "groupnorm_apply", key, 8.0 * n, 4.0 * n);
    if (silu) hipLaunchKernelGGL((gn_apply_kernel<T, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gn_apply_kernel<T, false>), grid, dim3(256), 0, st, a);
    return mos_check_launch("gn_apply");
}

template <typename T>
int gn_bwd(GnArgs a, int silu, hipStream_t st) {
    const dim3 grid(a.B * a.G, a.nsplit);
    char key[96];
    snprintf(key, sizeof(key), "B%d C%d HW%d%s", a.B, a.C, a.HW, silu ? " +silu" : "");
    const double n = (double)a.B * a.C * a.HW;
    {
        MosProfScope prof(st, "groupnorm_bwd_stats", key, 12.0 * n, 4.0 * n);
        if (silu) hipLaunchKernelGGL((gn_bwd_stats_kernel<T, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gn_bwd_stats_kernel<T, false>), grid, dim3(256), 0, st, a);
    }
    int rc = mos_check_launch("gn_bwd_stats");
    if (rc) return rc;
    MosProfScope prof(st, "groupnorm_bwd_apply", key, 14.0 * n, 6.0 * n);
    if (silu) hipLaunchKernelGGL((gn_bwd_apply_kernel<T, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gn_bwd_apply_kernel<T, false>), grid, dim3(256), 0, st, a);
    return mos_check_launch("gn_bwd_apply");
}


// ---- channels-last (NHWC) variants ----------------------------------------------------------------------------------
// x[b][p][c], c contiguous: the layout of the token-major attention path and of MIOpen's fp16 implicit-GEMM convolutions
// (which otherwise transpose NCHW <-> NHWC around every call). A thread owns a FIXED set of 8-channel vectors (so the
// per-channel affine / statistics constants live in registers) and strides over pixels; per-channel partial sums are
// folded to per-group sums through LDS. grid (B, nsplit pixel slices); second-stage combine in double, as above.
struct GnNhwcArgs {
    const void* x; const void* dy; void* out;
    const float* gamma; const float* beta;
    float* stats;                                  // [B*G][2] = mean, rstd
    float* partial;                                // [B][nsplit][G][2]
    int B, C, HW, G, cpg, nsplit, V, TP, R, ppb;   // V = C/8 vectors per pixel, TP threads per pixel, R pixel rows per block
    float eps;
    const void* ds;                                // backward only: gradient that bypasses the norm (added to dx), or NULL
    int64_t ds_ps;                                 // its PIXEL stride in elements (C = dense; > C: a channel slice of a wider NHWC
                                                   // tensor -- the gradient of one input of a concatenation, read in place: round 6)
};

template <typename T, int VT, bool BWD, bool SILU>
__global__ __launch_bounds__(256) void gn_nhwc_reduce_kernel(GnNhwcArgs a) {
    typedef typename MT<T>::v8 v8;
    extern __shared__ float gn_lds[];               // [R][C][2] per-(pixel row of the block, channel) sums
    float* red = gn_lds;
    __shared__ float mean_s[64], rstd_s[64];
    const int tid = threadIdx.x;
    const int r = tid / a.TP, tp = tid - r * a.TP;
    const int b = blockIdx.x, sp = blockIdx.y;
    const int p0 = sp * a.ppb, p1 = min(p0 + a.ppb, a.HW);
    if (BWD && tid < a.G) { mean_s[tid] = a.stats[(b * a.G + tid) * 2]; rstd_s[tid] = a.stats[(b * a.G + tid) * 2 + 1]; }
    __syncthreads();
    float s0[VT][8], s1[VT][8];
    float cm[VT][8], cr[VT][8], cg[VT][8], cb[VT][8];
#pragma unroll
    for (int k = 0; k < VT; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s0[k][i] = 0.f; s1[k][i] = 0.f;
            if (BWD) {
                const int c = min((tp + a.TP * k) * 8 + i, a.C - 1);
                const int g = c / a.cpg;
                cm[k][i] = mean_s[g]; cr[k][i] = rstd_s[g]; cg[k][i] = a.gamma[c]; cb[k][i] = a.beta[c];
            }
        }
    const T* xb = (const T*)a.x + (int64_t)b * a.HW * a.C;
    const T* db = BWD ? (const T*)a.dy + (int64_t)b * a.HW * a.C : nullptr;
    constexpr int U = 4;                         // pixel rows in flight per thread (independent 16 B loads)
    for (int p = p0 + r; p < p1; p += U * a.R) {
        u32x4 xr[U][VT], dr[U][VT];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < VT; ++k) {
                const int v = min(tp + a.TP * k, a.V - 1);
                const int64_t off = (int64_t)min(p + u * a.R, p1 - 1) * a.C + v * 8;
                xr[u][k] = ld16(xb + off);
                if (BWD) dr[u][k] = ld16(db + off);
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p + u * a.R >= p1) continue;
#pragma unroll
            for (int k = 0; k < VT; ++k) {
                if (tp + a.TP * k >= a.V) continue;
                const v8 xv = as_v8<T>(xr[u][k]);
                if (!BWD) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { const float f = (float)xv[i]; s0[k][i] += f; s1[k][i] += f * f; }
                } else {
                    const v8 dv = as_v8<T>(dr[u][k]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float xh = ((float)xv[i] - cm[k][i]) * cr[k][i];
                        float dz = (float)dv[i];
                        if (SILU) {
                            const float z = xh * cg[k][i] + cb[k][i];
                            const float sig = 1.f / (1.f + __expf(-z));
                            dz *= sig * (1.f + z * (1.f - sig));
                        }
                        const float gg = dz * cg[k][i];
                        s0[k][i] += gg; s1[k][i] += gg * xh;
                    }
                }
            }
        }
    }
    // Block sums in a FIXED order (round 6; LDS float atomics made the statistics of this form differ in the last bit from run to
    // run): every pixel row r of the block leaves its per-channel sums in its own LDS row, the rows are added in order
    const int C2 = 2 * a.C;
#pragma unroll
    for (int k = 0; k < VT; ++k) {
        const int v = tp + a.TP * k;
        if (v < a.V) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                red[r * C2 + (v * 8 + i) * 2] = s0[k][i];
                red[r * C2 + (v * 8 + i) * 2 + 1] = s1[k][i];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < C2; i += blockDim.x) {
        float t = red[i];
        for (int q = 1; q < a.R; ++q) t += red[q * C2 + i];
        red[i] = t;
    }
    __syncthreads();
    if (tid < a.G) {
        float t0 = 0.f, t1 = 0.f;
        for (int c = tid * a.cpg; c < (tid + 1) * a.cpg; ++c) { t0 += red[c * 2]; t1 += red[c * 2 + 1]; }
        float* pp = a.partial + (((int64_t)b * a.nsplit + sp) * a.G + tid) * 2;
        pp[0] = t0; pp[1] = t1;
    }
}

// DS (backward): out = half(dx) + ds -- the gradient of a residual connection that bypasses this norm, added where autograd
// would otherwise launch a separate accumulation kernel (same rounding points: dx rounded to T, then the half add)
// The per-group constants were combined once per image by gn_nhwc_finalize_kernel (a.partial + fin_off). (Rounds 1-2 let every
// apply block combine the nsplit slice partials itself: at the UNet's sizes that prologue -- up to 128 x G x 2 partials per block
// -- moved more L2 traffic per block than the block's own slice of the activation; removed in round 5.)
template <typename T, int VT, bool BWD, bool SILU, bool DS = false>
__global__ __launch_bounds__(256) void gn_nhwc_apply_kernel(GnNhwcArgs a) {
    static_assert(BWD || !DS, "the bypass gradient exists in backward only");
    typedef typename MT<T>::v8 v8;
    __shared__ float u_s[64], w_s[64], m_s[64], r_s[64];   // fwd: mean, rstd ; bwd: mean(g), mean(g xhat), mean, rstd
    const int tid = threadIdx.x;
    const int r = tid / a.TP, tp = tid - r * a.TP;
    const int b = blockIdx.x, sp = blockIdx.y;
    const int p0 = sp * a.ppb, p1 = min(p0 + a.ppb, a.HW);
    if (tid < a.G) {
        const float* fin = a.partial + (int64_t)a.B * a.nsplit * a.G * 2 + ((int64_t)b * a.G + tid) * 2;
        u_s[tid] = fin[0]; w_s[tid] = fin[1];
        if (BWD) { m_s[tid] = a.stats[(b * a.G + tid) * 2]; r_s[tid] = a.stats[(b * a.G + tid) * 2 + 1]; }
    }
    __syncthreads();
    // per-channel constants of this thread's vectors
    float c0[VT][8], c1[VT][8], c2[VT][8], c3[VT][8], c4[VT][8], c5[VT][8];
#pragma unroll
    for (int k = 0; k < VT; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = min((tp + a.TP * k) * 8 + i, a.C - 1);
            const int g = c / a.cpg;
            const float gam = a.gamma[c], bet = a.beta[c];
            if (!BWD) {
                c0[k][i] = gam * w_s[g];                       // y = c0 * x + c1
                c1[k][i] = bet - u_s[g] * gam * w_s[g];
            } else {
                c0[k][i] = m_s[g]; c1[k][i] = r_s[g]; c2[k][i] = gam; c3[k][i] = bet; c4[k][i] = u_s[g]; c5[k][i] = w_s[g];
            }
        }
    const T* xb = (const T*)a.x + (int64_t)b * a.HW * a.C;
    const T* db = BWD ? (const T*)a.dy + (int64_t)b * a.HW * a.C : nullptr;
    T* ob = (T*)a.out + (int64_t)b * a.HW * a.C;
    [[maybe_unused]] const T* sb = DS ? (const T*)a.ds + (int64_t)b * a.HW * a.ds_ps : nullptr;
    constexpr int U = 4;
    for (int p = p0 + r; p < p1; p += U * a.R) {
        u32x4 xr[U][VT], dr[U][VT];
        [[maybe_unused]] u32x4 sr[U][VT];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < VT; ++k) {
                const int v = min(tp + a.TP * k, a.V - 1);
                const int64_t off = (int64_t)min(p + u * a.R, p1 - 1) * a.C + v * 8;
                xr[u][k] = ld16(xb + off);
                if (BWD) dr[u][k] = ld16(db + off);
                if constexpr (DS) sr[u][k] = ld16(sb + (int64_t)min(p + u * a.R, p1 - 1) * a.ds_ps + v * 8);
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p + u * a.R >= p1) continue;
#pragma unroll
            for (int k = 0; k < VT; ++k) {
                const int v = tp + a.TP * k;
                if (v >= a.V) continue;
                const int64_t off = (int64_t)(p + u * a.R) * a.C + v * 8;
                const v8 xv = as_v8<T>(xr[u][k]);
                v8 o;
                if (!BWD) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float z = (float)xv[i] * c0[k][i] + c1[k][i];
                        if (SILU) z = silu_f(z);
                        o[i] = (T)z;
                    }
                } else {
                    const v8 dv = as_v8<T>(dr[u][k]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float xh = ((float)xv[i] - c0[k][i]) * c1[k][i];
                        float dz = (float)dv[i];
                        if (SILU) {
                            const float z = xh * c2[k][i] + c3[k][i];
                            const float sig = 1.f / (1.f + __expf(-z));
                            dz *= sig * (1.f + z * (1.f - sig));
                        }
                        o[i] = (T)(c1[k][i] * (dz * c2[k][i] - c4[k][i] - xh * c5[k][i]));
                    }
                    if constexpr (DS) {
                        const v8 sv = as_v8<T>(sr[u][k]);
#pragma unroll
                        for (int i = 0; i < 8; ++i) o[i] = (T)((float)o[i] + (float)sv[i]);
                    }
                }
                st16(ob + off, from_v8<T>(o));
            }
        }
    }
}

// One block per image: the nsplit slice partials of every group -> the per-group constants the apply kernel needs (forward: mean,
// rstd, also written to `stats`; backward: mean(g), mean(g xhat)). Each of the 2G (group, moment) columns is summed by two threads
// (even / odd slices, independent coalesced loads, double accumulation, fixed order: deterministic).
template <bool BWD>
__global__ __launch_bounds__(256) void gn_nhwc_finalize_kernel(GnNhwcArgs a) {
    __shared__ double part[2][128];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int e = tid & 127, q = tid >> 7, n2 = 2 * a.G;
    double acc = 0.0;
    if (e < n2) {
        const float* pb = a.partial + (int64_t)b * a.nsplit * n2 + e;
        int s = q;
        for (; s + 14 < a.nsplit; s += 16) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = pb[(int64_t)(s + 2 * j) * n2];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (double)v[j];
        }
        for (; s < a.nsplit; s += 2) acc += (double)pb[(int64_t)s * n2];
    }
    part[q][e] = acc;
    __syncthreads();
    if (tid < a.G) {
        const double t0 = part[0][tid * 2] + part[1][tid * 2], t1 = part[0][tid * 2 + 1] + part[1][tid * 2 + 1];
        const double n = (double)a.cpg * a.HW;
        float* fin = a.partial + (int64_t)a.B * a.nsplit * n2 + ((int64_t)b * a.G + tid) * 2;
        if (!BWD) {
            const double m = t0 / n;
            double var = t1 / n - m * m;
            if (var < 0.0) var = 0.0;
            const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)a.eps));
            fin[0] = mean; fin[1] = rstd;
            if (a.stats != nullptr) { a.stats[(b * a.G + tid) * 2] = mean; a.stats[(b * a.G + tid) * 2 + 1] = rstd; }
        } else {
            fin[0] = (float)(t0 / n); fin[1] = (float)(t1 / n);
        }
    }
}

bool gn_nhwc_plan(GnNhwcArgs& a) {
    a.V = a.C / 8;
    const int vt = (a.V + 255) / 256;
    if (vt > 2 || a.G > 64) return false;
    a.TP = (a.V + vt - 1) / vt;
    a.R = 256 / a.TP < 1 ? 1 : 256 / a.TP;
    int ns = (1024 + a.B - 1) / a.B;                 // aim at ~1024 workgroups, at most 128 slices per image
    const int maxs = (a.HW + 4 * a.R - 1) / (4 * a.R);   // at least one unrolled group (4 pixel rows) per thread
    if (ns > maxs) ns = maxs;
    if (ns > 128) ns = 128;
    if (ns < 1) ns = 1;
    a.ppb = (a.HW + ns - 1) / ns;
    a.nsplit = (a.HW + a.ppb - 1) / a.ppb;
    return true;
}

template <typename T, bool BWD>
int gn_nhwc_run(GnNhwcArgs a, int silu, hipStream_t st) {
    const int vt = (a.V + 255) / 256;
    const dim3 grid(a.B, a.nsplit), block(a.TP * a.R);
    const size_t lds = (size_t)2 * a.C * a.R * sizeof(float);      // <= 32 KB: R * C <= 256 threads x 8 channels x vt
    char key[96];
    snprintf(key, sizeof(key), "nhwc B%d C%d HW%d%s", a.B, a.C, a.HW, silu ? " +silu" : "");
    const double n = (double)a.B * a.C * a.HW;
    {
        MosProfScope prof(st, BWD ? "groupnorm_bwd_stats" : "groupnorm_stats", key, (BWD ? 12.0 : 3.0) * n, (BWD ? 4.0 : 2.0) * n);
        if (vt == 1) {
            if (silu) hipLaunchKernelGGL((gn_nhwc_reduce_kernel<T, 1, BWD, true>), grid, block, lds, st, a);
            else hipLaunchKernelGGL((gn_nhwc_reduce_kernel<T, 1, BWD, false>), grid, block, lds, st, a);
        } else {
            if (silu) hipLaunchKernelGGL((gn_nhwc_reduce_kernel<T, 2, BWD, true>), grid, block, lds, st, a);
            else hipLaunchKernelGGL((gn_nhwc_reduce_kernel<T, 2, BWD, false>), grid, block, lds, st, a);
        }
    }
    int rc = mos_check_launch("gn_nhwc_reduce");
    if (rc) return rc;
    {
        MosProfScope prof(st, BWD ? "groupnorm_bwd_finalize" : "groupnorm_finalize", key, 2.0 * a.B * a.nsplit * a.G,
                          8.0 * a.B * a.nsplit * a.G);
        hipLaunchKernelGGL((gn_nhwc_finalize_kernel<BWD>), dim3(a.B), dim3(256), 0, st, a);
    }
    rc = mos_check_launch("gn_nhwc_finalize");
    if (rc) return rc;
    MosProfScope prof(st, BWD ? "groupnorm_bwd_apply" : "groupnorm_apply", key, (BWD ? 14.0 : 8.0) * n, (BWD ? 6.0 : 4.0) * n);
#define GN_APPLY(VTN, S, D) hipLaunchKernelGGL((gn_nhwc_apply_kernel<T, VTN, BWD, S, D>), grid, block, 0, st, a)
    if constexpr (BWD) {
        if (a.ds != nullptr) {
            if (vt == 1) { if (silu) GN_APPLY(1, true, true); else GN_APPLY(1, false, true); }
            else { if (silu) GN_APPLY(2, true, true); else GN_APPLY(2, false, true); }
            return mos_check_launch("gn_nhwc_apply");
        }
    }
    if (vt == 1) { if (silu) GN_APPLY(1, true, false); else GN_APPLY(1, false, false); }
    else { if (silu) GN_APPLY(2, true, false); else GN_APPLY(2, false, false); }
#undef GN_APPLY
    return mos_check_launch("gn_nhwc_apply");
}

// Round 6: statistics that arrive WITH the map -- the producing convolution's epilogue left, per (pixel tile, channel), the sum and
// the sum of squares of the stored values (conv3x3_halo_kernel, mos_conv.hip). One 256-thread block per (image, group) adds its
// cpg channels over all tiles in a fixed order, in double, and writes the per-group constants where gn_nhwc_apply_kernel reads them
// (and `stats` for the backward): the statistics pass over the activation and its finalize launch are replaced by this one.
__global__ __launch_bounds__(256) void gn_chan_finalize_kernel(GnNhwcArgs a, const float* __restrict__ chan_part, int tiles) {
    __shared__ double red[4][2];
    const int b = blockIdx.x / a.G, g = blockIdx.x - b * a.G, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = tiles * a.cpg;
    double t0 = 0.0, t1 = 0.0;
    int i = tid;
    for (; i + 768 < n; i += 1024) {            // four independent loads in flight per thread (the VAE's maps: n = 4096 .. 12288)
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = i + 256 * u, tile = e / a.cpg, c = g * a.cpg + (e - tile * a.cpg);
            v[u] = *reinterpret_cast<const float2*>(chan_part + (((int64_t)b * tiles + tile) * a.C + c) * 2);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { t0 += (double)v[u].x; t1 += (double)v[u].y; }
    }
    for (; i < n; i += 256) {
        const int tile = i / a.cpg, c = g * a.cpg + (i - tile * a.cpg);
        const float2 v = *reinterpret_cast<const float2*>(chan_part + (((int64_t)b * tiles + tile) * a.C + c) * 2);
        t0 += (double)v.x; t1 += (double)v.y;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { t0 += __shfl_xor(t0, o); t1 += __shfl_xor(t1, o); }
    if (lane == 0) { red[wave][0] = t0; red[wave][1] = t1; }
    __syncthreads();
    if (tid == 0) {
        t0 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        t1 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        const double cnt = (double)a.cpg * a.HW;
        const double m = t0 / cnt;
        double var = t1 / cnt - m * m;
        if (var < 0.0) var = 0.0;
        const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)a.eps));
        float* fin = a.partial + (int64_t)a.B * a.nsplit * a.G * 2 + ((int64_t)b * a.G + g) * 2;
        fin[0] = mean; fin[1] = rstd;
        if (a.stats != nullptr) { a.stats[(b * a.G + g) * 2] = mean; a.stats[(b * a.G + g) * 2 + 1] = rstd; }
    }
}

// Round 6, second step: GroupNorm(+SiLU) from the producer's statistics in ONE launch. A workgroup owns a channel range of whole
// groups and whole 16-byte vectors (>= 64 channels where the map has them: 128-byte runs per pixel) and a pixel slice; its
// prologue adds the producer's per-(tile, channel) sums of ITS groups (a few KB: tiles x range x 2 floats, fixed order, double) --
// every workgroup of a range repeats that small sum instead of waiting for a finalize launch -- then streams its slice once:
// y = c0 x + c1 (+ SiLU), the arithmetic of gn_nhwc_apply_kernel. Full-width grid at every map size, one read, one write: replaces
// finalize + apply on the large maps and (with the statistics present) the 32-64-workgroup column kernel on the middle levels.
constexpr int GN_PRE_MAXG = 16;
struct GnPreArgs {
    const void* x; void* out; const float* gamma; const float* beta; float* stats; const float* chan_part;
    const float* fin;               // NULL, or [B][G][2] mean / rstd already combined by gn_chan_finalize_kernel: no prologue
    int B, C, HW, G, cpg, tiles;
    int cw, ngw, NV, RP, ppb;       // channels / groups / 16-byte vectors per workgroup range, pixels in flight, pixels per slice
    float eps;
};

template <typename T, bool SILU>
__global__ __launch_bounds__(256) void gn_pre_apply_kernel(GnPreArgs a) {
    typedef typename MT<T>::v8 v8;
    __shared__ float mean_s[GN_PRE_MAXG], rstd_s[GN_PRE_MAXG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int unit = blockIdx.x, sp = blockIdx.y, b = blockIdx.z;
    const int c_lo = unit * a.cw, g_lo = c_lo / a.cpg;
    if (a.fin != nullptr) {                             // two-launch form: the constants are there, the workgroup only streams
        if (tid < a.ngw) {
            mean_s[tid] = a.fin[((int64_t)b * a.G + g_lo + tid) * 2];
            rstd_s[tid] = a.fin[((int64_t)b * a.G + g_lo + tid) * 2 + 1];
        }
    } else
    for (int gi = wave; gi < a.ngw; gi += 4) {          // one wave per group of the range
        const int n = a.tiles * a.cpg;
        double t0 = 0.0, t1 = 0.0;
        for (int i = lane; i < n; i += 64) {
            const int tile = i / a.cpg, c = (g_lo + gi) * a.cpg + (i - tile * a.cpg);
            const float2 v = *reinterpret_cast<const float2*>(a.chan_part + (((int64_t)b * a.tiles + tile) * a.C + c) * 2);
            t0 += (double)v.x; t1 += (double)v.y;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { t0 += __shfl_xor(t0, o); t1 += __shfl_xor(t1, o); }
        if (lane == 0) {
            const double cnt = (double)a.cpg * a.HW, m = t0 / cnt;
            double var = t1 / cnt - m * m;
            if (var < 0.0) var = 0.0;
            const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)a.eps));
            mean_s[gi] = mean; rstd_s[gi] = rstd;
            if (sp == 0 && a.stats != nullptr) {
                a.stats[(b * a.G + g_lo + gi) * 2] = mean; a.stats[(b * a.G + g_lo + gi) * 2 + 1] = rstd;
            }
        }
    }
    __syncthreads();
    const int r = tid / a.NV, v = tid - r * a.NV;       // pixel lane, vector of the range
    if (r >= a.RP) return;
    float c0[8], c1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = c_lo + v * 8 + i, gi = c / a.cpg - g_lo;
        const float gam = a.gamma[c], bet = a.beta[c];
        c0[i] = gam * rstd_s[gi];
        c1[i] = bet - mean_s[gi] * gam * rstd_s[gi];
    }
    const int p0 = sp * a.ppb, p1 = min(p0 + a.ppb, a.HW);
    const T* xb = (const T*)a.x + (int64_t)b * a.HW * a.C + c_lo + v * 8;
    T* ob = (T*)a.out + (int64_t)b * a.HW * a.C + c_lo + v * 8;
    constexpr int U = 4;
    for (int p = p0 + r; p < p1; p += U * a.RP) {
        u32x4 xr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xr[u] = ld16(xb + (int64_t)min(p + u * a.RP, p1 - 1) * a.C);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p + u * a.RP >= p1) continue;
            const v8 xv = as_v8<T>(xr[u]);
            v8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float z = (float)xv[i] * c0[i] + c1[i];
                if (SILU) z = silu_f(z);
                o[i] = (T)z;
            }
            st16(ob + (int64_t)(p + u * a.RP) * a.C, from_v8<T>(o));
        }
    }
}

int gn_gcd(int x, int y);
// channel range of a workgroup: whole groups, whole vectors, >= 64 channels where possible, <= GN_PRE_MAXG groups, <= 256 threads
bool gn_pre_plan(const GnNhwcArgs& a, GnPreArgs& q, const float* chan_part, int tiles) {
    const int cpg = a.cpg;
    int cu = cpg / gn_gcd(cpg, 8) * 8;                  // lcm(cpg, 8)
    if (a.C % cu != 0) return false;
    int cw = cu;
    while (cw < 64 && a.C % (cw * 2) == 0) cw *= 2;     // (cw stays a multiple of cu that divides C)
    if (cw / cpg > GN_PRE_MAXG || cw / 8 > 256) return false;
    q.x = a.x; q.out = a.out; q.gamma = a.gamma; q.beta = a.beta; q.stats = a.stats; q.chan_part = chan_part;
    q.B = a.B; q.C = a.C; q.HW = a.HW; q.G = a.G; q.cpg = cpg; q.tiles = tiles; q.eps = a.eps;
    q.cw = cw; q.ngw = cw / cpg; q.NV = cw / 8; q.RP = 256 / q.NV;
    const int units = a.C / cw;
    int ns = (1024 + a.B * units - 1) / (a.B * units);  // ~1024 workgroups
    const int maxs = (a.HW + 4 * q.RP - 1) / (4 * q.RP);
    if (ns > maxs) ns = maxs;
    if (ns < 1) ns = 1;
    q.ppb = (a.HW + ns - 1) / ns;
    return true;
}

// The one-launch form's prologue is repeated by EVERY workgroup of a channel range: tiles x cpg partial pairs per group, a wave per
// group, 64 pairs per step. On the UNet's maps that is 2 .. 16 steps per wave; on the VAE's (1024 tiles of 16 x 16 pixels per 512 x
// 512 image) it was 256 steps -- as many bytes out of L2 as the map itself, and the launch ran at 2.3 TB/s. There the per-group
// constants come from one small launch (gn_chan_finalize_kernel) and the SAME full-width grid only streams (fin != NULL). Measured
// (profiles/r06c20_groupnorm_pre_forms.txt, us one / two launches): B4 C128 512x512 248 / 117 (4.4 TB/s), B1 C128 512x768 177 / 40,
// B1 C256 512x768 273 / 100, B4 C256 256x256 78 / 58, B4 C256 128x128 (16 steps) 22.5 / 18.7; the other way round on maps that sit
// on the launch floor: B4 C320 64x64 (10 steps) 14.9 / 16.7, B2 C320 64x96 (16 steps, 3.9 M elements) 15.2 / 16.6.
constexpr int GN_PRE_SMALL_STEPS = 12;              // up to here: one launch
constexpr double GN_PRE_BIG_MAP = 8.0e6;            // elements; from GN_PRE_SMALL_STEPS + 1 steps on, maps this big take two launches
constexpr int GN_PRE_MAX_PROLOGUE_STEPS = MOS_GN_PRE_MAX_STEPS;

template <typename T>
int gn_nhwc_run_pre(GnNhwcArgs a, const float* chan_part, int tiles, int flags, hipStream_t st) {
    const int silu = flags & 1;
    const int vt = (a.V + 255) / 256;
    const dim3 grid(a.B, a.nsplit), block(a.TP * a.R);
    char key[96];
    snprintf(key, sizeof(key), "nhwc B%d C%d HW%d%s", a.B, a.C, a.HW, silu ? " +silu" : "");
    const double n = (double)a.B * a.C * a.HW;
    GnPreArgs q = {};
    const bool planned = gn_pre_plan(a, q, chan_part, tiles);
    bool one = planned;
    if (one) {
        const int steps = ((tiles * a.cpg + 63) / 64) * ((q.ngw + 3) / 4);
        if (flags & 4) one = false;                      // (tests / tools: the two-launch form)
        else if (!(flags & 8))                           // (8: the one-launch form whatever the prologue costs)
            one = steps <= GN_PRE_SMALL_STEPS || (steps <= GN_PRE_MAX_PROLOGUE_STEPS && n < GN_PRE_BIG_MAP);
    }
    const int nsp = planned ? (a.HW + q.ppb - 1) / q.ppb : 0;
    const dim3 g3(planned ? (unsigned)(a.C / q.cw) : 1u, (unsigned)nsp, (unsigned)a.B);
    if (one) {                                           // one launch: statistics of the range in the prologue, then the slice
        MosProfScope prof(st, "groupnorm_pre", key, 8.0 * n, 4.0 * n);
        if (silu) hipLaunchKernelGGL((gn_pre_apply_kernel<T, true>), g3, dim3(256), 0, st, q);
        else hipLaunchKernelGGL((gn_pre_apply_kernel<T, false>), g3, dim3(256), 0, st, q);
        return mos_check_launch("gn_pre_apply");
    }
    {
        MosProfScope prof(st, "groupnorm_finalize_pre", key, 2.0 * a.B * tiles * a.C, 8.0 * a.B * tiles * a.C);
        hipLaunchKernelGGL(gn_chan_finalize_kernel, dim3(a.B * a.G), dim3(256), 0, st, a, chan_part, tiles);
    }
    int rc = mos_check_launch("gn_chan_finalize");
    if (rc) return rc;
    if (planned) {                                       // ... then the same full-width streaming grid, its prologue skipped
        q.fin = a.partial + (int64_t)a.B * a.nsplit * a.G * 2;
        MosProfScope prof(st, "groupnorm_pre_stream", key, 8.0 * n, 4.0 * n);
        if (silu) hipLaunchKernelGGL((gn_pre_apply_kernel<T, true>), g3, dim3(256), 0, st, q);
        else hipLaunchKernelGGL((gn_pre_apply_kernel<T, false>), g3, dim3(256), 0, st, q);
        return mos_check_launch("gn_pre_apply");
    }
    MosProfScope prof(st, "groupnorm_apply", key, 8.0 * n, 4.0 * n);
#define GN_APPLY(VTN, S) hipLaunchKernelGGL((gn_nhwc_apply_kernel<T, VTN, false, S, false>), grid, block, 0, st, a)
    if (vt == 1) { if (silu) GN_APPLY(1, true); else GN_APPLY(1, false); }
    else { if (silu) GN_APPLY(2, true); else GN_APPLY(2, false); }
#undef GN_APPLY
    return mos_check_launch("gn_nhwc_apply");
}

int gn_nhwc_check(const void* x, const void* out, const float* gamma, const float* beta, void* ws, GnNhwcArgs& a,
                  const char* who) {
    if (!x || !out || !gamma || !beta || !ws) return mos_set_error(MOS_ERR_BAD_ARG, "%s: NULL argument", who);
    if (a.B <= 0 || a.C <= 0 || a.HW <= 0 || a.G <= 0 || a.C % a.G != 0 || a.C % 8 != 0)
        return mos_set_error(MOS_ERR_BAD_ARG, "%s: B=%d C=%d HW=%d G=%d (need C %% G == 0, C %% 8 == 0)", who, a.B, a.C, a.HW, a.G);
    a.cpg = a.C / a.G;
    if (!gn_nhwc_plan(a)) return mos_set_error(MOS_ERR_UNSUPPORTED, "%s: C=%d G=%d exceeds C <= 4096, G <= 64", who, a.C, a.G);
    return MOS_OK;
}


// ---- channels-last, ONE launch, ONE read: the "column" kernel -------------------------------------------------------------
// VERDICT r03 item 2. The slice kernels above need three launches (slice partials -> per-image constants -> apply) and read the
// activation twice, because the pixels of one (image, group) are spread over the workgroups of a grid; at the UNet's map sizes
// each of the three launches sits on its ~7 us floor (rocprofv3 minimum of every one of them), 20+ us per norm for tensors that
// move in 1-5 us. Here a workgroup owns a COLUMN of the image instead: all HW pixels x the smallest channel range that is
// both whole groups and whole 16-byte vectors (lcm(cpg, 8) channels: 40 for C = 320 / 640 / 1280, 80 for 2560, 120 for 960 /
// 1920 -- runs of 80 .. 240 contiguous bytes per pixel). Its slab (<= 128 KB for every 32x32-and-smaller map of the UNet, and
// for the 32x48 level of a 512x768 sample) stays in REGISTERS (K <= 16 vectors per thread at 510 threads), so the statistics
// are the exact two-pass form (mean, then sum (x - mean)^2), and the tensor is read once and written once by one launch.
// Slabs that do not fit (64x64 maps) take the slice kernels (a streaming form of this kernel lost: 62 vs 21 us at level 0).
// Workgroups are renumbered so that neighbouring columns (which share 128-byte lines) run on the same XCD / L2.
struct GnColArgs {
    const void* x; const void* dy; const void* ds; void* out;
    const float* gamma; const float* beta;
    float* stats;                                  // [B*G][2] = mean, rstd (written forward, read backward)
    int B, C, HW, G, cpg;
    int64_t ds_ps;                                 // pixel stride of ds (see GnNhwcArgs)
    int NV, ng, units, S, RP, npass;               // vectors / groups per unit, units per image, active threads, rows per pass
    float eps;
};

constexpr int GN_COL_NG = 4;                       // groups per unit (lcm(cpg, 8) / cpg <= 4 for cpg even)
constexpr int GN_COL_K = 16;                       // resident vectors per thread
constexpr int GN_COL_T = 512;                      // threads per workgroup (two waves per SIMD: 256 VGPRs each, no spills)

// block totals of 2 moments x ng groups; (a0, b0) belong to local group gl0, (a1, b1) to gl0 + 1. Result in totd[2 * g + m].
__device__ __forceinline__ void gn_col_block_sum(float a0, float b0, float a1, float b1, int gl0, float (*red)[2 * GN_COL_NG],
                                                 double* totd, int nwaves) {
    float v[2 * GN_COL_NG];
#pragma unroll
    for (int g = 0; g < GN_COL_NG; ++g) {
        v[2 * g] = (g == gl0 ? a0 : 0.f) + (g == gl0 + 1 ? a1 : 0.f);
        v[2 * g + 1] = (g == gl0 ? b0 : 0.f) + (g == gl0 + 1 ? b1 : 0.f);
    }
#pragma unroll
    for (int j = 0; j < 2 * GN_COL_NG; ++j)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[j] += __shfl_xor(v[j], off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 2 * GN_COL_NG; ++j) red[wave][j] = v[j];
    }
    __syncthreads();
    if (threadIdx.x < 2 * GN_COL_NG) {
        double s = 0.0;
        for (int w = 0; w < nwaves; ++w) s += (double)red[w][threadIdx.x];     // fixed order: deterministic
        totd[threadIdx.x] = s;
    }
    __syncthreads();
}

template <typename T, int K, bool BWD, bool SILU, bool DS>
__global__ __launch_bounds__(GN_COL_T) void gn_col_kernel(GnColArgs a) {
    static_assert(BWD || !DS, "the bypass gradient exists in backward only");
    typedef typename MT<T>::v8 v8;
    static_assert(K > 0, "the slab of a column is register-resident (a streaming form was measured in round 4 and lost)");
    __shared__ float red[GN_COL_T / 64][2 * GN_COL_NG];
    __shared__ double totd[2 * GN_COL_NG];
    const int tid = threadIdx.x;
    const int nwaves = (blockDim.x + 63) >> 6;
    const int nwg = gridDim.x, wg = blockIdx.x;
    const int logical = (nwg % 8 == 0) ? (wg % 8) * (nwg / 8) + wg / 8 : wg;     // XCD x runs a contiguous range of columns
    const int b = logical / a.units, u = logical - b * a.units;
    const bool active = tid < a.S;
    const int vec = active ? tid % a.NV : 0, row = active ? tid / a.NV : 0;
    const int cbase = (u * a.NV + vec) * 8;        // first channel of this thread's vector
    const int gl0 = (vec * 8) / a.cpg;             // local group of element 0 ; elements [nb, 8) belong to gl0 + 1
    const int nb = min(8, (gl0 + 1) * a.cpg - vec * 8);
    const int gidx0 = b * a.G + u * a.ng + gl0;    // row of `stats`
    const bool two = nb < 8;                        // (then gl0 + 1 < ng: a unit is whole groups)
    float gam[8], bet[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { gam[i] = a.gamma[cbase + i]; bet[i] = a.beta[cbase + i]; }
    const int64_t img = (int64_t)b * a.HW * a.C;
    const T* xb = (const T*)a.x + img + cbase;
    [[maybe_unused]] const T* db = BWD ? (const T*)a.dy + img + cbase : nullptr;
    [[maybe_unused]] const T* sb = DS ? (const T*)a.ds + (int64_t)b * a.HW * a.ds_ps + cbase : nullptr;
    T* ob = (T*)a.out + img + cbase;
    const double ninv = 1.0 / ((double)a.cpg * (double)a.HW);

    u32x4 xr[K];
    [[maybe_unused]] u32x4 dr[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int p = row + k * a.RP;
        const bool ok = active && p < a.HW;
        xr[k] = ok ? ld16(xb + (int64_t)p * a.C) : u32x4{0, 0, 0, 0};
        if constexpr (BWD) dr[k] = ok ? ld16(db + (int64_t)p * a.C) : u32x4{0, 0, 0, 0};
    }

    float m0 = 0.f, r0 = 0.f, m1 = 0.f, r1 = 0.f;   // mean / rstd of gl0 and gl0 + 1
    if constexpr (!BWD) {
        // exact two-pass statistics from the registers
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const v8 xv = as_v8<T>(xr[k]);     // (rows past HW are zeros: they add nothing)
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float f = (float)xv[i]; if (i < nb) a0 += f; else a1 += f; }
        }
        gn_col_block_sum(a0, 0.f, a1, 0.f, gl0, red, totd, nwaves);
        m0 = (float)(totd[2 * gl0] * ninv);
        m1 = two ? (float)(totd[2 * (gl0 + 1)] * ninv) : 0.f;
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (!(active && row + k * a.RP < a.HW)) continue;
            const v8 xv = as_v8<T>(xr[k]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float d = (float)xv[i] - (i < nb ? m0 : m1);
                if (i < nb) q0 += d * d; else q1 += d * d;
            }
        }
        gn_col_block_sum(q0, 0.f, q1, 0.f, gl0, red, totd, nwaves);
        r0 = (float)(1.0 / sqrt(totd[2 * gl0] * ninv + (double)a.eps));
        r1 = two ? (float)(1.0 / sqrt(totd[2 * (gl0 + 1)] * ninv + (double)a.eps)) : 0.f;
    
        if (a.stats != nullptr && active && row == 0) {   // one writer per group: the thread whose vector starts the group
            if (vec * 8 == gl0 * a.cpg) { a.stats[gidx0 * 2] = m0; a.stats[gidx0 * 2 + 1] = r0; }
            if (two) { a.stats[(gidx0 + 1) * 2] = m1; a.stats[(gidx0 + 1) * 2 + 1] = r1; }
        }
        float c0[8], c1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float mm = i < nb ? m0 : m1, rr = i < nb ? r0 : r1;
            c0[i] = gam[i] * rr; c1[i] = bet[i] - mm * gam[i] * rr;
        }
        auto apply = [&](u32x4 raw, int p) {
            const v8 xv = as_v8<T>(raw);
            v8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float z = (float)xv[i] * c0[i] + c1[i];
                if (SILU) z = silu_f(z);
                o[i] = (T)z;
            }
            st16(ob + (int64_t)p * a.C, from_v8<T>(o));
        };
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int p = row + k * a.RP;
            if (active && p < a.HW) apply(xr[k], p);
        }
    
    } else {
        m0 = a.stats[gidx0 * 2]; r0 = a.stats[gidx0 * 2 + 1];
        if (two) { m1 = a.stats[(gidx0 + 1) * 2]; r1 = a.stats[(gidx0 + 1) * 2 + 1]; }
        // g = dL/dz * gamma (z = xhat * gamma + beta, y = silu(z) or z) and xhat of one vector
        auto terms = [&](u32x4 xraw, u32x4 draw, float* gg, float* xh) {
            const v8 xv = as_v8<T>(xraw), dv = as_v8<T>(draw);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xh[i] = ((float)xv[i] - (i < nb ? m0 : m1)) * (i < nb ? r0 : r1);
                float dz = (float)dv[i];
                if (SILU) {
                    const float z = xh[i] * gam[i] + bet[i];
                    const float sig = 1.f / (1.f + __expf(-z));
                    dz *= sig * (1.f + z * (1.f - sig));
                }
                gg[i] = dz * gam[i];
            }
        };
        float a0 = 0.f, b0 = 0.f, a1 = 0.f, b1 = 0.f;
        auto accumulate = [&](u32x4 xraw, u32x4 draw) {
            float gg[8], xh[8];
            terms(xraw, draw, gg, xh);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i < nb) { a0 += gg[i]; b0 += gg[i] * xh[i]; } else { a1 += gg[i]; b1 += gg[i] * xh[i]; }
            }
        };
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (active && row + k * a.RP < a.HW) accumulate(xr[k], dr[k]);
    
        gn_col_block_sum(a0, b0, a1, b1, gl0, red, totd, nwaves);
        const float mg0 = (float)(totd[2 * gl0] * ninv), mx0 = (float)(totd[2 * gl0 + 1] * ninv);
        const float mg1 = two ? (float)(totd[2 * (gl0 + 1)] * ninv) : 0.f, mx1 = two ? (float)(totd[2 * (gl0 + 1) + 1] * ninv) : 0.f;
        auto apply = [&](u32x4 xraw, u32x4 draw, int p) {
            float gg[8], xh[8];
            terms(xraw, draw, gg, xh);
            v8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                o[i] = (T)((i < nb ? r0 : r1) * (gg[i] - (i < nb ? mg0 : mg1) - xh[i] * (i < nb ? mx0 : mx1)));
            if constexpr (DS) {
                const v8 sv = as_v8<T>(ld16(sb + (int64_t)p * a.ds_ps));
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (T)((float)o[i] + (float)sv[i]);
            }
            st16(ob + (int64_t)p * a.C, from_v8<T>(o));
        };
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int p = row + k * a.RP;
            if (active && p < a.HW) apply(xr[k], dr[k], p);
        }
    
    }
}

int gn_gcd(int x, int y) { while (y) { const int t = x % y; x = y; y = t; } return x; }

// true: `c` is filled in and the column kernel takes this call (the slab fits GN_COL_K vectors per thread)
bool gn_col_plan(const GnNhwcArgs& a, GnColArgs& c) {
    const int cpg = a.C / a.G;
    const int cu = cpg / gn_gcd(cpg, 8) * 8;        // lcm(cpg, 8): whole groups AND whole 16-byte vectors
    if (cpg < 8 || cu < 32 || cu > 128 || a.C % cu != 0 || cu / cpg > GN_COL_NG) return false;   // (cpg >= 8: a vector spans <= 2 groups)
    c.B = a.B; c.C = a.C; c.HW = a.HW; c.G = a.G; c.cpg = cpg; c.eps = a.eps;
    c.NV = cu / 8; c.ng = cu / cpg; c.units = a.C / cu;
    const int64_t vectors = (int64_t)a.HW * c.NV;
    int threads = (int)((vectors + 63) / 64 * 64);
    if (threads > GN_COL_T) threads = GN_COL_T;
    if (threads < 64) threads = 64;
    c.S = threads / c.NV * c.NV;
    c.RP = c.S / c.NV;
    c.npass = (a.HW + c.RP - 1) / c.RP;
    if (c.npass > GN_COL_K) return false;       // larger slabs (level 0, VAE): the slice kernels
    c.x = a.x; c.dy = a.dy; c.ds = a.ds; c.ds_ps = a.ds_ps; c.out = a.out; c.gamma = a.gamma; c.beta = a.beta; c.stats = a.stats;
    return true;
}

template <typename T, bool BWD>
int gn_col_run(const GnColArgs& c, int silu, hipStream_t st) {
    const int threads = (c.S + 63) / 64 * 64;
    const dim3 grid(c.units * c.B), block(threads);
    char key[96];
    snprintf(key, sizeof(key), "nhwc B%d C%d HW%d%s", c.B, c.C, c.HW, silu ? " +silu" : "");
    const double n = (double)c.B * c.C * c.HW;
    const bool has_ds = BWD && c.ds != nullptr;
    MosProfScope prof(st, BWD ? "groupnorm_bwd_fused" : "groupnorm_fused", key, (BWD ? 26.0 : 11.0) * n,
                      (BWD ? (has_ds ? 8.0 : 6.0) : 4.0) * n);
#define GN_COL(S_, D_) hipLaunchKernelGGL((gn_col_kernel<T, GN_COL_K, BWD, S_, D_>), grid, block, 0, st, c)
    if constexpr (BWD) {
        if (has_ds) {
            if (silu) GN_COL(true, true); else GN_COL(false, true);
            return mos_check_launch("gn_col");
        }
    }
    if (silu) GN_COL(true, false); else GN_COL(false, false);
#undef GN_COL
    return mos_check_launch("gn_col");
}

// flags: bit 0 = SiLU, bit 1 (MOS_GN_FORCE_SLICES) = the three-launch slice form even where the column kernel applies (tests / A-B)
template <bool BWD>
int gn_nhwc_dispatch(GnNhwcArgs& a, int flags, int dtype, hipStream_t st, const char* who) {
    const int silu = flags & 1;
    GnColArgs c = {};
    if (!(flags & MOS_GN_FORCE_SLICES) && gn_col_plan(a, c)) {
        if (dtype == MOS_F16) return gn_col_run<f16_t, BWD>(c, silu, st);
        if (dtype == MOS_BF16) return gn_col_run<bf16_t, BWD>(c, silu, st);
        return mos_set_error(MOS_ERR_UNSUPPORTED, "%s: dtype %d", who, dtype);
    }
    if (dtype == MOS_F16) return gn_nhwc_run<f16_t, BWD>(a, silu, st);
    if (dtype == MOS_BF16) return gn_nhwc_run<bf16_t, BWD>(a, silu, st);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "%s: dtype %d", who, dtype);
}

}  // namespace

extern "C" {

int64_t mos_groupnorm_workspace_bytes(int B, int C, int HW, int G) {
    if (B <= 0 || C <= 0 || HW <= 0 || G <= 0 || C % G) return 0;
    return (int64_t)B * G * gn_nsplit((int64_t)(C / G) * HW) * 2 * (int64_t)sizeof(float);
}

int mos_groupnorm_silu_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, void* ws, int B,
                           int C, int HW, int G, float eps, int silu, int dtype, void* stream) {
    int rc = gn_check(x, y, gamma, beta, ws, B, C, HW, G, "mos_groupnorm_silu_fwd");
    if (rc) return rc;
    GnArgs a = gn_args(x, nullptr, y, gamma, beta, stats, ws, B, C, HW, G, eps);
    if (dtype == MOS_F16) return gn_fwd<f16_t>(a, silu, (hipStream_t)stream);
    if (dtype == MOS_BF16) return gn_fwd<bf16_t>(a, silu, (hipStream_t)stream);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_groupnorm_silu_fwd: dtype %d", dtype);
}

int mos_groupnorm_silu_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* stats,
                           void* dx, void* ws, int B, int C, int HW, int G, int silu, int dtype, void* stream) {
    int rc = gn_check(x, dx, gamma, beta, ws, B, C, HW, G, "mos_groupnorm_silu_bwd");
    if (rc) return rc;
    MOS_REQUIRE(dy && stats, "mos_groupnorm_silu_bwd: NULL dy / stats");
    GnArgs a = gn_args(x, dy, dx, gamma, beta, const_cast<float*>(stats), ws, B, C, HW, G, 0.f);
    if (dtype == MOS_F16) return gn_bwd<f16_t>(a, silu, (hipStream_t)stream);
    if (dtype == MOS_BF16) return gn_bwd<bf16_t>(a, silu, (hipStream_t)stream);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_groupnorm_silu_bwd: dtype %d", dtype);
}

int64_t mos_groupnorm_nhwc_workspace_bytes(int B, int C, int HW, int G) {
    GnNhwcArgs a = {};
    a.B = B; a.C = C; a.HW = HW; a.G = G;
    if (B <= 0 || C <= 0 || HW <= 0 || G <= 0 || C % G || C % 8) return 0;
    a.cpg = C / G;
    if (!gn_nhwc_plan(a)) return 0;
    return ((int64_t)B * a.nsplit * G * 2 + (int64_t)B * G * 2) * (int64_t)sizeof(float);   // slice partials + finalised constants
}

int mos_groupnorm_silu_fwd_nhwc(const void* x, const float* gamma, const float* beta, void* y, float* stats, void* ws,
                                int B, int C, int HW, int G, float eps, int silu, int dtype, void* stream) {
    GnNhwcArgs a = {};
    a.x = x; a.out = y; a.gamma = gamma; a.beta = beta; a.stats = stats; a.partial = (float*)ws;
    a.B = B; a.C = C; a.HW = HW; a.G = G; a.eps = eps;
    int rc = gn_nhwc_check(x, y, gamma, beta, ws, a, "mos_groupnorm_silu_fwd_nhwc");
    if (rc) return rc;
    return gn_nhwc_dispatch<false>(a, silu, dtype, (hipStream_t)stream, "mos_groupnorm_silu_fwd_nhwc");
}

/* GroupNorm(+SiLU) forward on a map whose per-channel statistics came with it (round 6): `chan_part` [B][tiles][C][2] fp32 = sum and
 * sum of squares of the stored values per (pixel tile, channel), as mos_conv3x3_nhwc_gn leaves them. Same y / stats / ws as
 * mos_groupnorm_silu_fwd_nhwc; two launches (finalize over the tiles, apply) instead of three and ONE read of x instead of two.
 * mos_groupnorm_nhwc_reads_twice: 1 where the plain entry point would take the three-launch slice form (the producer's statistics
 * pay), 0 where it runs the one-launch column kernel (they do not). */
int mos_groupnorm_nhwc_reads_twice(int B, int C, int HW, int G) {
    GnNhwcArgs a = {};
    a.B = B; a.C = C; a.HW = HW; a.G = G;
    if (B <= 0 || C <= 0 || HW <= 0 || G <= 0 || C % G || C % 8) return 0;
    a.cpg = C / G;
    if (!gn_nhwc_plan(a)) return 0;
    GnColArgs c = {};
    return gn_col_plan(a, c) ? 0 : 1;
}

int mos_groupnorm_silu_fwd_nhwc_pre(const void* x, const float* chan_part, int tiles_per_image, const float* gamma,
                                    const float* beta, void* y, float* stats, void* ws, int B, int C, int HW, int G, float eps,
                                    int silu, int dtype, void* stream) {
    GnNhwcArgs a = {};
    a.x = x; a.out = y; a.gamma = gamma; a.beta = beta; a.stats = stats; a.partial = (float*)ws;
    a.B = B; a.C = C; a.HW = HW; a.G = G; a.eps = eps;
    int rc = gn_nhwc_check(x, y, gamma, beta, ws, a, "mos_groupnorm_silu_fwd_nhwc_pre");
    if (rc) return rc;
    MOS_REQUIRE(chan_part && tiles_per_image > 0, "mos_groupnorm_silu_fwd_nhwc_pre: no channel statistics");
    if (dtype == MOS_F16) return gn_nhwc_run_pre<f16_t>(a, chan_part, tiles_per_image, silu, (hipStream_t)stream);
    if (dtype == MOS_BF16) return gn_nhwc_run_pre<bf16_t>(a, chan_part, tiles_per_image, silu, (hipStream_t)stream);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_groupnorm_silu_fwd_nhwc_pre: dtype %d", dtype);
}

int mos_groupnorm_silu_bwd_nhwc(const void* dy, const void* x, const float* gamma, const float* beta, const float* stats,
                                void* dx, void* ws, int B, int C, int HW, int G, int silu, int dtype, void* stream) {
    GnNhwcArgs a = {};
    a.x = x; a.dy = dy; a.out = dx; a.gamma = gamma; a.beta = beta; a.stats = const_cast<float*>(stats); a.partial = (float*)ws;
    a.B = B; a.C = C; a.HW = HW; a.G = G; a.eps = 0.f;
    int rc = gn_nhwc_check(x, dx, gamma, beta, ws, a, "mos_groupnorm_silu_bwd_nhwc");
    if (rc) return rc;
    MOS_REQUIRE(dy && stats, "mos_groupnorm_silu_bwd_nhwc: NULL dy / stats");
    return gn_nhwc_dispatch<true>(a, silu, dtype, (hipStream_t)stream, "mos_groupnorm_silu_bwd_nhwc");
}

/* mos_groupnorm_silu_bwd_nhwc plus the gradient `ds` (x's shape, layout and dtype) of a residual connection that bypasses
 * the norm: dx = half(GN_bwd(dy)) + ds, i.e. exactly what autograd's accumulation of the two gradients yields, without the
 * extra elementwise launch (ResnetBlock2D / Transformer2DModel inputs feed both a GroupNorm and a skip path). */
int mos_groupnorm_silu_bwd_nhwc_res(const void* dy, const void* ds, const void* x, const float* gamma, const float* beta,
                                    const float* stats, void* dx, void* ws, int B, int C, int HW, int G, int silu, int dtype,
                                    void* stream) {
    return mos_groupnorm_silu_bwd_nhwc_res_ps(dy, ds, (int64_t)C, x, gamma, beta, stats, dx, ws, B, C, HW, G, silu, dtype, stream);
}

/* The same with `ds` read in place from a channel slice of a wider channels-last tensor (round 6): ds_pixel_stride = elements
 * between consecutive pixels of ds (>= C, a multiple of 8; the batch stride is HW * ds_pixel_stride) -- the gradient autograd hands
 * to one input of a torch.cat along the channels (the UNet's skip concatenations) without the contiguous copy. */
int mos_groupnorm_silu_bwd_nhwc_res_ps(const void* dy, const void* ds, int64_t ds_pixel_stride, const void* x, const float* gamma,
                                       const float* beta, const float* stats, void* dx, void* ws, int B, int C, int HW, int G,
                                       int silu, int dtype, void* stream) {
    GnNhwcArgs a = {};
    a.x = x; a.dy = dy; a.out = dx; a.gamma = gamma; a.beta = beta; a.stats = const_cast<float*>(stats); a.partial = (float*)ws;
    a.B = B; a.C = C; a.HW = HW; a.G = G; a.eps = 0.f; a.ds = ds; a.ds_ps = ds_pixel_stride;
    int rc = gn_nhwc_check(x, dx, gamma, beta, ws, a, "mos_groupnorm_silu_bwd_nhwc_res");
    if (rc) return rc;
    MOS_REQUIRE(dy && stats && ds, "mos_groupnorm_silu_bwd_nhwc_res: NULL dy / ds / stats");
    MOS_REQUIRE(ds_pixel_stride >= C && ds_pixel_stride % 8 == 0 && ((uint64_t)ds & 15) == 0,
                "mos_groupnorm_silu_bwd_nhwc_res_ps: ds pixel stride %lld (need >= C, a multiple of 8, 16-byte aligned base)",
                (long long)ds_pixel_stride);
    return gn_nhwc_dispatch<true>(a, silu, dtype, (hipStream_t)stream, "mos_groupnorm_silu_bwd_nhwc_res");
}

}  // extern "C"

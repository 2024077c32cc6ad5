// mos_gemm.hip — LoRA-augmented linear layers for gfx950.
//
// Replaces LoRALinearLayer.forward (reference mixofshow/models/edlora.py:244-246):
//     y = orig(x) + alpha * lora_up(lora_down(x))
// which the reference runs as 3 GEMM launches (base, K=..->4, 4->N) + scale + add per site.
// Here:  t = x . A16^T  (one skinny pass over x for ALL fused sites),  then ONE MFMA GEMM with the
// rank dimension appended to the contraction:  y = [x | t] . [W | alpha*B]^T + bias.
// Backward (W frozen): dt = dy . BpT^T ; dx = [dy | dt] . [W^T | A^T]^T ; dA = dt^T x ; dB = t^T dy.
//
// MFMA conventions (v_mfma_f32_16x16x32_{f16,bf16}; D[i][j] = sum_k A[i][k] B[k][j]):
//   lane l supplies A[i = l&15][k = 8*(l>>4) .. +8] and B[k = 8*(l>>4) .. +8][j = l&15];
//   D: lane l holds D[i = 4*(l>>4) + reg][j = l&15], reg = 0..3.
// We feed A <- weight rows (n), B <- activation rows (m), so every lane ends up with 4
// CONSECUTIVE output features of one token: an 8-byte store.
#include <cstdio>
#include <type_traits>
#include "mos_common.h"

// tuning knobs (tools/build_variant.sh)
#ifndef MOS_TN_REDUCE_UNROLL
#define MOS_TN_REDUCE_UNROLL 0
#endif
#ifndef MOS_TN_TARGET_WG
#define MOS_TN_TARGET_WG 512      // workgroups the LoRA-gradient kernel aims at (tuning knob)
#endif

namespace {

constexpr int GEMM_BM = 128;
constexpr int GEMM_BK = 64;
constexpr int GEMM_LDS_STRIDE = GEMM_BK + 8;  // 144 B rows: 16B-slot index r*9 mod 16 is a bijection

template <typename T, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(
    const T* __restrict__ X, int64_t ldx, const T* __restrict__ W, int64_t ldw,
    const T* __restrict__ Taug, const T* __restrict__ Baug, const float* __restrict__ bias,
    T* __restrict__ Y, int64_t ldy, int M, int N, int K) {
    typedef typename MT<T>::v8 v8;
    typedef typename MT<T>::v4 v4;
    constexpr int NJ = BN / 32;             // 16-wide n tiles per wave (wave tile = 64 m x BN/2 n)
    constexpr int XCH = GEMM_BM * 8 / 256;  // 16B chunks of the X tile per thread (4)
    constexpr int WCH = BN * 8 / 256;       // 16B chunks of the W tile per thread (4 or 2)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Xs = reinterpret_cast<T*>(smem_raw);                       // [2][128][72]
    T* Ws = Xs + 2 * GEMM_BM * GEMM_LDS_STRIDE;                   // [2][BN][72]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * BN;
    const int m0 = blockIdx.y * GEMM_BM;

    f32x4 acc[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 xr[XCH], wr[WCH];
    const int nk = (K + GEMM_BK - 1) / GEMM_BK;

    auto load_tile = [&](int kt) {
        const int k0 = kt * GEMM_BK;
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int c = tid + 256 * i;
            const int row = c >> 3, cc = c & 7;
            const int m = min(m0 + row, M - 1);
            const int k = k0 + cc * 8;
            xr[i] = (k < K) ? ld16(X + (int64_t)m * ldx + k) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int c = tid + 256 * i;
            const int row = c >> 3, cc = c & 7;
            const int n = min(n0 + row, N - 1);
            const int k = k0 + cc * 8;
            wr[i] = (k < K) ? ld16(W + (int64_t)n * ldw + k) : u32x4{0, 0, 0, 0};
        }
    };
    auto store_tile = [&](int buf) {
        T* xs = Xs + buf * GEMM_BM * GEMM_LDS_STRIDE;
        T* ws = Ws + buf * BN * GEMM_LDS_STRIDE;
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int c = tid + 256 * i;
            st16(xs + (c >> 3) * GEMM_LDS_STRIDE + (c & 7) * 8, xr[i]);
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int c = tid + 256 * i;
            st16(ws + (c >> 3) * GEMM_LDS_STRIDE + (c & 7) * 8, wr[i]);
        }
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);  // global loads in flight under the MFMAs below
        const T* xs = Xs + cur * GEMM_BM * GEMM_LDS_STRIDE + (wm * 64 + l15) * GEMM_LDS_STRIDE + lg * 8;
        const T* ws = Ws + cur * BN * GEMM_LDS_STRIDE + (wn * (BN / 2) + l15) * GEMM_LDS_STRIDE + lg * 8;
#pragma unroll
        for (int kk = 0; kk < GEMM_BK / 32; ++kk) {
            v8 bfrag[4], afrag[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                bfrag[i] = as_v8<T>(ld16(xs + i * 16 * GEMM_LDS_STRIDE + kk * 32));
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                afrag[j] = as_v8<T>(ld16(ws + j * 16 * GEMM_LDS_STRIDE + kk * 32));
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = MT<T>::mfma16(afrag[j], bfrag[i], acc[j][i]);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // Rank augmentation: y += t[M,16] . Baug[N,16]^T, one K=16 MFMA per tile
    // (v_mfma_f32_16x16x16: lane supplies A[i=l&15][k=4*(l>>4)..+4], B[k][j=l&15]).
    if (Taug != nullptr) {
        v4 tb[4], ba[NJ];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = min(m0 + wm * 64 + i * 16 + l15, M - 1);
            tb[i] = __builtin_bit_cast(v4, ld8(Taug + (int64_t)m * MOS_LORA_PAD + lg * 4));
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = min(n0 + wn * (BN / 2) + j * 16 + l15, N - 1);
            ba[j] = __builtin_bit_cast(v4, ld8(Baug + (int64_t)n * MOS_LORA_PAD + lg * 4));
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = MT<T>::mfma16k16(ba[j], tb[i], acc[j][i]);
    }

    // Epilogue: lane holds features nb..nb+3 of token m.
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int nb = n0 + wn * (BN / 2) + j * 16 + lg * 4;
        if (nb >= N) continue;
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        if (bias != nullptr) {
            b0 = bias[nb]; b1 = bias[nb + 1]; b2 = bias[nb + 2]; b3 = bias[nb + 3];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + l15;
            if (m < M) {
                const f32x4 a = acc[j][i];
                st8(Y + (int64_t)m * ldy + nb, pack4<T>(a[0] + b0, a[1] + b1, a[2] + b2, a[3] + b3));
            }
        }
    }
}

// t[M,16] = X[M,K] . S[16,K]^T  — lora_down of all fused sites in one pass over X (HBM-bound).
// One wave = 16 tokens (one MFMA column tile): many small waves keep every CU busy at M = 5k..16k; the K loop is
// unrolled x4 with all loads issued unconditionally up front (a `cond ? load : 0` select would serialise them).
template <typename T>
__global__ __launch_bounds__(256) void skinny_nt_kernel(const T* __restrict__ X, int64_t ldx,
                                                        const T* __restrict__ S, T* __restrict__ Tout,
                                                        int M, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int mbase = (blockIdx.x * 4 + wave) * 16;
    if (mbase >= M) return;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const T* xrow = X + (int64_t)min(mbase + l15, M - 1) * ldx + lg * 8;
    const T* srow = S + (int64_t)l15 * K + lg * 8;
    int k = 0;
    for (; k + 128 <= K; k += 128) {
        u32x4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = ld16(srow + k + 32 * u); b[u] = ld16(xrow + k + 32 * u); }
        acc0 = MT<T>::mfma16(as_v8<T>(a[0]), as_v8<T>(b[0]), acc0);
        acc1 = MT<T>::mfma16(as_v8<T>(a[1]), as_v8<T>(b[1]), acc1);
        acc0 = MT<T>::mfma16(as_v8<T>(a[2]), as_v8<T>(b[2]), acc0);
        acc1 = MT<T>::mfma16(as_v8<T>(a[3]), as_v8<T>(b[3]), acc1);
    }
    for (; k < K; k += 32) {  // tail (K % 8 == 0): chunks past K contribute zeros
        const int kk = min(k + lg * 8, K - 8) - lg * 8;
        const bool ok = (k + lg * 8) < K;
        u32x4 a = ld16(srow + kk), b = ld16(xrow + kk);
        if (!ok) a = u32x4{0, 0, 0, 0};
        acc0 = MT<T>::mfma16(as_v8<T>(a), as_v8<T>(b), acc0);
    }
    const int m = mbase + l15;
    if (m < M)
        st8(Tout + (int64_t)m * MOS_LORA_PAD + lg * 4,
            pack4<T>(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]));
}

// partial[chunk][j][c] = sum_{m in chunk} P[m][j] * Z[m][c]   (j < NJ) — LoRA factor gradients.
// Reduction over tokens is HBM-bound (Z read once); VALU FMAs: NJ per loaded element.
// Block: 256 threads = 32 row-lanes x 8 column-lanes (8 columns each) -> 64 columns per block.
template <typename T, int NJ>
__global__ __launch_bounds__(256) void skinny_tn_kernel(const T* __restrict__ P, const T* __restrict__ Z,
                                                        int64_t ldz, float* __restrict__ partial,
                                                        int M, int C, int rows_per_chunk) {
    typedef typename MT<T>::v8 v8;
    __shared__ float red[4][NJ][64];
    const int tid = threadIdx.x;
    const int cg = tid & 7, ry = tid >> 3;           // lane: ry%8 = lane>>3
    const int c0 = blockIdx.y * 64 + cg * 8;
    const int mb = blockIdx.x * rows_per_chunk;
    const int me = min(mb + rows_per_chunk, M);
    float acc[NJ][8];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
    if (c0 < C) {
        for (int m = mb + ry; m < me; m += 32) {
            const v8 z = as_v8<T>(ld16(Z + (int64_t)m * ldz + c0));
            float zf[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) zf[e] = (float)z[e];
            const T* prow = P + (int64_t)m * MOS_LORA_PAD;
#pragma unroll
            for (int j4 = 0; j4 < NJ; j4 += 4) {
                const typename MT<T>::v4 p = __builtin_bit_cast(typename MT<T>::v4, ld8(prow + j4));
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float pf = (float)p[jj];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[j4 + jj][e] += pf * zf[e];
                }
            }
        }
    }
    // reduce the 8 row-lanes inside each wave (lane bits 3..5), then the 4 waves through LDS
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = acc[j][e];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            acc[j][e] = v;
        }
    const int lane = tid & 63, wave = tid >> 6;
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wave][j][lane * 8 + e] = acc[j][e];
    }
    __syncthreads();
    for (int idx = tid; idx < NJ * 64; idx += 256) {
        const int j = idx >> 6, c = idx & 63;
        const int col = blockIdx.y * 64 + c;
        if (col < C) {
            const float s = red[0][j][c] + red[1][j][c] + red[2][j][c] + red[3][j][c];
            partial[((int64_t)blockIdx.x * NJ + j) * C + col] = s;
        }
    }
}

// out[j][c] = sum_chunk partial[chunk][j][c]; rows j >= NJ of the 16-row output are zeroed.
__global__ void skinny_tn_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                        int nchunk, int NJ, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= MOS_LORA_PAD * C) return;
    const int j = idx / C, c = idx - j * C;
#if MOS_TN_REDUCE_UNROLL
    // experimental: four independent accumulators so that the loads of a thread are in flight together (the plain
    // loop is a chain of ~100 dependent L2 round trips per thread: ~10 us for a few hundred KB)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < NJ) {
        const float* pp = partial + (int64_t)j * C + c;
        const int64_t cs = (int64_t)NJ * C;
        int ch = 0;
        for (; ch + 4 <= nchunk; ch += 4) {
            s0 += pp[(ch + 0) * cs];
            s1 += pp[(ch + 1) * cs];
            s2 += pp[(ch + 2) * cs];
            s3 += pp[(ch + 3) * cs];
        }
        for (; ch < nchunk; ++ch) s0 += pp[ch * cs];
    }
    out[idx] = (s0 + s1) + (s2 + s3);
#else
    float s = 0.f;
    if (j < NJ)
        for (int ch = 0; ch < nchunk; ++ch) s += partial[((int64_t)ch * NJ + j) * C + c];
    out[idx] = s;
#endif
}

template <typename T>
__global__ void lora_pack_kernel(mos_lora_sites s, T* __restrict__ A16, T* __restrict__ A16T,
                                 T* __restrict__ Bp16, T* __restrict__ BpT) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int nA = MOS_LORA_PAD * s.K;
    if (idx < nA) {
        const int j = idx / s.K, k = idx - j * s.K;
        float v = 0.f;
        const int g = j / s.rank;
        if (g < s.n_sites) v = s.down[g][(int64_t)(j - g * s.rank) * s.K + k];
        A16[idx] = (T)v;
        A16T[(int64_t)k * MOS_LORA_PAD + j] = (T)v;
    }
    const int nB = s.N * MOS_LORA_PAD;
    if (idx < nB) {
        const int n = idx / MOS_LORA_PAD, j = idx - n * MOS_LORA_PAD;
        float v = 0.f;
        const int g = j / s.rank;
        if (g < s.n_sites && n >= s.n_begin[g] && n < s.n_begin[g] + s.n_rows[g])
            v = s.alpha[g] * s.up[g][(int64_t)(n - s.n_begin[g]) * s.rank + (j - g * s.rank)];
        Bp16[idx] = (T)v;
        BpT[(int64_t)j * s.N + n] = (T)v;
    }
}

template <typename T>
int launch_gemm(const void* X, int64_t ldx, const void* W, int64_t ldw, const void* t, const void* Bp,
                const float* bias, void* Y, int64_t ldy, int M, int N, int K, hipStream_t st) {
    // BN=64 wastes no columns at N = 320 (5 tiles); BN=128 otherwise halves the X re-reads.
    char key[96];
    snprintf(key, sizeof(key), "%s M%d N%d K%d%s", sizeof(T) == 2 && std::is_same<T, f16_t>::value ? "f16" : "bf16", M, N, K,
             t ? " +lora" : "");
    MosProfScope prof(st, "gemm_nt", key, 2.0 * M * (double)N * (K + (t ? 16 : 0)),
                      2.0 * ((double)M * K + (double)N * K + (double)M * N));
    const bool wide = (N % 128 == 0) && ((int64_t)((N + 127) / 128) * ((M + 127) / 128) >= 256);
    if (wide) {
        constexpr int BN = 128;
        const size_t lds = 2 * (GEMM_BM + BN) * GEMM_LDS_STRIDE * sizeof(T);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<T, BN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        dim3 grid((N + BN - 1) / BN, (M + GEMM_BM - 1) / GEMM_BM);
        hipLaunchKernelGGL((gemm_nt_kernel<T, BN>), grid, dim3(256), lds, st, (const T*)X, ldx, (const T*)W, ldw,
                           (const T*)t, (const T*)Bp, bias, (T*)Y, ldy, M, N, K);
    } else {
        constexpr int BN = 64;
        const size_t lds = 2 * (GEMM_BM + BN) * GEMM_LDS_STRIDE * sizeof(T);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<T, BN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        dim3 grid((N + BN - 1) / BN, (M + GEMM_BM - 1) / GEMM_BM);
        hipLaunchKernelGGL((gemm_nt_kernel<T, BN>), grid, dim3(256), lds, st, (const T*)X, ldx, (const T*)W, ldw,
                           (const T*)t, (const T*)Bp, bias, (T*)Y, ldy, M, N, K);
    }
    return mos_check_launch("gemm_nt");
}

template <typename T>
int launch_skinny_nt(const void* X, int64_t ldx, const void* S, void* Tout, int M, int K, hipStream_t st) {
    dim3 grid((M + 63) / 64);
    char key[64];
    snprintf(key, sizeof(key), "M%d K%d", M, K);
    MosProfScope prof(st, "lora_down(skinny_nt)", key, 2.0 * M * 16.0 * K, 2.0 * ((double)M * K + 16.0 * K + 16.0 * M));
    hipLaunchKernelGGL((skinny_nt_kernel<T>), grid, dim3(256), 0, st, (const T*)X, ldx, (const T*)S, (T*)Tout, M, K);
    return mos_check_launch("skinny_nt");
}

inline int tn_rows_per_chunk(int M, int C) {
    const int colblocks = (C + 63) / 64;
    int nchunk = (MOS_TN_TARGET_WG + colblocks - 1) / colblocks;   // aim at >= ~512 workgroups
    const int maxchunk = (M + 63) / 64;
    if (nchunk > maxchunk) nchunk = maxchunk;
    if (nchunk < 1) nchunk = 1;
    int rpc = (M + nchunk - 1) / nchunk;
    rpc = (rpc + 31) / 32 * 32;
    return rpc;
}

template <typename T, int NJ>
int launch_skinny_tn_nj(const void* P, const void* Z, int64_t ldz, float* out, float* partial, int M, int C,
                        hipStream_t st) {
    const int rpc = tn_rows_per_chunk(M, C);
    const int nchunk = (M + rpc - 1) / rpc;
    dim3 grid(nchunk, (C + 63) / 64);
    char key[64];
    snprintf(key, sizeof(key), "M%d C%d r%d", M, C, NJ);
    MosProfScope prof(st, "lora_grad(skinny_tn)", key, 2.0 * M * (double)NJ * C, 2.0 * ((double)M * C + 16.0 * M));
    hipLaunchKernelGGL((skinny_tn_kernel<T, NJ>), grid, dim3(256), 0, st, (const T*)P, (const T*)Z, ldz, partial, M, C,
                       rpc);
    int rc = mos_check_launch("skinny_tn");
    if (rc) return rc;
    const int tot = MOS_LORA_PAD * C;
    hipLaunchKernelGGL(skinny_tn_reduce_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, partial, out, nchunk, NJ, C);
    return mos_check_launch("skinny_tn_reduce");
}

// cols = number of packed LoRA columns in use (n_sites * rank); rounded up to a multiple of 4
template <typename T>
int launch_skinny_tn(const void* P, const void* Z, int64_t ldz, float* out, float* partial, int M, int C, int cols,
                     hipStream_t st) {
    if (cols <= 4) return launch_skinny_tn_nj<T, 4>(P, Z, ldz, out, partial, M, C, st);
    if (cols <= 8) return launch_skinny_tn_nj<T, 8>(P, Z, ldz, out, partial, M, C, st);
    if (cols <= 12) return launch_skinny_tn_nj<T, 12>(P, Z, ldz, out, partial, M, C, st);
    return launch_skinny_tn_nj<T, 16>(P, Z, ldz, out, partial, M, C, st);
}

}  // namespace

extern "C" {

int mos_lora_pack(const mos_lora_sites* s, int dtype, void* A16, void* A16T, void* Bp16, void* BpT, void* stream) {
    MOS_REQUIRE(s && A16 && A16T && Bp16 && BpT, "mos_lora_pack: NULL argument");
    MOS_REQUIRE(s->n_sites >= 1 && s->n_sites <= 4 && s->rank >= 1 && s->n_sites * s->rank <= MOS_LORA_PAD,
                "mos_lora_pack: n_sites*rank must be <= %d (got %d x %d)", MOS_LORA_PAD, s->n_sites, s->rank);
    const int tot = (MOS_LORA_PAD * s->K > s->N * MOS_LORA_PAD) ? MOS_LORA_PAD * s->K : s->N * MOS_LORA_PAD;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MOS_F16)
        hipLaunchKernelGGL((lora_pack_kernel<f16_t>), dim3((tot + 255) / 256), dim3(256), 0, st, *s, (f16_t*)A16,
                           (f16_t*)A16T, (f16_t*)Bp16, (f16_t*)BpT);
    else if (dtype == MOS_BF16)
        hipLaunchKernelGGL((lora_pack_kernel<bf16_t>), dim3((tot + 255) / 256), dim3(256), 0, st, *s, (bf16_t*)A16,
                           (bf16_t*)A16T, (bf16_t*)Bp16, (bf16_t*)BpT);
    else
        return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_lora_pack: dtype %d", dtype);
    return mos_check_launch("lora_pack");
}

int mos_lora_down(const void* x, int64_t ldx, const void* A16, void* t, int M, int K, int dtype, void* stream) {
    MOS_REQUIRE(x && A16 && t, "mos_lora_down: NULL argument");
    MOS_REQUIRE(M > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0, "mos_lora_down: M=%d K=%d ldx=%lld (K, ldx %% 8)", M, K,
                (long long)ldx);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MOS_F16) return launch_skinny_nt<f16_t>(x, ldx, A16, t, M, K, st);
    if (dtype == MOS_BF16) return launch_skinny_nt<bf16_t>(x, ldx, A16, t, M, K, st);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_lora_down: dtype %d", dtype);
}

int mos_lora_linear_fwd(const void* x, int64_t ldx, const void* W, int64_t ldw, const void* t, const void* Bp16,
                        const float* bias, void* y, int64_t ldy, int M, int N, int K, int dtype, void* stream) {
    MOS_REQUIRE(x && W && y, "mos_lora_linear_fwd: NULL argument");
    MOS_REQUIRE((t == nullptr) == (Bp16 == nullptr), "mos_lora_linear_fwd: t and Bp16 must both be set or both NULL");
    MOS_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0 && N % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldy % 4 == 0,
                "mos_lora_linear_fwd: M=%d N=%d K=%d ldx=%lld ldw=%lld ldy=%lld", M, N, K, (long long)ldx,
                (long long)ldw, (long long)ldy);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MOS_F16) return launch_gemm<f16_t>(x, ldx, W, ldw, t, Bp16, bias, y, ldy, M, N, K, st);
    if (dtype == MOS_BF16) return launch_gemm<bf16_t>(x, ldx, W, ldw, t, Bp16, bias, y, ldy, M, N, K, st);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_lora_linear_fwd: dtype %d", dtype);
}

int64_t mos_lora_bwd_workspace_bytes(int M, int N, int K) {
    int64_t best = 0;
    for (int C : {N, K}) {
        const int rpc = tn_rows_per_chunk(M, C);
        const int64_t nchunk = (M + rpc - 1) / rpc;
        const int64_t b = nchunk * MOS_LORA_PAD * C * (int64_t)sizeof(float);
        if (b > best) best = b;
    }
    return best;
}

int mos_lora_linear_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* Wt, int64_t ldwt,
                        const void* t, const void* A16T, const void* BpT, void* dt, void* dx, int64_t lddx,
                        float* dA16, float* dBpT, void* ws, int M, int N, int K, int lora_cols, int dtype, void* stream) {
    MOS_REQUIRE(dy, "mos_lora_linear_bwd: NULL dy");
    MOS_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0 && N % 8 == 0 && lddy % 8 == 0,
                "mos_lora_linear_bwd: M=%d N=%d K=%d lddy=%lld", M, N, K, (long long)lddy);
    const bool lora = (BpT != nullptr);
    MOS_REQUIRE(!lora || (dt && A16T && t && x && ws), "mos_lora_linear_bwd: LoRA path needs dt, A16T, t, x, ws");
    MOS_REQUIRE(dx == nullptr || (Wt && lddx % 4 == 0 && ldwt % 8 == 0), "mos_lora_linear_bwd: dx needs Wt");
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    if (dtype != MOS_F16 && dtype != MOS_BF16) return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_lora_linear_bwd: dtype %d", dtype);
    const bool h = (dtype == MOS_F16);
    if (lora) {  // dt = dy . BpT^T
        rc = h ? launch_skinny_nt<f16_t>(dy, lddy, BpT, dt, M, N, st) : launch_skinny_nt<bf16_t>(dy, lddy, BpT, dt, M, N, st);
        if (rc) return rc;
    }
    if (dx) {  // dx = [dy | dt] . [Wt | A16T]^T
        rc = h ? launch_gemm<f16_t>(dy, lddy, Wt, ldwt, lora ? dt : nullptr, lora ? A16T : nullptr, nullptr, dx, lddx, M, K, N, st)
               : launch_gemm<bf16_t>(dy, lddy, Wt, ldwt, lora ? dt : nullptr, lora ? A16T : nullptr, nullptr, dx, lddx, M, K, N, st);
        if (rc) return rc;
    }
    if (lora && dA16) {  // dA16[16,K] = dt^T . x
        MOS_REQUIRE(ldx % 8 == 0, "mos_lora_linear_bwd: ldx %% 8");
        rc = h ? launch_skinny_tn<f16_t>(dt, x, ldx, dA16, (float*)ws, M, K, lora_cols, st)
               : launch_skinny_tn<bf16_t>(dt, x, ldx, dA16, (float*)ws, M, K, lora_cols, st);
        if (rc) return rc;
    }
    if (lora && dBpT) {  // dBpT[16,N] = t^T . dy
        rc = h ? launch_skinny_tn<f16_t>(t, dy, lddy, dBpT, (float*)ws, M, N, lora_cols, st)
               : launch_skinny_tn<bf16_t>(t, dy, lddy, dBpT, (float*)ws, M, N, lora_cols, st);
        if (rc) return rc;
    }
    return MOS_OK;
}

}  // extern "C"

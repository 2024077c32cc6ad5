// mos_gram.hip — gradient-fusion least squares on the Gram form (gfx950).
//
// Reference: gradient_fusion.py:22-35 (chunk_compute_mse) evaluated inside the L-BFGS closure
// (:62-76): every function evaluation re-uploads all of X,Y in 5000-row chunks and runs 2 GEMMs.
// Since  L(W) = mean((X W^T - Y)^2) = (tr(W G W^T) - 2 tr(W P^T) + c) / (n*Cout)  with
// G = X^T X, P = Y^T X, c = sum(Y^2), the data is streamed ONCE (mos_gram_accumulate, MFMA, fp32
// partials per row-chunk combined in fp64) and each closure costs one small fp64 product W.G
// (mos_lsq_loss_grad_gram).
//
// Gram kernel: out[(Cout+Cin), Cin] = [Y | X]^T X.  The contraction index (sample n) is the slow
// index of both operands, so 64-sample tiles are transposed into LDS ([column][n], packed pairs,
// conflict-free ds_write_b32) and consumed as K-contiguous MFMA operands (32x32x16, ds_read_b128).
#include <cstdio>
#include <cstdlib>
#include "mos_common.h"

namespace {

constexpr int GT = 128;       // output tile (rows of [Y|X]^T) x (cols of X)
constexpr int GN = 64;        // samples per LDS tile
constexpr int GTS = GN + 8;   // transposed row stride: 144 B = 16*9 -> b128 reads conflict-free

// transposed staging of columns [c0, c0+128) of Z (row-major [n][ldz]) for samples [n0, n0+64)
template <typename T, bool SUMSQ>
__device__ __forceinline__ float stage_cols_transposed(T* ldsT, const T* Z, int64_t ldz, int c0, int C, int64_t n0,
                                                        int64_t nend, int tid) {
    float ss = 0.f;
    for (int it = tid; it < 32 * (GT / 8); it += 256) {
        const int p = it & 31, cc = it >> 5;
        const int64_t r0 = n0 + 2 * p, r1 = r0 + 1;
        const int col = c0 + cc * 8;
        const bool cv = col < C;  // C % 8 == 0
        const u32x4 v0 = (cv && r0 < nend) ? ld16(Z + r0 * ldz + col) : u32x4{0, 0, 0, 0};
        const u32x4 v1 = (cv && r1 < nend) ? ld16(Z + r1 * ldz + col) : u32x4{0, 0, 0, 0};
        if constexpr (SUMSQ) {
            const typename MT<T>::v8 a = as_v8<T>(v0), b = as_v8<T>(v1);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float x = (float)a[e], y = (float)b[e]; ss += x * x + y * y; }
        }
        uint32_t* dst = reinterpret_cast<uint32_t*>(ldsT + (cc * 8) * GTS + 2 * p);
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i * (GTS / 2)] = half_of(v0, i) | (half_of(v1, i) << 16);
    }
    return ss;
}

// grid: (col blocks of X, row blocks of [Y|X], chunks of n)
template <typename T>
__global__ __launch_bounds__(256) void gram_kernel(const T* __restrict__ X, int64_t ldx, const T* __restrict__ Y,
                                                   int64_t ldy, int64_t n, int Cin, int Cout, int64_t rows_per_chunk,
                                                   float* __restrict__ partial, double* __restrict__ csum) {
    typedef typename MT<T>::v8 v8;
    __shared__ __attribute__((aligned(16))) T At[GT * GTS];
    __shared__ __attribute__((aligned(16))) T Bt[GT * GTS];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int j0 = blockIdx.x * GT;  // column of X
    const int i0 = blockIdx.y * GT;  // row of the stacked output: [0,Cout) -> Y columns, then X columns
    const int nYb = (Cout + GT - 1) / GT;  // row blocks that belong to Y (Cout padded to blocks)
    const bool isY = (int)blockIdx.y < nYb;
    const T* Z = isY ? Y : X;
    const int64_t ldz = isY ? ldy : ldx;
    const int zc0 = isY ? i0 : (i0 - nYb * GT);
    const int ZC = isY ? Cout : Cin;
    const int64_t nb = (int64_t)blockIdx.z * rows_per_chunk;
    const int64_t ne = min(nb + rows_per_chunk, n);
    const bool do_ss = isY && blockIdx.x == 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float ss = 0.f;
    for (int64_t n0 = nb; n0 < ne; n0 += GN) {
        __syncthreads();
        if (do_ss) ss += stage_cols_transposed<T, true>(At, Z, ldz, zc0, ZC, n0, ne, tid);
        else stage_cols_transposed<T, false>(At, Z, ldz, zc0, ZC, n0, ne, tid);
        stage_cols_transposed<T, false>(Bt, X, ldx, j0, Cin, n0, ne, tid);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < GN / 16; ++ks) {
            v8 af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = as_v8<T>(ld16(At + (wr * 64 + a * 32 + l31) * GTS + ks * 16 + hh * 8));
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = as_v8<T>(ld16(Bt + (wc * 64 + b * 32 + l31) * GTS + ks * 16 + hh * 8));
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = MT<T>::mfma32(af[a], bf[b], acc[a][b]);
        }
    }
    // partial[chunk][row][col], rows indexed in the PADDED stacked space (nYb*GT + Cin rows)
    const int Rtot = nYb * GT + ((Cin + GT - 1) / GT) * GT;
    float* pz = partial + (int64_t)blockIdx.z * Rtot * Cin;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = j0 + wc * 64 + b * 32 + l31;
            if (col < Cin) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    pz[(int64_t)row * Cin + col] = acc[a][b][r];
                }
            }
        }
    if (do_ss) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        if (tid == 0) atomicAdd(csum, (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3]);
    }
}

// G/P += sum over chunks of the fp32 partials, combined in fp64.
__global__ void gram_reduce_kernel(const float* __restrict__ partial, int nchunk, int Cin, int Cout, int nYb,
                                   int Rtot, double* __restrict__ G, double* __restrict__ P) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tot = (int64_t)(Cout + Cin) * Cin;
    if (idx >= tot) return;
    const int i = idx / Cin, j = idx - (int64_t)i * Cin;
    const int prow = (i < Cout) ? i : (nYb * GT + (i - Cout));
    double s = 0.0;
    for (int ch = 0; ch < nchunk; ++ch) s += (double)partial[((int64_t)ch * Rtot + prow) * Cin + j];
    if (i < Cout) P[(int64_t)i * Cin + j] += s;
    else G[(int64_t)(i - Cout) * Cin + j] += s;
}

// R = W.G - P (fp64, 64 x 64 tile per block); grad = 2R/nm ; block partial of sum((R - P) o W).
// On the fp64 matrix cores (v_mfma_f64_16x16x4_f64: D[i][j] += sum_k A[i][k] B[k][j], lane l supplies
// A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; D: lane l holds rows (l >> 4) + 4 * reg, column l & 15 -- NOT the f32
// row map). 64 x 64 block tile, 4 waves of 32 x 32 (2 x 2 MFMA tiles), W and G staged through LDS in 16-deep K slabs (the fp64
// VALU form of rounds 1-3, 4 x 4 outputs per thread, was kept as an A/B switch until round 5 and is gone). The L-BFGS closure of gradient fusion is this kernel ~45 k times per 14-concept job (VALU form:
// 235 us per call at 768 x 768, 3.9 TFLOP/s); the fp64 MFMA peak is ~79 TFLOP/s.
typedef __attribute__((ext_vector_type(4))) double f64x4;
__global__ __launch_bounds__(256) void lsq_grad_mfma_kernel(const double* __restrict__ W, const double* __restrict__ G,
                                                            const double* __restrict__ P, double inv_nm, int Cout, int Cin,
                                                            double* __restrict__ grad, double* __restrict__ blk_partial) {
    __shared__ double Ws[16][64 + 1];
    __shared__ double Gs[16][64 + 1];
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l15 = lane & 15, lg = lane >> 4;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    // the next 16-deep slab is fetched into registers while the MFMAs of the current one run (first version: load -> LDS ->
    // sync -> MFMA in sequence paid a memory round trip per slab: 160 us at 768 x 768, profiles/r04_lsq_fp64_mfma_vs_valu_first.txt)
    double wreg[4], greg[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q;
            const int kk = e & 15, ii = e >> 4;  // W tile: rows i (64) x k (16), k contiguous in memory
            const int gi = i0 + ii, gk = k0 + kk;
            wreg[q] = (gi < Cout && gk < Cin) ? W[(int64_t)gi * Cin + gk] : 0.0;
            const int jj = e & 63, k2 = e >> 6;  // G tile: rows k (16) x cols j (64), j contiguous
            const int gj = j0 + jj, gk2 = k0 + k2;
            greg[q] = (gj < Cin && gk2 < Cin) ? G[(int64_t)gk2 * Cin + gj] : 0.0;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < Cin; k0 += 16) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q;
            Ws[e & 15][e >> 4] = wreg[q];
            Gs[e >> 6][e & 63] = greg[q];
        }
        __syncthreads();
        if (k0 + 16 < Cin) fetch(k0 + 16);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            double af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = Ws[k4 * 4 + lg][wr * 32 + t * 16 + l15];
                bf[t] = Gs[k4 * 4 + lg][wc * 32 + t * 16 + l15];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    }
    double part = 0.0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = i0 + wr * 32 + a * 16 + lg + 4 * r, gj = j0 + wc * 32 + b * 16 + l15;
                if (gi < Cout && gj < Cin) {
                    const int64_t o = (int64_t)gi * Cin + gj;
                    const double pv = P[o], w = W[o];
                    const double rr = acc[a][b][r] - pv;
                    grad[o] = 2.0 * rr * inv_nm;
                    part += (rr - pv) * w;
                }
            }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    if (tid == 0) blk_partial[blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void lsq_loss_finalize_kernel(const double* __restrict__ blk_partial, int nblk, const double* __restrict__ c,
                                         double inv_nm, double* __restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nblk; ++i) s += blk_partial[i];  // fixed order: deterministic
        *loss = (s + *c) * inv_nm;
    }
}

inline int64_t gram_rows_per_chunk(int64_t n, int Cin, int Cout) {
    const int64_t tiles = (int64_t)((Cin + GT - 1) / GT) * (((Cout + GT - 1) / GT) + ((Cin + GT - 1) / GT));
    int64_t nchunk = (1024 + tiles - 1) / tiles;
    const int64_t maxchunk = (n + 4 * GN - 1) / (4 * GN);
    if (nchunk > maxchunk) nchunk = maxchunk;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > 64) nchunk = 64;
    int64_t rpc = (n + nchunk - 1) / nchunk;
    rpc = (rpc + GN - 1) / GN * GN;
    return rpc;
}

}  // namespace

extern "C" {

int64_t mos_gram_workspace_bytes(int64_t n, int Cin, int Cout) {
    if (n <= 0 || Cin <= 0 || Cout <= 0) return 0;
    const int64_t rpc = gram_rows_per_chunk(n, Cin, Cout);
    const int64_t nchunk = (n + rpc - 1) / rpc;
    const int64_t Rtot = (int64_t)((Cout + GT - 1) / GT) * GT + (int64_t)((Cin + GT - 1) / GT) * GT;
    return nchunk * Rtot * Cin * (int64_t)sizeof(float);
}

int mos_gram_accumulate(const void* X, int64_t ldx, const void* Y, int64_t ldy, int64_t n, int Cin, int Cout,
                        int dtype, double* G, double* P, double* c, void* ws, void* stream) {
    MOS_REQUIRE(X && Y && G && P && c && ws, "mos_gram_accumulate: NULL argument");
    MOS_REQUIRE(n > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0,
                "mos_gram_accumulate: n=%lld Cin=%d Cout=%d ldx=%lld ldy=%lld (channels and strides %% 8)",
                (long long)n, Cin, Cout, (long long)ldx, (long long)ldy);
    hipStream_t st = (hipStream_t)stream;
    const int64_t rpc = gram_rows_per_chunk(n, Cin, Cout);
    const int nchunk = (int)((n + rpc - 1) / rpc);
    const int nYb = (Cout + GT - 1) / GT, nXb = (Cin + GT - 1) / GT;
    const int Rtot = (nYb + nXb) * GT;
    dim3 grid(nXb, nYb + nXb, nchunk);
    char key[96];
    snprintf(key, sizeof(key), "n%lld Cin%d Cout%d", (long long)n, Cin, Cout);
    MosProfScope prof(st, "gram", key, 2.0 * (double)n * Cin * ((double)Cin + Cout), 2.0 * (double)n * ((double)Cin + Cout));
    if (dtype == MOS_F16)
        hipLaunchKernelGGL((gram_kernel<f16_t>), grid, dim3(256), 0, st, (const f16_t*)X, ldx, (const f16_t*)Y, ldy, n,
                           Cin, Cout, rpc, (float*)ws, c);
    else if (dtype == MOS_BF16)
        hipLaunchKernelGGL((gram_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)X, ldx, (const bf16_t*)Y, ldy,
                           n, Cin, Cout, rpc, (float*)ws, c);
    else
        return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_gram_accumulate: dtype %d", dtype);
    int rc = mos_check_launch("gram");
    if (rc) return rc;
    const int64_t tot = (int64_t)(Cout + Cin) * Cin;
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float*)ws,
                       nchunk, Cin, Cout, nYb, Rtot, G, P);
    return mos_check_launch("gram_reduce");
}

int64_t mos_lsq_workspace_bytes(int Cout, int Cin) {
    return (int64_t)((Cout + 63) / 64) * ((Cin + 63) / 64) * (int64_t)sizeof(double);
}

int mos_lsq_loss_grad_gram(const double* W, const double* G, const double* P, const double* c, double n_times_cout,
                           int Cout, int Cin, double* loss, double* grad, void* ws, void* stream) {
    MOS_REQUIRE(W && G && P && c && loss && grad && ws, "mos_lsq_loss_grad_gram: NULL argument");
    MOS_REQUIRE(Cout > 0 && Cin > 0 && n_times_cout > 0, "mos_lsq_loss_grad_gram: Cout=%d Cin=%d", Cout, Cin);
    hipStream_t st = (hipStream_t)stream;
    const double inv = 1.0 / n_times_cout;
    dim3 grid((Cin + 63) / 64, (Cout + 63) / 64);
    char key[64];
    snprintf(key, sizeof(key), "Cout%d Cin%d", Cout, Cin);
    MosProfScope prof(st, "lsq_loss_grad", key, 2.0 * Cout * (double)Cin * Cin, 8.0 * (3.0 * Cout * Cin + (double)Cin * Cin));
    hipLaunchKernelGGL(lsq_grad_mfma_kernel, grid, dim3(256), 0, st, W, G, P, inv, Cout, Cin, grad, (double*)ws);
    int rc = mos_check_launch("lsq_grad");
    if (rc) return rc;
    hipLaunchKernelGGL(lsq_loss_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, (int)(grid.x * grid.y), c,
                       inv, loss);
    return mos_check_launch("lsq_loss_finalize");
}

}  // extern "C"

// mos_gemm.hip — LoRA-augmented linear layers for gfx950.
//
// Replaces LoRALinearLayer.forward (reference mixofshow/models/edlora.py:244-246):
//     y = orig(x) + alpha * lora_up(lora_down(x))
// which the reference runs as 3 GEMM launches (base, K=..->4, 4->N) + scale + add per site.
// Here ONE kernel per direction:
//   forward : y = x.W^T + (x.A16^T).Bp16^T + b.  The down projection t = x.A16^T rides in the same K loop as 16 extra
//             "weight rows" (A16 staged next to the W tile); its fp32 accumulator, rounded to half, IS the B operand of
//             the rank-16 epilogue MFMA (same lane layout), so t never leaves registers; the n-tile-0 blocks also store
//             it for the backward.
//   backward: dx = dy.Wt^T + (dy.BpT^T).A16T^T with dt = dy.BpT^T produced the same way (W frozen: no dW), then
//   lora_grad: dA = dt^T x and dB = t^T dy, both in ONE token-reduction launch + one ordered (deterministic) final sum that
//             writes straight into the fp32 parameter gradients (alpha folded, per-site slices, optional accumulate) —
//             no per-parameter glue kernels on the host side.
//
// MFMA conventions (v_mfma_f32_16x16x32_{f16,bf16}; D[i][j] = sum_k A[i][k] B[k][j]):
//   lane l supplies A[i = l&15][k = 8*(l>>4) .. +8] and B[k = 8*(l>>4) .. +8][j = l&15];
//   D: lane l holds D[i = 4*(l>>4) + reg][j = l&15], reg = 0..3.
// We feed A <- weight rows (n), B <- activation rows (m), so every lane ends up with 4
// CONSECUTIVE output features of one token: an 8-byte store.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "mos_common.h"

// tuning knobs (tools/build_variant.sh)
#ifndef MOS_TN_REDUCE_UNROLL
#define MOS_TN_REDUCE_UNROLL 0
#endif
#ifndef MOS_TN_TARGET_WG
#define MOS_TN_TARGET_WG 512      // workgroups the legacy LoRA-gradient kernel aims at
#endif
#ifndef MOS_GRAD_TARGET_WG
// workgroups the token reduction of ONE LoRA group aims at. 1024 while every group was its own launch (rounds 2-4); since the groups
// of a backward pass share one launch per rank class (mos_lora_grad_all, ~15 k blocks) fewer, longer blocks per group win: fewer
// partial sums to write and to sum, fewer re-reads of the 16-wide t / dt rows -- same box, per step: 0.81 ms at 1024, 0.61 at 512,
// 0.51 at 256 (3.0 TB/s; profiles/r05c7_ab_same_box_lora_grad_tuning.txt). Round 6, same box, the three launches of a step:
// 0.598 ms at 512, 0.505 at 256, 0.456 at 128, 0.462 at 96, 0.450 at 64, 0.464 at 32 (the rank-12 class of the text encoder:
// 311 -> 270 us; profiles/r06c26_lora_grad_block_target.txt)
#define MOS_GRAD_TARGET_WG 128
#endif
#ifndef MOS_GEMM_DEEP_MAX_WG
#define MOS_GEMM_DEEP_MAX_WG 96   // GEMMs with at most this many workgroups use the deep-stage variants (measured: a win only
#endif                            // for M = 256; at 160+ workgroups the larger LDS footprint costs more than it hides); 0 = off


namespace {

constexpr int GEMM_BK = 64;                   // default K depth of a stage

struct GemmArgs {
    const void* X; const void* W; const void* Taug; const void* Adown; const void* Baug; const float* bias;
    void* Y; void* Tout;
    int64_t ldx, ldw, ldy;
    int M, N, K, mt, nt;
    // epilogue variant (mos_lora_linear_fwd_ex): R = residual [M, N] added to the ROUNDED result (the rounding points of
    // "GEMM, then an add kernel"). (A GEGLU epilogue -- value * gelu(gate) from interleaved weight rows -- was built in round 4,
    // measured no faster than hipBLASLt + the geglu kernel at the sampling sizes, and removed in round 5.)
    const void* R; int64_t ldr;
};

// Y[M,N] = X[M,K].W[N,K]^T (+ t.Baug^T) (+ bias), t = X.Adown^T computed in the same K loop (FUSED) or read from Taug.
//   * Tiles are fetched with buffer loads whose descriptors end at the last valid row: rows past M / N read as zeros in
//     hardware (no clamps, no selects in the prefetch); KTAIL (K % 64 != 0, not an SD-1.5 shape) adds a column select.
//   * Block -> tile map is XCD-aware: the nt blocks that share one X row-tile get consecutive slots on ONE XCD (workgroup
//     id % 8), so X is pulled into a single L2 once instead of being re-fetched by up to nt XCDs.
//   * BK = K depth of one pipeline stage. The MFMA work of a 64-deep stage is ~80 ns per wave against ~0.7 us of memory
//     latency, so with few workgroups the K loop is a chain of exposed latencies. Deeper stages (BK 256 / 128: 4x / 2x the
//     bytes in flight per block) were measured: +30 % SLOWER at 160-460 workgroups (one block per CU left), a win only
//     below ~100 workgroups (M = 256). hipBLASLt sits at the same level on these shapes (profiles/r02_kernel_bench_gemm_vs_hipblaslt.txt).
//   * The output tile goes through LDS so that every store instruction writes whole 16 B chunks of consecutive features
//     (a lane's accumulators are 4 features of one token: written directly, a wave's store touches 16 rows x 32 B).
template <typename T, int BM, int BN, int BK, bool FUSED, bool KTAIL, bool DMA, int NS>
__global__ __launch_bounds__(256) void gemm_lora_kernel(const GemmArgs a) {
    static_assert(!DMA || (BK == 64 && !KTAIL), "the LDS-DMA tile image is [row][8 swizzled 16 B chunks]");
    static_assert(NS == 2 || DMA, "more than two stages: LDS-DMA ring only");
    typedef typename MT<T>::v8 v8;
    typedef typename MT<T>::v4 v4;
    constexpr int MI = BM / 32;             // 16-row m sub-tiles per wave (wave tile = BM/2 x BN/2)
    constexpr int NJ = BN / 32;             // 16-wide n sub-tiles per wave
    constexpr int CPR = BK / 8;             // 16 B chunks per tile row
    constexpr int XCH = BM * CPR / 256;     // chunks of the X tile per thread
    constexpr int WCH = BN * CPR / 256;
    constexpr int ACH = (16 * CPR + 255) / 256;
    // LDS row stride. Register staging: (BK/8 + 1) 16-byte slots, odd -> conflict-free b128 reads. LDS-DMA: the image is
    // lane-linear, rows cannot be padded; chunk c of row r sits at slot c ^ (r & 7) instead (same property, see mos_conv.hip).
    constexpr int LS = DMA ? BK : BK + 8;
    constexpr int CS = BN + 8;              // output staging stride

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Xs = reinterpret_cast<T*>(smem_raw);                       // [NS][BM][LS]
    T* Ws = Xs + NS * BM * LS;                                    // [NS][BN][LS]
    T* As = Ws + NS * BN * LS;                                    // [NS][16][LS]   (FUSED)

    const int w = blockIdx.x;
    const int slot = w >> 3;
    const int n_tile = slot % a.nt;
    const int m_tile = (slot / a.nt) * 8 + (w & 7);
    if (m_tile >= a.mt) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably uniform: LDS-DMA destinations (M0) become SALU values
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = n_tile * BN;
    const int m0 = m_tile * BM;
    const int M = a.M, N = a.N, K = a.K;

    const rsrc_t xsrc = make_rsrc(a.X, (uint32_t)((((int64_t)M - 1) * a.ldx + K) * (int64_t)sizeof(T)));
    const rsrc_t wsrc = make_rsrc(a.W, (uint32_t)((((int64_t)N - 1) * a.ldw + K) * (int64_t)sizeof(T)));
    const rsrc_t asrc = make_rsrc(FUSED ? a.Adown : a.W, (uint32_t)(FUSED ? 16 * (int64_t)K * sizeof(T) : 16));

    // chunk (16 B) this thread moves in row c / CPR: column c % CPR, or its swizzled partner under LDS-DMA
    auto chunk_col = [](int c) { return DMA ? ((c & 7) ^ ((c >> 3) & 7)) : (c % CPR); };
    int xoff[XCH], woff[WCH], aoff[ACH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int c = tid + 256 * i;
        xoff[i] = (int)((((int64_t)(m0 + c / CPR)) * a.ldx + chunk_col(c) * 8) * (int64_t)sizeof(T));
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int c = tid + 256 * i;
        woff[i] = (int)((((int64_t)(n0 + c / CPR)) * a.ldw + chunk_col(c) * 8) * (int64_t)sizeof(T));
    }
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int c = DMA ? (tid & 127) : min(tid + 256 * i, 16 * CPR - 1);
        aoff[i] = (int)(((int64_t)(c / CPR) * K + chunk_col(c) * 8) * (int64_t)sizeof(T));
    }

    f32x4 acc[NJ][MI];
    f32x4 acct[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        acct[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    u32x4 xr[XCH], wr[WCH], ar[ACH];
    const int nk = (K + BK - 1) / BK;

    // DMA: tile kt -> LDS buffer buf; wave w's i-th piece = slots (4i + w) * 64 .. The 16 A rows are two pieces: waves 2, 3
    // repeat those of waves 0, 1 (same bytes, same place) so that every wave has the same number of DMAs in flight, which
    // is what the ring's counted wait needs. Ring tiles past the end of K are issued out of range (zeros, no traffic).
    constexpr int OOB = 0x7FFFFF00;
    auto dma_tile = [&](int kt, int buf) {
        const bool live = (NS == 2) || kt < nk;
        const int kb = kt * BK * (int)sizeof(T);
        T* xs = Xs + buf * BM * LS + wave * 512;
        T* ws = Ws + buf * BN * LS + wave * 512;
#pragma unroll
        for (int i = 0; i < XCH; ++i) dma16(xsrc, xs + i * 2048, live ? xoff[i] + kb : OOB);
#pragma unroll
        for (int i = 0; i < WCH; ++i) dma16(wsrc, ws + i * 2048, live ? woff[i] + kb : OOB);
        if constexpr (FUSED) dma16(asrc, As + buf * 16 * LS + (wave & 1) * 512, live ? aoff[0] + kb : OOB);
    };
    auto load_tile = [&](int kt) {
        const int kb = kt * BK * (int)sizeof(T);
#pragma unroll
        for (int i = 0; i < XCH; ++i) xr[i] = ldbuf16(xsrc, xoff[i] + kb);
#pragma unroll
        for (int i = 0; i < WCH; ++i) wr[i] = ldbuf16(wsrc, woff[i] + kb);
        if constexpr (FUSED) {
#pragma unroll
            for (int i = 0; i < ACH; ++i)
                if (tid + 256 * i < 16 * CPR) ar[i] = ldbuf16(asrc, aoff[i] + kb);
        }
        if constexpr (KTAIL) {      // columns past K inside a valid row belong to the next row: mask them (BK = 64 here)
            const bool ok = (kt * BK + (tid % CPR) * 8) < K;
            if (!ok) {
#pragma unroll
                for (int i = 0; i < XCH; ++i) xr[i] = u32x4{0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < WCH; ++i) wr[i] = u32x4{0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < ACH; ++i) ar[i] = u32x4{0, 0, 0, 0};
            }
        }
    };
    auto store_tile = [&](int buf) {
        T* xs = Xs + buf * BM * LS;
        T* ws = Ws + buf * BN * LS;
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int c = tid + 256 * i;
            st16(xs + (c / CPR) * LS + (c % CPR) * 8, xr[i]);
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int c = tid + 256 * i;
            st16(ws + (c / CPR) * LS + (c % CPR) * 8, wr[i]);
        }
        if constexpr (FUSED) {
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                const int c = tid + 256 * i;
                if (c < 16 * CPR) st16(As + buf * 16 * LS + (c / CPR) * LS + (c % CPR) * 8, ar[i]);
            }
        }
    };

    // fragment chunk of k-step kk: lg + 4 kk, at its swizzled slot under DMA (every fragment row r has r & 7 == l15 & 7)
    const int fc0 = DMA ? ((lg ^ (l15 & 7)) * 8) : lg * 8;
    auto compute_tile = [&](int buf) {
        const T* xs = Xs + buf * BM * LS + (wm * (BM / 2) + l15) * LS;
        const T* ws = Ws + buf * BN * LS + (wn * (BN / 2) + l15) * LS;
        const T* as = As + buf * 16 * LS + l15 * LS;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            const int fc = DMA ? (fc0 ^ (kk * 32)) : (fc0 + kk * 32);
            v8 bfrag[MI], afrag[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) bfrag[i] = as_v8<T>(ld16(xs + i * 16 * LS + fc));
#pragma unroll
            for (int j = 0; j < NJ; ++j) afrag[j] = as_v8<T>(ld16(ws + j * 16 * LS + fc));
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[j][i] = MT<T>::mfma16(afrag[j], bfrag[i], acc[j][i]);
            if constexpr (FUSED) {
                const v8 at = as_v8<T>(ld16(as + fc));
#pragma unroll
                for (int i = 0; i < MI; ++i) acct[i] = MT<T>::mfma16(at, bfrag[i], acct[i]);
            }
        }
    };

    if constexpr (DMA && NS > 2) {
        // ring: NS - 1 tiles in flight; tile kt is complete for this wave once at most the NS - 2 younger ones are
        // outstanding, and for the block past the barrier, which also retires every read of tile kt - 1 (the next target)
        constexpr int L = XCH + WCH + (FUSED ? 1 : 0);
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) dma_tile(s, s);
        int cur = 0, nxt = NS - 1;
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NS - 2) * L) : "memory");
            dma_tile(kt + NS - 1, nxt);
            compute_tile(cur);
            cur = (cur + 1 == NS) ? 0 : cur + 1;
            nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the zero tiles issued past the end
        __syncthreads();
    } else {
        if constexpr (DMA) {
            dma_tile(0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            load_tile(0);
            store_tile(0);
        }
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) {                   // next tile in flight under the MFMAs below
                if constexpr (DMA) dma_tile(kt + 1, cur ^ 1);
                else load_tile(kt + 1);
            }
            compute_tile(cur);
            if constexpr (DMA) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (kt + 1 < nk) store_tile(cur ^ 1);
            }
            __syncthreads();
        }
    }

    // Rank augmentation: y += t[M,16] . Baug[N,16]^T, one K=16 MFMA per tile
    // (v_mfma_f32_16x16x16: lane supplies A[i=l&15][k=4*(l>>4)..+4], B[k][j=l&15]).
    // FUSED: acct[i] holds t[m = l15][r = 4*lg .. +3] of m sub-tile i — exactly that B operand once rounded to T.
    if (FUSED || a.Taug != nullptr) {
        v4 tb[MI], ba[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + wm * (BM / 2) + i * 16 + l15;
            if constexpr (FUSED) {
                const u32x2 tp = pack4<T>(acct[i][0], acct[i][1], acct[i][2], acct[i][3]);
                tb[i] = __builtin_bit_cast(v4, tp);
                if (a.Tout != nullptr && n_tile == 0 && wn == 0 && m < M)
                    st8(reinterpret_cast<T*>(a.Tout) + (int64_t)m * MOS_LORA_PAD + lg * 4, tp);
            } else {
                const int mc = min(m, M - 1);
                tb[i] = __builtin_bit_cast(v4, ld8(reinterpret_cast<const T*>(a.Taug) + (int64_t)mc * MOS_LORA_PAD + lg * 4));
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = min(n0 + wn * (BN / 2) + j * 16 + l15, N - 1);
            ba[j] = __builtin_bit_cast(v4, ld8(reinterpret_cast<const T*>(a.Baug) + (int64_t)n * MOS_LORA_PAD + lg * 4));
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[j][i] = MT<T>::mfma16k16(ba[j], tb[i], acc[j][i]);
    }

    // Epilogue: bias, round, stage the BM x BN tile in LDS (the K loop's final barrier has retired every tile read), then
    // row-major 16 B stores.
    T* Cs = reinterpret_cast<T*>(smem_raw);                       // [BM][CS]
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int nl = wn * (BN / 2) + j * 16 + lg * 4;
        const int nb = min(n0 + nl, N - 4);
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        if (a.bias != nullptr) {
            b0 = a.bias[nb]; b1 = a.bias[nb + 1]; b2 = a.bias[nb + 2]; b3 = a.bias[nb + 3];
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ml = wm * (BM / 2) + i * 16 + l15;
            const f32x4 v = acc[j][i];
            st8(Cs + ml * CS + nl, pack4<T>(v[0] + b0, v[1] + b1, v[2] + b2, v[3] + b3));
        }
    }
    __syncthreads();
    T* Y = reinterpret_cast<T*>(a.Y);
    constexpr int OCH = BM * (BN / 8) / 256;
    const T* Rr = reinterpret_cast<const T*>(a.R);
#pragma unroll
    for (int i = 0; i < OCH; ++i) {
        const int c = tid + 256 * i;
        const int row = c / (BN / 8), col = (c % (BN / 8)) * 8;
        if (m0 + row < M && n0 + col < N) {                       // N % 8 == 0: a chunk is entirely inside or outside
            u32x4 out = ld16(Cs + row * CS + col);
            if (Rr != nullptr) {
                const v8 cv = as_v8<T>(out), rv = as_v8<T>(ld16(Rr + (int64_t)(m0 + row) * a.ldr + n0 + col));
                v8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (T)((float)cv[e] + (float)rv[e]);
                out = from_v8<T>(o);
            }
            st16(Y + (int64_t)(m0 + row) * a.ldy + n0 + col, out);
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

template <typename T, int BM, int BN, int BK, bool FUSED, bool KTAIL, bool DMA, int NS>
int launch_gemm_cfg(const GemmArgs& a, hipStream_t st) {
    size_t lds = (size_t)NS * (BM + BN + (FUSED ? 16 : 0)) * (DMA ? BK : BK + 8) * sizeof(T);
    const size_t stage = (size_t)BM * (BN + 8) * sizeof(T);
    if (stage > lds) lds = stage;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_lora_kernel<T, BM, BN, BK, FUSED, KTAIL, DMA, NS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    GemmArgs b = a;
    b.mt = (a.M + BM - 1) / BM;
    b.nt = (a.N + BN - 1) / BN;
    const int mt8 = (b.mt + 7) / 8 * 8;
    hipLaunchKernelGGL((gemm_lora_kernel<T, BM, BN, BK, FUSED, KTAIL, DMA, NS>), dim3(mt8 * b.nt), dim3(256), lds, st, b);
    return mos_check_launch("gemm_lora");
}

// Tile choice: 128-wide n tiles when N allows (halves the X re-reads and the relative cost of the fused down
// projection), 128-row m tiles when that still yields >= 384 workgroups, otherwise 64-row / 64-wide tiles so that the
// small-M levels (M = 1024, 256) spread over the 256 CUs. With few workgroups per CU the K loop is latency-bound: those
// configurations take deep stages (BK 256 / 128), see the kernel comment.
template <typename T, bool FUSED>
int launch_gemm_t(const GemmArgs& a, hipStream_t st) {
    if (a.K % GEMM_BK != 0) return launch_gemm_cfg<T, 64, 64, 64, FUSED, true, false, 2>(a, st);
    // (256-row workgroup tiles -- 256 x 128 and 256 x 64, one wave per SIMD -- were measured in round 5 and lost on every shape:
    //  M4928 N2304 K768 32 -> 42 us, M4096 N640 K640 13 -> 20 us, M16384 N960 K320 26 -> 26 us; profiles/r05c1_wide_tiles.txt)
    int bn = (a.N % 128 == 0) ? 128 : 64, bm = 128;
    auto tiles = [&](int m, int n) { return (int64_t)((a.M + m - 1) / m) * ((a.N + n - 1) / n); };
    if (tiles(bm, bn) < 384) bm = 64;
    if (tiles(bm, bn) < 256 && bn == 128) bn = 64;
    if (bm == 128 && bn == 128) return launch_gemm_cfg<T, 128, 128, 64, FUSED, false, true, 2>(a, st);
    if (bm == 128) return launch_gemm_cfg<T, 128, 64, 64, FUSED, false, true, 2>(a, st);
    const bool ring = tiles(bm, bn) <= 640;      // grids of <= 640 workgroups: latency-bound K loop, DMA ring instead of the double buffer
    if (ring) {
        if (bn == 128) return launch_gemm_cfg<T, 64, 128, 64, FUSED, false, true, 3>(a, st);
        return launch_gemm_cfg<T, 64, 64, 64, FUSED, false, true, 4>(a, st);
    }
    const bool deep = tiles(bm, bn) <= MOS_GEMM_DEEP_MAX_WG;
    if (bn == 128) {
        if (deep && a.K % 128 == 0) return launch_gemm_cfg<T, 64, 128, 128, FUSED, false, false, 2>(a, st);
        return launch_gemm_cfg<T, 64, 128, 64, FUSED, false, true, 2>(a, st);
    }
    if (deep && a.K % 256 == 0) return launch_gemm_cfg<T, 64, 64, 256, FUSED, false, false, 2>(a, st);
    if (deep && a.K % 128 == 0) return launch_gemm_cfg<T, 64, 64, 128, FUSED, false, false, 2>(a, st);
    return launch_gemm_cfg<T, 64, 64, 64, FUSED, false, true, 2>(a, st);
}

// adown != NULL: fused down projection (t computed in-kernel, stored to tout if given); else t (may be NULL) is read.
template <typename T>
int launch_gemm(const void* X, int64_t ldx, const void* W, int64_t ldw, const void* t, const void* adown, const void* Bp,
                const float* bias, void* Y, int64_t ldy, void* tout, int M, int N, int K, hipStream_t st,
                const void* residual = nullptr, int64_t ldr = 0) {
    char key[112];
    const bool lora = (t != nullptr) || (adown != nullptr);
    snprintf(key, sizeof(key), "%s M%d N%d K%d%s%s", std::is_same<T, f16_t>::value ? "f16" : "bf16", M, N, K,
             adown ? " +lora(fused)" : (t ? " +lora" : ""), residual ? " +res" : "");
    const double nout = (double)N;
    MosProfScope prof(st, "gemm_nt", key, 2.0 * M * (double)N * (K + (lora ? 16 : 0)) + (adown ? 2.0 * M * 16.0 * K : 0.0),
                      2.0 * ((double)M * K + (double)N * K + (double)M * nout * (residual ? 2.0 : 1.0)));
    GemmArgs a;
    a.X = X; a.W = W; a.Taug = t; a.Adown = adown; a.Baug = Bp; a.bias = bias; a.Y = Y; a.Tout = tout;
    a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.M = M; a.N = N; a.K = K; a.mt = a.nt = 0;
    a.R = residual; a.ldr = ldr;
    return adown ? launch_gemm_t<T, true>(a, st) : launch_gemm_t<T, false>(a, st);
}

// ---- fused LoRA factor gradients -------------------------------------------------------------------------------------
// job 0: dA[j][c] = sum_m dt[m][j] x[m][c]   (c < K)      job 1: dB[j][n] = sum_m t[m][j] dy[m][n]   (n < N)
// grid (nchunk, cbK + cbN): a block reduces `rpc` tokens of one 64-column block into partial[chunk] (both jobs in ONE
// launch); lora_grad_final_kernel then sums the partials in chunk order.
// The token reduction is HBM-bound (x and dy are read once); arithmetic is NJ FMAs per loaded element on the VALU.
struct LoraGradArgs {
    const void* P[2]; const void* Z[2];
    int64_t ldz[2];
    int C[2], cb[2];
    float* partial[2];
    float* raw[2];
    int M, rpc, nchunk;
    mos_lora_grad_out out;
};

// one block of the token reduction: chunk bx of the tokens, column block by (of job 0, then of job 1); `J` is LoraGradArgs or a
// mos_lora_grad_job (same field names). Shared by the per-group launch and the all-groups launch: identical arithmetic.
template <typename T, int NJ, typename J>
__device__ __forceinline__ void lora_grad_block(const J& a, int bx, int by, float (&red)[4][NJ][64]) {
    typedef typename MT<T>::v8 v8;
    typedef typename MT<T>::v4 v4;
    const int tid = threadIdx.x;
    const int job = by >= a.cb[0] ? 1 : 0;
    const int colblk = by - (job ? a.cb[0] : 0);
    const int C = a.C[job];
    const T* P = reinterpret_cast<const T*>(a.P[job]);
    const T* Z = reinterpret_cast<const T*>(a.Z[job]);
    const int64_t ldz = a.ldz[job];
    const int cg = tid & 7, ry = tid >> 3;
    const int c0 = colblk * 64 + cg * 8;
    const int mb = bx * a.rpc;
    const int me = min(mb + a.rpc, a.M);
    float acc[NJ][8];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
    if (c0 < C) {
        constexpr int U = 4;      // (8 measured in round 5: no gain, profiles/r05c7_ab_same_box_lora_grad_tuning.txt)
        for (int m = mb + ry; m < me; m += 32 * U) {
            u32x4 z[U];
            u32x2 p[U][NJ / 4];
#pragma unroll
            for (int u = 0; u < U; ++u) {          // all loads of the unrolled group are issued before any use
                const int mm = min(m + 32 * u, me - 1);
                z[u] = ld16(Z + (int64_t)mm * ldz + c0);
#pragma unroll
                for (int q = 0; q < NJ / 4; ++q) p[u][q] = ld8(P + (int64_t)mm * MOS_LORA_PAD + q * 4);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (m + 32 * u < me) {
                    const v8 zv = as_v8<T>(z[u]);
                    float zf[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) zf[e] = (float)zv[e];
#pragma unroll
                    for (int q = 0; q < NJ / 4; ++q) {
                        const v4 pv = __builtin_bit_cast(v4, p[u][q]);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const float pf = (float)pv[jj];
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[q * 4 + jj][e] += pf * zf[e];
                        }
                    }
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
    float* part = a.partial[job];
    for (int idx = tid; idx < NJ * 64; idx += 256) {
        const int j = idx >> 6, c = idx & 63;
        const int col = colblk * 64 + c;
        if (col < C) part[((int64_t)bx * NJ + j) * C + col] = red[0][j][c] + red[1][j][c] + red[2][j][c] + red[3][j][c];
    }
}

template <typename T, int NJ>
__global__ __launch_bounds__(256) void lora_grad_kernel(const LoraGradArgs a) {
    __shared__ float red[4][NJ][64];
    lora_grad_block<T, NJ>(a, (int)blockIdx.x, (int)blockIdx.y, red);
}

// EVERY deferred group of a backward pass in ONE launch (round 5): the per-group launches (~100 per SD-1.5 step, 7-25 us each:
// small grids on their latency floor, 0.85-1.8 TB/s) become one streaming pass over all x / dy of the step. Block -> job by binary
// search over the block offsets of the table (device memory); inside a job the blocks are numbered chunk-fastest. One launch per
// padded-rank class NJ (4 / 8 / 12 / 16: a merged kernel would run every class at the 264 registers of NJ = 16, one wave per SIMD).
template <typename T, int NJ>
__global__ __launch_bounds__(256) void lora_grad_all_kernel(const mos_lora_grad_job* __restrict__ jobs, int n_jobs) {
    __shared__ float red[4][NJ][64];
    const int bid = (int)blockIdx.x;
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block_begin <= bid) lo = mid; else hi = mid - 1;
    }
    const mos_lora_grad_job j = jobs[lo];
    const int local = bid - j.block_begin;
    lora_grad_block<T, NJ>(j, local % j.nchunk, local / j.nchunk, red);
}

// Final, ORDERED sum over token chunks (deterministic) of both jobs, written as raw 16 x C matrices (legacy API) or straight
// into the per-site fp32 parameter gradients (alpha folded into the up factor, optional accumulate). A separate launch
// on purpose: a kernel boundary makes the partials visible across the 8 XCD-private L2s for free, whereas the
// "last block reduces" idiom needs an agent-scope release/acquire per block = an L2 write-back each (measured: 64-100 us
// per launch instead of ~10).
template <int NJ>
__global__ __launch_bounds__(256) void lora_grad_final_kernel(const LoraGradArgs a) {
    const int job = (int)blockIdx.x >= a.cb[0] ? 1 : 0;
    const int colblk = (int)blockIdx.x - (job ? a.cb[0] : 0);
    const int C = a.C[job];
    const float* part = a.partial[job];
    const int r = a.out.rank;
    for (int idx = threadIdx.x; idx < MOS_LORA_PAD * 64; idx += 256) {
        const int j = idx >> 6, c = idx & 63;
        const int col = colblk * 64 + c;
        if (col >= C) continue;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (j < NJ) {
            const float* pp = part + (int64_t)j * C + col;
            const int64_t cs = (int64_t)NJ * C;
            int ch = 0;
            for (; ch + 4 <= a.nchunk; ch += 4) {   // four independent chains: the loads of a thread are in flight together
                s0 += pp[(ch + 0) * cs];
                s1 += pp[(ch + 1) * cs];
                s2 += pp[(ch + 2) * cs];
                s3 += pp[(ch + 3) * cs];
            }
            for (; ch < a.nchunk; ++ch) s0 += pp[ch * cs];
        }
        const float s = (s0 + s1) + (s2 + s3);
        if (a.raw[job] != nullptr) {
            a.raw[job][(int64_t)j * C + col] = s;      // rows >= NJ are zero
            continue;
        }
        const int g = j / r;
        if (j >= NJ || g >= a.out.n_sites) continue;
        const int jj = j - g * r;
        if (job == 0) {
            float* dst = a.out.down_grad[g];
            if (dst != nullptr) {
                float* q = dst + (int64_t)jj * C + col;
                *q = a.out.accumulate_down[g] ? (*q + s) : s;
            }
        } else {
            float* dst = a.out.up_grad[g];
            const int nn = col - a.out.n_begin[g];
            if (dst != nullptr && nn >= 0 && nn < a.out.n_rows[g]) {
                float* q = dst + (int64_t)nn * r + jj;
                const float v = a.out.alpha[g] * s;
                *q = a.out.accumulate_up[g] ? (*q + v) : v;
            }
        }
    }
}

// The same ordered final sum for EVERY LoRA group of a backward pass in ONE launch (records in device memory, like
// lora_pack_all): ~90 final-sum launches per training step become one. Block -> record by the records' block_begin.
__global__ __launch_bounds__(256) void lora_grad_final_all_kernel(const mos_lora_final_rec* __restrict__ recs, int n_recs) {
    __shared__ mos_lora_final_rec rec;
    __shared__ int which;
    if (threadIdx.x == 0) {
        int w = 0;
        for (int i = 1; i < n_recs; ++i)
            if ((int)blockIdx.x >= recs[i].block_begin) w = i;
        which = w;
    }
    __syncthreads();
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(recs + which);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&rec);
        for (int i = threadIdx.x; i < (int)(sizeof(mos_lora_final_rec) / 4); i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int blk = (int)blockIdx.x - rec.block_begin;
    const int job = blk >= rec.cb[0] ? 1 : 0;
    const int colblk = blk - (job ? rec.cb[0] : 0);
    const int C = rec.C[job], NJ = rec.nj, r = rec.out.rank;
    const float* part = rec.partial[job];
    for (int idx = threadIdx.x; idx < NJ * 64; idx += 256) {
        const int j = idx >> 6, c = idx & 63;
        const int col = colblk * 64 + c;
        if (col >= C) continue;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const float* pp = part + (int64_t)j * C + col;
        const int64_t cs = (int64_t)NJ * C;
        int ch = 0;
        for (; ch + 4 <= rec.nchunk; ch += 4) {
            s0 += pp[(ch + 0) * cs];
            s1 += pp[(ch + 1) * cs];
            s2 += pp[(ch + 2) * cs];
            s3 += pp[(ch + 3) * cs];
        }
        for (; ch < rec.nchunk; ++ch) s0 += pp[ch * cs];
        const float s = (s0 + s1) + (s2 + s3);     // same association as lora_grad_final_kernel: bit-identical results
        const int g = j / r;
        if (g >= rec.out.n_sites) continue;
        const int jj = j - g * r;
        if (job == 0) {
            float* dst = rec.out.down_grad[g];
            if (dst != nullptr) {
                float* q = dst + (int64_t)jj * C + col;
                *q = rec.out.accumulate_down[g] ? (*q + s) : s;
            }
        } else {
            float* dst = rec.out.up_grad[g];
            const int nn = col - rec.out.n_begin[g];
            if (dst != nullptr && nn >= 0 && nn < rec.out.n_rows[g]) {
                float* q = dst + (int64_t)nn * r + jj;
                const float v = rec.out.alpha[g] * s;
                *q = rec.out.accumulate_up[g] ? (*q + v) : v;
            }
        }
    }
}

inline void lora_grad_plan(int M, int N, int K, int* rpc, int* nchunk) {
    const int cbt = (K + 63) / 64 + (N + 63) / 64;
    int nc = (MOS_GRAD_TARGET_WG + cbt - 1) / cbt;
    const int maxc = (M + 63) / 64;
    if (nc > maxc) nc = maxc;
    if (nc < 1) nc = 1;
    int r = (M + nc - 1) / nc;
    r = (r + 31) / 32 * 32;
    *rpc = r;
    *nchunk = (M + r - 1) / r;
}

template <typename T>
int launch_lora_grad(const void* dt, const void* x, int64_t ldx, const void* t, const void* dy, int64_t lddy, float* rawA,
                     float* rawB, const mos_lora_grad_out* out, float* ws, int M, int N, int K, int cols,
                     hipStream_t st, mos_lora_final_rec* defer = nullptr, mos_lora_grad_job* job = nullptr) {
    LoraGradArgs a;
    a.P[0] = dt; a.Z[0] = x; a.ldz[0] = ldx; a.C[0] = K; a.cb[0] = (K + 63) / 64;
    a.P[1] = t; a.Z[1] = dy; a.ldz[1] = lddy; a.C[1] = N; a.cb[1] = (N + 63) / 64;
    a.raw[0] = rawA; a.raw[1] = rawB;
    a.M = M;
    lora_grad_plan(M, N, K, &a.rpc, &a.nchunk);
    const int nj = cols <= 4 ? 4 : cols <= 8 ? 8 : cols <= 12 ? 12 : 16;
    a.partial[0] = ws;
    a.partial[1] = ws + (int64_t)a.nchunk * nj * K;
    if (out != nullptr) a.out = *out; else { mos_lora_grad_out z = {}; z.rank = 1; a.out = z; }
    char key[64];
    snprintf(key, sizeof(key), "M%d K%d N%d r%d", M, K, N, nj);
    if (defer != nullptr && job != nullptr) {    // final sums AND token reduction batched by the caller: nothing is launched here
        for (int q = 0; q < 2; ++q) { job->P[q] = a.P[q]; job->Z[q] = a.Z[q]; job->ldz[q] = a.ldz[q]; job->C[q] = a.C[q];
                                      job->cb[q] = a.cb[q]; job->partial[q] = a.partial[q]; }
        job->M = a.M; job->rpc = a.rpc; job->nchunk = a.nchunk; job->nj = nj; job->block_begin = 0;
        job->n_blocks = a.nchunk * (a.cb[0] + a.cb[1]);
        job->flops = 2.0 * M * (double)nj * ((double)K + N);
        job->bytes = 2.0 * ((double)M * ((double)K + N) + 32.0 * M);
        defer->partial[0] = a.partial[0]; defer->partial[1] = a.partial[1];
        defer->C[0] = a.C[0]; defer->C[1] = a.C[1]; defer->cb[0] = a.cb[0]; defer->cb[1] = a.cb[1];
        defer->nchunk = a.nchunk; defer->nj = nj; defer->block_begin = 0; defer->n_blocks = a.cb[0] + a.cb[1];
        defer->out = a.out;
        return MOS_OK;
    }
    MosProfScope prof(st, "lora_grad", key, 2.0 * M * (double)nj * ((double)K + N), 2.0 * ((double)M * ((double)K + N) + 32.0 * M));
    dim3 grid(a.nchunk, a.cb[0] + a.cb[1]);
    dim3 fgrid(a.cb[0] + a.cb[1]);
    if (defer != nullptr) {      // token reduction only; the caller batches the final sums (mos_lora_grad_final_all)
        switch (nj) {
            case 4: hipLaunchKernelGGL((lora_grad_kernel<T, 4>), grid, dim3(256), 0, st, a); break;
            case 8: hipLaunchKernelGGL((lora_grad_kernel<T, 8>), grid, dim3(256), 0, st, a); break;
            case 12: hipLaunchKernelGGL((lora_grad_kernel<T, 12>), grid, dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL((lora_grad_kernel<T, 16>), grid, dim3(256), 0, st, a); break;
        }
        defer->partial[0] = a.partial[0]; defer->partial[1] = a.partial[1];
        defer->C[0] = a.C[0]; defer->C[1] = a.C[1]; defer->cb[0] = a.cb[0]; defer->cb[1] = a.cb[1];
        defer->nchunk = a.nchunk; defer->nj = nj; defer->block_begin = 0; defer->n_blocks = a.cb[0] + a.cb[1];
        defer->out = a.out;
        return mos_check_launch("lora_grad");
    }
    switch (nj) {
        case 4: hipLaunchKernelGGL((lora_grad_kernel<T, 4>), grid, dim3(256), 0, st, a);
                hipLaunchKernelGGL((lora_grad_final_kernel<4>), fgrid, dim3(256), 0, st, a); break;
        case 8: hipLaunchKernelGGL((lora_grad_kernel<T, 8>), grid, dim3(256), 0, st, a);
                hipLaunchKernelGGL((lora_grad_final_kernel<8>), fgrid, dim3(256), 0, st, a); break;
        case 12: hipLaunchKernelGGL((lora_grad_kernel<T, 12>), grid, dim3(256), 0, st, a);
                 hipLaunchKernelGGL((lora_grad_final_kernel<12>), fgrid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((lora_grad_kernel<T, 16>), grid, dim3(256), 0, st, a);
                 hipLaunchKernelGGL((lora_grad_final_kernel<16>), fgrid, dim3(256), 0, st, a); break;
    }
    return mos_check_launch("lora_grad");
}

// every group of a model in ONE launch: grid (x blocks, n_groups); descriptors live in device memory
template <typename T>
__global__ void lora_pack_all_kernel(const mos_lora_group* __restrict__ groups) {
    const mos_lora_group g = groups[blockIdx.y];
    const mos_lora_sites& s = g.s;
    T* A16 = reinterpret_cast<T*>(g.A16); T* A16T = reinterpret_cast<T*>(g.A16T);
    T* Bp16 = reinterpret_cast<T*>(g.Bp16); T* BpT = reinterpret_cast<T*>(g.BpT);
    const int nA = MOS_LORA_PAD * s.K, nB = s.N * MOS_LORA_PAD;
    const int tot = nA > nB ? nA : nB;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += gridDim.x * blockDim.x) {
        if (idx < nA) {
            const int j = idx / s.K, k = idx - j * s.K;
            float v = 0.f;
            const int gi = j / s.rank;
            if (gi < s.n_sites) v = s.down[gi][(int64_t)(j - gi * s.rank) * s.K + k];
            A16[idx] = (T)v;
            A16T[(int64_t)k * MOS_LORA_PAD + j] = (T)v;
        }
        if (idx < nB) {
            const int n = idx / MOS_LORA_PAD, j = idx - n * MOS_LORA_PAD;
            float v = 0.f;
            const int gi = j / s.rank;
            if (gi < s.n_sites && n >= s.n_begin[gi] && n < s.n_begin[gi] + s.n_rows[gi])
                v = s.alpha[gi] * s.up[gi][(int64_t)(n - s.n_begin[gi]) * s.rank + (j - gi * s.rank)];
            Bp16[idx] = (T)v;
            BpT[(int64_t)j * s.N + n] = (T)v;
        }
    }
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

int mos_lora_pack_all(const mos_lora_group* groups_dev, int n_groups, int max_elems, int dtype, void* stream) {
    MOS_REQUIRE(groups_dev && n_groups >= 1 && max_elems >= 1, "mos_lora_pack_all: groups=%p n_groups=%d max_elems=%d",
                (const void*)groups_dev, n_groups, max_elems);
    hipStream_t st = (hipStream_t)stream;
    int xb = (max_elems + 255) / 256;
    if (xb > 64) xb = 64;
    MosProfScope prof(st, "lora_pack_all", "", 0.0, 0.0);
    if (dtype == MOS_F16)
        hipLaunchKernelGGL((lora_pack_all_kernel<f16_t>), dim3(xb, n_groups), dim3(256), 0, st, groups_dev);
    else if (dtype == MOS_BF16)
        hipLaunchKernelGGL((lora_pack_all_kernel<bf16_t>), dim3(xb, n_groups), dim3(256), 0, st, groups_dev);
    else
        return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_lora_pack_all: dtype %d", dtype);
    return mos_check_launch("lora_pack_all");
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

static int gemm_dims_ok(const char* who, int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldy) {
    MOS_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0 && N % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldy % 4 == 0,
                "%s: M=%d N=%d K=%d ldx=%lld ldw=%lld ldy=%lld", who, M, N, K, (long long)ldx, (long long)ldw, (long long)ldy);
    MOS_REQUIRE(((int64_t)M * ldx + K) * 2 < (1ll << 31) && ((int64_t)N * ldw + K) * 2 < (1ll << 31),
                "%s: operand exceeds the 2 GiB range of one buffer descriptor (M=%d ldx=%lld N=%d ldw=%lld)", who, M,
                (long long)ldx, N, (long long)ldw);
    return MOS_OK;
}

int mos_lora_linear_fwd(const void* x, int64_t ldx, const void* W, int64_t ldw, const void* t, const void* Bp16,
                        const float* bias, void* y, int64_t ldy, int M, int N, int K, int dtype, void* stream) {
    MOS_REQUIRE(x && W && y, "mos_lora_linear_fwd: NULL argument");
    MOS_REQUIRE((t == nullptr) == (Bp16 == nullptr), "mos_lora_linear_fwd: t and Bp16 must both be set or both NULL");
    int rc = gemm_dims_ok("mos_lora_linear_fwd", M, N, K, ldx, ldw, ldy);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MOS_F16) return launch_gemm<f16_t>(x, ldx, W, ldw, t, nullptr, Bp16, bias, y, ldy, nullptr, M, N, K, st);
    if (dtype == MOS_BF16) return launch_gemm<bf16_t>(x, ldx, W, ldw, t, nullptr, Bp16, bias, y, ldy, nullptr, M, N, K, st);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_lora_linear_fwd: dtype %d", dtype);
}

int mos_lora_linear_fused_fwd(const void* x, int64_t ldx, const void* W, int64_t ldw, const void* A16, const void* Bp16,
                              const float* bias, void* y, int64_t ldy, void* t_out, int M, int N, int K, int dtype,
                              void* stream) {
    MOS_REQUIRE(x && W && y && A16 && Bp16, "mos_lora_linear_fused_fwd: NULL argument");
    int rc = gemm_dims_ok("mos_lora_linear_fused_fwd", M, N, K, ldx, ldw, ldy);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MOS_F16) return launch_gemm<f16_t>(x, ldx, W, ldw, nullptr, A16, Bp16, bias, y, ldy, t_out, M, N, K, st);
    if (dtype == MOS_BF16) return launch_gemm<bf16_t>(x, ldx, W, ldw, nullptr, A16, Bp16, bias, y, ldy, t_out, M, N, K, st);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_lora_linear_fused_fwd: dtype %d", dtype);
}

/* Forward GEMM (plain: A16 = Bp16 = NULL, or LoRA-fused) with an epilogue variant, see mos_gemm_epilogue. */
int mos_lora_linear_fwd_ex(const void* x, int64_t ldx, const void* W, int64_t ldw, const void* A16, const void* Bp16,
                           const float* bias, void* y, int64_t ldy, void* t_out, int M, int N, int K, int dtype,
                           const mos_gemm_epilogue* epi, void* stream) {
    MOS_REQUIRE(x && W && y && epi, "mos_lora_linear_fwd_ex: NULL argument");
    MOS_REQUIRE((A16 == nullptr) == (Bp16 == nullptr), "mos_lora_linear_fwd_ex: A16 and Bp16 must both be set or both NULL");
    const int nout = N;
    int rc = gemm_dims_ok("mos_lora_linear_fwd_ex", M, N, K, ldx, ldw, ldy);
    if (rc) return rc;
    MOS_REQUIRE(epi->residual == nullptr || (epi->ldr % 8 == 0 && epi->ldr >= nout),
                "mos_lora_linear_fwd_ex: residual row stride %lld (need %% 8 == 0, >= %d)", (long long)epi->ldr, nout);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MOS_F16)
        return launch_gemm<f16_t>(x, ldx, W, ldw, nullptr, A16, Bp16, bias, y, ldy, t_out, M, N, K, st, epi->residual, epi->ldr);
    if (dtype == MOS_BF16)
        return launch_gemm<bf16_t>(x, ldx, W, ldw, nullptr, A16, Bp16, bias, y, ldy, t_out, M, N, K, st, epi->residual, epi->ldr);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_lora_linear_fwd_ex: dtype %d", dtype);
}

int64_t mos_lora_bwd_workspace_bytes(int M, int N, int K) {
    // legacy two-pass plan and the fused single-launch plan share one size function: the larger of the two
    int64_t best = 0;
    for (int C : {N, K}) {
        const int rpc = tn_rows_per_chunk(M, C);
        const int64_t nchunk = (M + rpc - 1) / rpc;
        const int64_t b = nchunk * MOS_LORA_PAD * C * (int64_t)sizeof(float);
        if (b > best) best = b;
    }
    if (M > 0 && N > 0 && K > 0) {
        int rpc, nchunk;
        lora_grad_plan(M, N, K, &rpc, &nchunk);
        const int64_t b = (int64_t)nchunk * MOS_LORA_PAD * ((int64_t)N + K) * (int64_t)sizeof(float);
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
    if (dx) {  // dx = dy . Wt^T + (dy . BpT^T) . A16T^T, dt written on the way
        rc = gemm_dims_ok("mos_lora_linear_bwd", M, K, N, lddy, ldwt, lddx);
        if (rc) return rc;
        rc = h ? launch_gemm<f16_t>(dy, lddy, Wt, ldwt, nullptr, lora ? BpT : nullptr, lora ? A16T : nullptr, nullptr, dx, lddx, lora ? dt : nullptr, M, K, N, st)
               : launch_gemm<bf16_t>(dy, lddy, Wt, ldwt, nullptr, lora ? BpT : nullptr, lora ? A16T : nullptr, nullptr, dx, lddx, lora ? dt : nullptr, M, K, N, st);
        if (rc) return rc;
    } else if (lora) {  // no input gradient wanted: dt = dy . BpT^T on its own
        rc = h ? launch_skinny_nt<f16_t>(dy, lddy, BpT, dt, M, N, st) : launch_skinny_nt<bf16_t>(dy, lddy, BpT, dt, M, N, st);
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

static int fused_bwd_impl(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* Wt, int64_t ldwt,
                          const void* t, const void* A16T, const void* BpT, void* dt, void* dx, int64_t lddx,
                          const mos_lora_grad_out* grads_host, void* ws, int M, int N, int K,
                          int lora_cols, int dtype, void* stream, mos_lora_final_rec* defer, mos_lora_grad_job* job = nullptr) {
    MOS_REQUIRE(dy && x && t && A16T && BpT && dt, "mos_lora_linear_fused_bwd: NULL argument");
    MOS_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0 && N % 8 == 0 && lddy % 8 == 0 && ldx % 8 == 0,
                "mos_lora_linear_fused_bwd: M=%d N=%d K=%d lddy=%lld ldx=%lld", M, N, K, (long long)lddy, (long long)ldx);
    MOS_REQUIRE(dx == nullptr || (Wt && lddx % 4 == 0 && ldwt % 8 == 0), "mos_lora_linear_fused_bwd: dx needs Wt");
    MOS_REQUIRE(grads_host == nullptr || ws, "mos_lora_linear_fused_bwd: gradients need ws");
    MOS_REQUIRE(grads_host == nullptr || (grads_host->n_sites >= 1 && grads_host->n_sites <= 4 && grads_host->rank >= 1 &&
                                          grads_host->n_sites * grads_host->rank <= MOS_LORA_PAD &&
                                          lora_cols == grads_host->n_sites * grads_host->rank),
                "mos_lora_linear_fused_bwd: bad gradient descriptor (sites x rank must equal lora_cols = %d)", lora_cols);
    if (dtype != MOS_F16 && dtype != MOS_BF16) return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_lora_linear_fused_bwd: dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const bool h = (dtype == MOS_F16);
    int rc;
    if (dx) {
        rc = gemm_dims_ok("mos_lora_linear_fused_bwd", M, K, N, lddy, ldwt, lddx);
        if (rc) return rc;
        rc = h ? launch_gemm<f16_t>(dy, lddy, Wt, ldwt, nullptr, BpT, A16T, nullptr, dx, lddx, dt, M, K, N, st)
               : launch_gemm<bf16_t>(dy, lddy, Wt, ldwt, nullptr, BpT, A16T, nullptr, dx, lddx, dt, M, K, N, st);
    } else {
        rc = h ? launch_skinny_nt<f16_t>(dy, lddy, BpT, dt, M, N, st) : launch_skinny_nt<bf16_t>(dy, lddy, BpT, dt, M, N, st);
    }
    if (rc) return rc;
    if (grads_host == nullptr) return MOS_OK;
    return h ? launch_lora_grad<f16_t>(dt, x, ldx, t, dy, lddy, nullptr, nullptr, grads_host, (float*)ws, M, N, K, lora_cols, st, defer, job)
             : launch_lora_grad<bf16_t>(dt, x, ldx, t, dy, lddy, nullptr, nullptr, grads_host, (float*)ws, M, N, K, lora_cols, st, defer, job);
}

int mos_lora_linear_fused_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* Wt, int64_t ldwt,
                              const void* t, const void* A16T, const void* BpT, void* dt, void* dx, int64_t lddx,
                              const mos_lora_grad_out* grads_host, void* ws, int M, int N, int K,
                              int lora_cols, int dtype, void* stream) {
    return fused_bwd_impl(dy, lddy, x, ldx, Wt, ldwt, t, A16T, BpT, dt, dx, lddx, grads_host, ws, M, N, K, lora_cols, dtype,
                          stream, nullptr);
}

int mos_lora_linear_fused_bwd_deferred(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* Wt, int64_t ldwt,
                                       const void* t, const void* A16T, const void* BpT, void* dt, void* dx, int64_t lddx,
                                       const mos_lora_grad_out* grads_host, void* ws, int M, int N, int K,
                                       int lora_cols, int dtype, void* stream, mos_lora_final_rec* rec_host) {
    MOS_REQUIRE(rec_host && grads_host, "mos_lora_linear_fused_bwd_deferred: needs grads_host and rec_host");
    return fused_bwd_impl(dy, lddy, x, ldx, Wt, ldwt, t, A16T, BpT, dt, dx, lddx, grads_host, ws, M, N, K, lora_cols, dtype,
                          stream, rec_host);
}

int mos_lora_linear_fused_bwd_deferred_all(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* Wt, int64_t ldwt,
                                           const void* t, const void* A16T, const void* BpT, void* dt, void* dx, int64_t lddx,
                                           const mos_lora_grad_out* grads_host, void* ws, int M, int N, int K,
                                           int lora_cols, int dtype, void* stream, mos_lora_final_rec* rec_host,
                                           mos_lora_grad_job* job_host) {
    MOS_REQUIRE(rec_host && grads_host && job_host, "mos_lora_linear_fused_bwd_deferred_all: needs grads_host, rec_host and job_host");
    return fused_bwd_impl(dy, lddy, x, ldx, Wt, ldwt, t, A16T, BpT, dt, dx, lddx, grads_host, ws, M, N, K, lora_cols, dtype,
                          stream, rec_host, job_host);
}

int mos_lora_grad_all(const mos_lora_grad_job* jobs_dev, int n_jobs, int total_blocks, int nj, int dtype, double flops, double bytes,
                      void* stream) {
    MOS_REQUIRE(jobs_dev && n_jobs > 0 && total_blocks > 0, "mos_lora_grad_all: n_jobs=%d total_blocks=%d", n_jobs, total_blocks);
    MOS_REQUIRE(nj == 4 || nj == 8 || nj == 12 || nj == 16, "mos_lora_grad_all: nj=%d (4, 8, 12 or 16: every job of the table has it)", nj);
    MOS_REQUIRE(dtype == MOS_F16 || dtype == MOS_BF16, "mos_lora_grad_all: dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    char key[64];
    snprintf(key, sizeof(key), "r%d groups%d blocks%d", nj, n_jobs, total_blocks);
    MosProfScope prof(st, "lora_grad", key, flops, bytes);
    const dim3 grid(total_blocks), block(256);
#define LGA(T_, NJ_) hipLaunchKernelGGL((lora_grad_all_kernel<T_, NJ_>), grid, block, 0, st, jobs_dev, n_jobs)
    if (dtype == MOS_F16) { switch (nj) { case 4: LGA(f16_t, 4); break; case 8: LGA(f16_t, 8); break; case 12: LGA(f16_t, 12); break; default: LGA(f16_t, 16); break; } }
    else { switch (nj) { case 4: LGA(bf16_t, 4); break; case 8: LGA(bf16_t, 8); break; case 12: LGA(bf16_t, 12); break; default: LGA(bf16_t, 16); break; } }
#undef LGA
    return mos_check_launch("lora_grad_all");
}

int mos_lora_grad_final_all(const mos_lora_final_rec* recs_dev, int n_recs, int total_blocks, void* stream) {
    MOS_REQUIRE(recs_dev && n_recs > 0 && total_blocks > 0, "mos_lora_grad_final_all: n_recs=%d total_blocks=%d", n_recs, total_blocks);
    hipStream_t st = (hipStream_t)stream;
    char key[48];
    snprintf(key, sizeof(key), "groups%d blocks%d", n_recs, total_blocks);
    MosProfScope prof(st, "lora_grad_final_all", key, 0.0, 0.0);
    hipLaunchKernelGGL(lora_grad_final_all_kernel, dim3(total_blocks), dim3(256), 0, st, recs_dev, n_recs);
    return mos_check_launch("lora_grad_final_all");
}


}  // extern "C"

// mos_attn.hip — fused attention for the SD-1.5 UNet (head dims 40 / 80 / 160, 8 heads) and its CLIP text tower
// (head dim 64, 12 heads, causal) on gfx950.
//
// Replaces, in one kernel per direction, what the reference does with baddbmm + softmax + bmm on a
// materialised (B*H, N, Nkv) probability tensor, or with xformers (reference
// mixofshow/models/edlora.py:77-83,151-156; pipeline_regionally_t2iadapter.py:111-116), and the
// per-region einsum/softmax/einsum + boolean-mask scatter of region_rewrite (:32-86).
//
// Design (CDNA4, wave64, v_mfma_f32_32x32x16_{f16,bf16}):
//   * "swapped" products so that every softmax statistic is lane-local:
//       S^T = K . Q^T      (A <- K rows from LDS, B <- Q rows held in registers)
//       O^T = V^T . P^T    (A <- V^T from LDS,    B <- P^T = the S^T accumulator itself)
//     MFMA 32x32x16: lane l supplies A[i=l&31][k=8*(l>>5)..+8], B[k=8*(l>>5)..+8][j=l&31];
//     D: lane l holds D[i=(r&3)+8*(r>>2)+4*(l>>5)][j=l&31], r=0..15.
//     With S^T in D-layout, lane l owns query column q=l&31 and 16 kv rows; registers 8s..8s+7 of
//     the accumulator are EXACTLY a B operand for contraction step s if the A operand enumerates
//     kv in the same order: k=8h+j  <->  kv = 16s + 4h + (j<4 ? j : j+4)   (h = l>>5).
//     So P never leaves registers: no LDS round trip, no permlane.  V^T (and K^T, Q^T, dO^T in the
//     backward) is produced once per tile by a transposing LDS store shared by the 4 waves.
//   * the head_to_batch_dim permutes of the reference are folded into addressing: q/k/v/o are read
//     and written in token-major (B, N, H*d) layout with explicit strides.
//   * grid: blockIdx.x = head + H*(q_block + n_q_blocks*batch): blocks land on XCD (id % 8) = head,
//     so all query blocks that re-read one head's K/V share one XCD's L2.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "mos_common.h"

// Tuning history (measured on MI355X, profiles/r02_attention_variants.txt): the forward takes its softmax row sums from
// the P.V MFMA (ones row in the V^T padding, -8 % at d = 40); software-pipelined dK/dV drafts (+37..+44 %), statistics folded
// into the pad columns (+-0), 8-wave dQ blocks (+47 %), wave-slot staggering and -fno-slp-vectorize (+-0) were measured and
// removed. MOS_DQ_OCC stays a knob (tools/build_variant.sh).
#ifndef MOS_DQ_OCC
#define MOS_DQ_OCC 2
#endif
#ifndef MOS_ATTN_PV16          // 0: variant build with the 32x32x16 P.V form at d = 40 (same-box A/B)
#define MOS_ATTN_PV16 1
#endif

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;
constexpr int KV_TILE = 64;
// transposed-tile row stride: 72 el = 144 B = 16*9 -> every row is 16-byte aligned and the 16 lanes of a
// ds_read_b128 lane group (consecutive rows) land on 16 distinct 16-byte bank slots (9 is odd): conflict-free
constexpr int TS = KV_TILE + 8;
// Column order inside a transposed tile: within each group of 16 tokens, bits 2 and 3 of the token index are
// swapped, i.e. tokens are stored [0-3, 8-11, 4-7, 12-15]. A lane half h needs tokens {4h..4h+3, 8+4h..8+4h+3} of
// the group (the rows its S^T accumulator registers hold, see the header comment): with this order they are 8
// CONTIGUOUS elements, one ds_read_b128 (256 B/clk) instead of a ds_read2_b64 (128 B/clk).
__device__ __forceinline__ constexpr int tr_col(int c) { return (c & ~12) | ((c & 4) << 1) | ((c & 8) >> 1); }

// Column order of a transposed V tile for the 16x16x32 form of O^T = V^T . P^T (round 6, d = 40 forward): within each 32-key
// sub-tile the four k-groups of the MFMA (lane >> 4) hold keys {0-3, 8-11}, {16-19, 24-27}, {4-7, 12-15}, {20-23, 28-31} -- what four
// v_permlane16_swap of the packed S^T accumulator leave in them (see attend) -- each as 8 CONTIGUOUS elements (one ds_read_b128).
__device__ __forceinline__ constexpr int tr_col16(int c) {
    return (c & 32) | ((((c >> 2) & 1) * 2 + ((c >> 4) & 1)) << 3) | (((c >> 3) & 1) << 2) | (c & 3);
}

template <int D>
struct HD {
    static constexpr int DK = (D + 15) / 16 * 16;  // contraction length of Q.K^T (zero padded)
    static constexpr int DV = (D + 31) / 32 * 32;  // rows of the transposed outputs (zero padded)
    static constexpr int KS = DK / 16;
    static constexpr int DT = DV / 32;
    static constexpr int RS = DK + 8;  // row-major tile stride: (DK/8+1) odd -> b128 reads conflict-free
    static constexpr int DCH = D / 8;  // 16-byte chunks per row
    static constexpr int ROW_TILE_ELEMS = KV_TILE * RS;
    static constexpr int TR_TILE_ELEMS = DV * TS;
};

// dK/dV kernel: 4 staged tiles (Q, dO, Q^T, dO^T); double buffering fits the 160 KB LDS up to d = 80
template <int D> struct DKDV_NBUF { static constexpr int value = D <= 80 ? 2 : 1; };

struct AttnArgs {
    const void* q; const void* k; const void* v; void* o;
    float* lse; const int32_t* tok_idx; float* pcols; int n_pcols;
    int B, H, Nq, Nkv, nqb;
    int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
    float scale;
    int causal;      // 1: key index > query index is masked (CLIP text tower)
};

struct AttnBwdArgs {
    const void* q; const void* k; const void* v; const void* dO;
    const void* o; int64_t o_bs, o_rs; const float* pcols;     // dQ kernel only: it forms D = rowsum(dO o O) (+ sum_t dpcols pcols)
    const float* lse; float* Dvec; const int32_t* tok_idx; const float* dpcols; int n_pcols;   // in its prologue and leaves it in Dvec
    void* dq; void* dk; void* dv; float* part;  // part: fp32 split partials [2][nsplit][B*H][Nkv][D]
    int B, H, Nq, Nkv, nqb, nkb, nsplit, q_per_split;
    int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
    float scale;
    int causal;
};

// ---- LDS tiles ----------------------------------------------------------------------------------
// Row-major tile [64][RS]: rows = tokens. Transposed tile [DV][TS]: element (d, token), columns in tr_col order.
template <typename T, int D, int NT = 256>
__device__ __forceinline__ void zero_row_pads(T* lds, int tid) {  // columns [D, DK) of a row-major tile
    constexpr int DK = HD<D>::DK, RS = HD<D>::RS;
    if constexpr (DK > D) {
        for (int c = tid; c < KV_TILE * ((DK - D) / 8); c += NT) {
            const int row = c / ((DK - D) / 8), cc = c % ((DK - D) / 8);
            st16(lds + row * RS + D + cc * 8, u32x4{0, 0, 0, 0});
        }
    }
}
template <typename T, int D, int NT = 256>
__device__ __forceinline__ void zero_tr_pads(T* ldsT, int tid) {  // rows [D, DV) of a transposed tile
    constexpr int DV = HD<D>::DV;
    if constexpr (DV > D) {
        uint32_t* p = reinterpret_cast<uint32_t*>(ldsT + D * TS);
        for (int c = tid; c < (DV - D) * TS / 2; c += NT) p[c] = 0u;
    }
}
// B-operand fragments of a register-resident row (query / dO / key / value row of this lane).
template <typename T, int D>
__device__ __forceinline__ void load_row_frags(typename MT<T>::v8 (&f)[HD<D>::KS], const T* row, bool valid, int hh) {
#pragma unroll
    for (int ks = 0; ks < HD<D>::KS; ++ks) {
        const int col = ks * 16 + hh * 8;
        f[ks] = as_v8<T>((valid && col < D) ? ld16(row + col) : u32x4{0, 0, 0, 0});
    }
}
template <typename T>
__device__ __forceinline__ typename MT<T>::v8 acc_to_bfrag(const f32x16& x, int s2) {
    typename MT<T>::v8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (T)x[s2 * 8 + j];
    return v;
}
// A operand of a transposed tile for contraction step (t, s2): rows = 32*dt + (l&31).
template <typename T>
__device__ __forceinline__ typename MT<T>::v8 tr_afrag(const T* ldsT_row, int t, int s2, int hh) {
    return as_v8<T>(ld16(ldsT_row + 32 * t + 16 * s2 + 8 * hh));
}
__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// ---- split staging (issue global loads early, write LDS late: the loads fly under the MFMAs) --------
// Tiles are fetched with BUFFER loads through a descriptor that covers exactly the valid rows of one (batch, head)
// slice: rows past the end of the sequence read as zeros in hardware, so the ragged last tile needs no clamped
// addresses, no selects and no masks, and the per-tile address arithmetic is one 32-bit add per load (a wave
// issues about one instruction per 4 cycles, so this bookkeeping is paid in MFMA issue slots). A plain
// `cond ? load : 0` would also make hipcc branch around the load and wait vmcnt(0) right after it, serialising
// the prefetch behind the MFMAs it is supposed to overlap (cdna guide, ".s-level traps" (c)).
// bytes of a token-major slice: n rows of D contiguous elements, row stride rs elements (other heads in between)
template <typename T, int D>
__device__ __forceinline__ uint32_t slice_bytes(int n, int64_t rs) {
    return (uint32_t)(((int64_t)(n - 1) * rs + D) * (int64_t)sizeof(T));
}

template <typename T, int D, int NT = 256>
struct RowStage {
    static constexpr int DCH = HD<D>::DCH, RS = HD<D>::RS;
    static constexpr int N = (KV_TILE * DCH + NT - 1) / NT;
    u32x4 r[N];
    int voff[N];   // byte offset of chunk i inside a tile: (row * rs + 8 * cc) * sizeof(T)
    __device__ __forceinline__ void init(int64_t rs, int tid) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = min(tid + NT * i, KV_TILE * DCH - 1);
            const int row = c / DCH, cc = c - row * DCH;
            voff[i] = (row * (int)rs + cc * 8) * (int)sizeof(T);
        }
    }
    __device__ __forceinline__ void load(rsrc_t src, int tile_byte_off) {
#pragma unroll
        for (int i = 0; i < N; ++i) r[i] = ldbuf16(src, voff[i] + tile_byte_off);
    }
    __device__ __forceinline__ void store(T* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = tid + NT * i;
            const int row = c / DCH, cc = c - row * DCH;
            if (N * NT <= KV_TILE * DCH || c < KV_TILE * DCH) st16(lds + row * RS + cc * 8, r[i]);
        }
    }
};
template <typename T, int D, int NT = 256>
struct TrStage {
    static constexpr int DCH = HD<D>::DCH;
    static constexpr int N = (32 * DCH + NT - 1) / NT;
    u32x4 r0[N], r1[N];
    int voff[N];   // byte offset of (row 2p, chunk cc); row 2p+1 is + row_bytes
    int row_bytes;
    __device__ __forceinline__ void init(int64_t rs, int tid) {
        row_bytes = (int)rs * (int)sizeof(T);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int it = min(tid + NT * i, 32 * DCH - 1);
            const int p = it & 31, cc = it >> 5;
            voff[i] = (2 * p * (int)rs + cc * 8) * (int)sizeof(T);
        }
    }
    __device__ __forceinline__ void load(rsrc_t src, int tile_byte_off) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            r0[i] = ldbuf16(src, voff[i] + tile_byte_off);
            r1[i] = ldbuf16(src, voff[i] + tile_byte_off + row_bytes);
        }
    }
    // the row-major image [64][RS] of the same tile from the same registers (kernels that need both images of a
    // tensor fetch it once)
    __device__ __forceinline__ void store_rows(T* lds, int tid) const {
        constexpr int RS = HD<D>::RS;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int it = tid + NT * i;
            const int p = it & 31, cc = it >> 5;
            if (N * NT <= 32 * DCH || it < 32 * DCH) {
                st16(lds + (2 * p) * RS + cc * 8, r0[i]);
                st16(lds + (2 * p + 1) * RS + cc * 8, r1[i]);
            }
        }
    }
    template <bool MAP16 = false>
    __device__ __forceinline__ void store(T* ldsT, int tid) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int it = tid + NT * i;
            const int p = it & 31, cc = it >> 5;
            if (N * NT <= 32 * DCH || it < 32 * DCH) {
                uint32_t* dst = reinterpret_cast<uint32_t*>(ldsT + (cc * 8) * TS + (MAP16 ? tr_col16(2 * p) : tr_col(2 * p)));
                // word e of the packed pair = {row 2p+1 elem e (high half), row 2p elem e (low half)}:
                // v_perm_b32 byte selectors over {src0 = r1 word (bytes 4..7), src1 = r0 word (bytes 0..3)}
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    dst[(2 * w) * (TS / 2)] = __builtin_amdgcn_perm(r1[i][w], r0[i][w], 0x05040100u);
                    dst[(2 * w + 1) * (TS / 2)] = __builtin_amdgcn_perm(r1[i][w], r0[i][w], 0x07060302u);
                }
            }
        }
    }
};

// ---- one attention pass of a wave's NQ x 32 queries over all keys of one source ----------------
// Returns unnormalised O^T accumulators, running max m (raw score units) and PER-LANE partial sums l.
// LSUM: the caller put a row of ones into row D of the transposed V tiles (a padding row), so accumulator row D of O^T
// IS the running row sum (rescaled with O^T for free): no per-element adds here, `l` is left untouched.
// CAUSAL is a template parameter and the ragged last tile is a peeled copy of the loop body (TAIL): as run-time flags
// both masks were if-converted by hipcc into a compare + select per score element in EVERY tile -- ~350 of the 760
// instructions of the d = 40 forward loop, for masks the UNet never uses (no causal attention; 4096/1024/256/64 keys).
// PV16 (round 6, d = 40 self-attention forward; VERDICT r05 next #3, measured): O^T = V^T . P^T on v_mfma_f32_16x16x32 with d padded
// 40 -> 48 (three 16-row tiles) instead of 40 -> 64 (two 32-row tiles of the 32x32x16 form): 24 x 16 instead of 16 x 32 MFMA cycles per
// 64-key tile and 64 queries, 6 instead of 8 A-fragment reads. The 32x32 accumulator layout of S^T is NOT a 16x16x32 B operand; on
// gfx950 it becomes two of them (queries 0-15 / 16-31 of the block) with FOUR v_permlane16_swap_b32 per 32x32 tile: the packed
// accumulator words of key blocks {0, 1} and {2, 3} trade their odd / even 16-lane rows, after which lane l holds query l & 15 and
// the eight keys of k-group l >> 4 in the order tr_col16 gives the V^T tile. Accumulators o16[iq][u][dm]: query 16 u + (l & 15),
// rows d = 16 dm + 4 (l >> 4) + r.
template <typename T, int D, int NQ, bool PCOLS, bool LSUM = false, bool CAUSAL = false, bool PV16 = false>
__device__ __forceinline__ void attend(const T* kbase, int64_t k_rs, const T* vbase, int64_t v_rs, int Nkv,
                                       float c /* scale*log2e */, T* Ks_, T* Vt_,
                                       const typename MT<T>::v8 (&qf)[NQ][HD<D>::KS],
                                       f32x16 (&o)[PV16 ? 1 : NQ][PV16 ? 1 : HD<D>::DT], f32x4 (&o16)[PV16 ? NQ : 1][2][3],
                                       float (&m)[NQ], float (&l)[NQ],
                                       const int (&tok)[MOS_MAX_PCOLS], int n_pcols,
                                       float (&cap)[NQ][MOS_MAX_PCOLS], int tid, int l31, int hh, int qfirst = 0) {
    typedef typename MT<T>::v8 v8;
    constexpr int KS = HD<D>::KS, DT = HD<D>::DT, RS = HD<D>::RS;
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) {
        m[iq] = NEG_BIG; l[iq] = 0.f;
        if constexpr (PV16) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int dm = 0; dm < 3; ++dm) o16[iq][u][dm] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[iq][dt][r] = 0.f;
        }
    }
    // Double-buffered LDS tiles (Ks/Vt hold 2 tiles each) + register prefetch: the global loads of tile i+1 are
    // issued before the MFMAs of tile i and written to the other buffer after them; ONE barrier per tile.
    RowStage<T, D> kst;
    TrStage<T, D> vst;
    kst.init(k_rs, tid);
    vst.init(v_rs, tid);
    const rsrc_t ksrc = make_rsrc(kbase, slice_bytes<T, D>(Nkv, k_rs));
    const rsrc_t vsrc = make_rsrc(vbase, slice_bytes<T, D>(Nkv, v_rs));
    const int k_tile_bytes = KV_TILE * (int)k_rs * (int)sizeof(T), v_tile_bytes = KV_TILE * (int)v_rs * (int)sizeof(T);
    __syncthreads();  // earlier users of the LDS buffers (previous source / prologue zeroing) are done
    kst.load(ksrc, 0);
    vst.load(vsrc, 0);
    kst.store(Ks_, tid);
    vst.template store<PV16>(Vt_, tid);
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see the dK/dV kernel
    int cur = 0;
    // tail_c: 0 = full tile, 1 = ragged tile (peeled copy), 2 = decided at run time (one body: the PCOLS instantiations --
    // 77-key cross attention, two tiles, latency-bound -- spill registers with two copies of the body)
    auto tile = [&](const int kv0, auto tail_c) __attribute__((always_inline)) {
        constexpr int TM = decltype(tail_c)::value;
        const bool tail = TM == 2 ? (kv0 + KV_TILE > Nkv) : (TM == 1);
        const T* Ks = Ks_ + cur * HD<D>::ROW_TILE_ELEMS;
        const T* Vt = Vt_ + cur * HD<D>::TR_TILE_ELEMS;
        const bool more = kv0 + KV_TILE < Nkv;
        if (more) {
            kst.load(ksrc, (kv0 / KV_TILE + 1) * k_tile_bytes);
            vst.load(vsrc, (kv0 / KV_TILE + 1) * v_tile_bytes);
        }

        f32x16 s[NQ][2];
#pragma unroll
        for (int iq = 0; iq < NQ; ++iq)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[iq][t][r] = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v8 a = as_v8<T>(ld16(Ks + (32 * t + l31) * RS + ks * 16 + hh * 8));
#pragma unroll
                for (int iq = 0; iq < NQ; ++iq) s[iq][t] = MT<T>::mfma32(a, qf[iq][ks], s[iq][t]);
            }
        v8 pf[NQ][2][2];
#pragma unroll
        for (int iq = 0; iq < NQ; ++iq) {
            if (tail) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kv0 + 32 * t + acc_row(r, hh) >= Nkv) s[iq][t][r] = NEG_BIG;
            }
            if constexpr (CAUSAL) {       // this lane's query of sub-tile iq is qfirst + 32*iq + l31
                const int qidx = qfirst + 32 * iq + l31;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kv0 + 32 * t + acc_row(r, hh) > qidx) s[iq][t][r] = NEG_BIG;
            }
            if constexpr (PCOLS) {
#pragma unroll
                for (int tt = 0; tt < MOS_MAX_PCOLS; ++tt)
                    if (tt < n_pcols) {
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                if (kv0 + 32 * t + acc_row(r, hh) == tok[tt]) cap[iq][tt] = s[iq][t][r];
                    }
            }
            float mx = m[iq];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[iq][t][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float alpha = __builtin_amdgcn_exp2f((m[iq] - mx) * c);
            m[iq] = mx;
            const float mc = mx * c;
            float ls = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[iq][t][r] * c - mc);
                    s[iq][t][r] = p;
                    if constexpr (!LSUM) ls += p;
                }
            if constexpr (!LSUM) l[iq] = l[iq] * alpha + ls;
            if (__any(alpha != 1.0f)) {  // wave-uniform: once the running max has settled no lane rescales
                if constexpr (PV16) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {       // this lane's accumulators belong to query 16 u + (l & 15)
                        const float au = __shfl(alpha, 16 * u + (l31 & 15));
#pragma unroll
                        for (int dm = 0; dm < 3; ++dm)
#pragma unroll
                            for (int r = 0; r < 4; ++r) o16[iq][u][dm][r] *= au;
                    }
                } else {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[iq][dt][r] *= alpha;
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) pf[iq][t][s2] = acc_to_bfrag<T>(s[iq][t], s2);
        }
        if constexpr (PV16) {
            // pf[iq][t][0] = packed words of key blocks 0, 1 (accumulator registers 0..7), pf[iq][t][1] = of key blocks 2, 3: four
            // row swaps turn the pair into the B operands of query groups u = 0 (lanes' queries 0-15) and u = 1 (16-31)
            v8 pb[NQ][2][2];
#pragma unroll
            for (int iq = 0; iq < NQ; ++iq)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const u32x4 lo = from_v8<T>(pf[iq][t][0]), hi = from_v8<T>(pf[iq][t][1]);
                    u32x4 b0, b1;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const auto sw = __builtin_amdgcn_permlane16_swap(lo[w], hi[w], false, false);
                        b0[w] = sw[0]; b1[w] = sw[1];
                    }
                    pb[iq][t][0] = as_v8<T>(b0); pb[iq][t][1] = as_v8<T>(b1);
                }
            const int kg = (l31 >> 4) + 2 * hh;              // lane >> 4
#pragma unroll
            for (int dm = 0; dm < 3; ++dm) {
                const T* vrow = Vt + (16 * dm + (l31 & 15)) * TS + 8 * kg;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const v8 a = as_v8<T>(ld16(vrow + 32 * t));
#pragma unroll
                    for (int iq = 0; iq < NQ; ++iq)
#pragma unroll
                        for (int u = 0; u < 2; ++u) o16[iq][u][dm] = MT<T>::mfma16(a, pb[iq][t][u], o16[iq][u][dm]);
                }
            }
        } else {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const T* vrow = Vt + (32 * dt + l31) * TS;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const v8 a = tr_afrag<T>(vrow, t, s2, hh);
#pragma unroll
                        for (int iq = 0; iq < NQ; ++iq) o[iq][dt] = MT<T>::mfma32(a, pf[iq][t][s2], o[iq][dt]);
                    }
            }
        }
        if (more) {
            kst.store(Ks_ + (cur ^ 1) * HD<D>::ROW_TILE_ELEMS, tid);
            vst.template store<PV16>(Vt_ + (cur ^ 1) * HD<D>::TR_TILE_ELEMS, tid);
        }
        __syncthreads();  // tile i+1 visible; every wave is done reading tile i before it is overwritten next round
        cur ^= 1;
    };
    int kv0 = 0;
    if constexpr (PCOLS) {
        for (; kv0 < Nkv; kv0 += KV_TILE) tile(kv0, std::integral_constant<int, 2>{});
    } else {
        for (; kv0 + KV_TILE <= Nkv; kv0 += KV_TILE) tile(kv0, std::integral_constant<int, 0>{});
        if (kv0 < Nkv) tile(kv0, std::integral_constant<int, 1>{});
    }
}

template <typename T, int D>
__device__ __forceinline__ void store_out_rows(T* orow, bool valid, const f32x16 (&o)[HD<D>::DT], float mul, int hh) {
    if (!valid) return;
#pragma unroll
    for (int dt = 0; dt < HD<D>::DT; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int db = 32 * dt + 8 * r4 + 4 * hh;
            if (db < D)
                st8(orow + db, pack4<T>(o[dt][4 * r4] * mul, o[dt][4 * r4 + 1] * mul, o[dt][4 * r4 + 2] * mul,
                                         o[dt][4 * r4 + 3] * mul));
        }
}

// ---- forward -----------------------------------------------------------------------------------
template <typename T, int D, int QW, bool PCOLS, bool CAUSAL = false>
__global__ __launch_bounds__(256, ((D <= 40 || (D <= 80 && QW == 32)) ? 2 : 1)) void attn_fwd_kernel(AttnArgs a) {
    typedef typename MT<T>::v8 v8;
    constexpr int NQ = QW / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Ks = reinterpret_cast<T*>(smem_raw);
    T* Vt = Ks + 2 * HD<D>::ROW_TILE_ELEMS;  // [2] K tiles, then [2] V^T tiles (double buffered)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int h = blockIdx.x % a.H;
    const int rest = blockIdx.x / a.H;
    const int qb = rest % a.nqb, b = rest / a.nqb;
    const int q0 = qb * (4 * QW) + wave * QW;

    zero_row_pads<T, D>(Ks, tid); zero_row_pads<T, D>(Ks + HD<D>::ROW_TILE_ELEMS, tid);
    zero_tr_pads<T, D>(Vt, tid); zero_tr_pads<T, D>(Vt + HD<D>::TR_TILE_ELEMS, tid);
    constexpr bool LSUM = HD<D>::DV > D;    // d = 40, 80: a free padding row exists
    if constexpr (LSUM) {      // row D of both V^T buffers = 1 for every key (keys past Nkv have p = 0 anyway)
        __syncthreads();       // after the zero fill of the padding rows (other threads' words)
        if (tid < KV_TILE) {
            Vt[D * TS + tid] = (T)1.0f;
            Vt[HD<D>::TR_TILE_ELEMS + D * TS + tid] = (T)1.0f;
        }
    }

    const T* qp = (const T*)a.q + (int64_t)b * a.q_bs + h * D;
    const T* kp = (const T*)a.k + (int64_t)b * a.k_bs + h * D;
    const T* vp = (const T*)a.v + (int64_t)b * a.v_bs + h * D;
    T* op = (T*)a.o + (int64_t)b * a.o_bs + h * D;

    v8 qf[NQ][HD<D>::KS];
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) {
        const int qi = q0 + 32 * iq + l31;
        load_row_frags<T, D>(qf[iq], qp + (int64_t)min(qi, a.Nq - 1) * a.q_rs, qi < a.Nq, hh);
    }
    int tok[MOS_MAX_PCOLS] = {-1, -1, -1, -1};
    float cap[NQ][MOS_MAX_PCOLS];
    if constexpr (PCOLS) {
#pragma unroll
        for (int tt = 0; tt < MOS_MAX_PCOLS; ++tt) {
            if (tt < a.n_pcols) tok[tt] = a.tok_idx[b * a.n_pcols + tt];
#pragma unroll
            for (int iq = 0; iq < NQ; ++iq) cap[iq][tt] = NEG_BIG;
        }
    }
    // level-0 self attention with 32 queries per wave (small grids: the CFG pair of a sample): the 16x16x32 P.V form. Measured,
    // same box (profiles/r06c14_*): B2 N6144 (32 queries per wave) 172.9 -> 169.0 us in the sample, 191.0 / 183.9 -> 186.4 / 182.5 in
    // the kernel table; B4 N4096 with 64 queries per wave 158.8 / 160.2 -> 162.8 / 163.2 us (SLOWER: the A fragments were already
    // shared by two query blocks there, the extra swaps and the narrower MFMAs cost more than the 14 % of matrix cycles they save)
    // -- so the retile the round-5 verdict asked to be measured is worth 2 % where it helps: MFMA issue does not bound these kernels.
    constexpr bool PV16 = MOS_ATTN_PV16 && D == 40 && !PCOLS && !CAUSAL && NQ == 1;
    f32x16 o[PV16 ? 1 : NQ][PV16 ? 1 : HD<D>::DT];
    f32x4 o16[PV16 ? NQ : 1][2][3];
    float m[NQ], l[NQ];
    const float c = a.scale * LOG2E;
    attend<T, D, NQ, PCOLS, LSUM, CAUSAL, PV16>(kp, a.k_rs, vp, a.v_rs, a.Nkv, c, Ks, Vt, qf, o, o16, m, l, tok, a.n_pcols, cap,
                                                tid, l31, hh, q0);
    if constexpr (PV16) {
        const int n = l31 & 15, kg = (l31 >> 4) + 2 * hh;
#pragma unroll
        for (int iq = 0; iq < NQ; ++iq)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int qi = q0 + 32 * iq + 16 * u + n;
                // row D = 40 of O^T (the ones row of V^T: the row sum) = tile dm 2, row 8 = register 0 of the lanes with kg = 2
                const float lt = __shfl(o16[iq][u][2][0], 32 + n);
                const float mq = __shfl(m[iq], 16 * u + n);
                const float inv = 1.0f / lt;
                if (qi < a.Nq) {
                    T* orow = op + (int64_t)qi * a.o_rs;
#pragma unroll
                    for (int dm = 0; dm < 3; ++dm) {
                        const int d0 = 16 * dm + 4 * kg;
                        if (d0 < D)
                            st8(orow + d0, pack4<T>(o16[iq][u][dm][0] * inv, o16[iq][u][dm][1] * inv, o16[iq][u][dm][2] * inv,
                                                     o16[iq][u][dm][3] * inv));
                    }
                    if (a.lse != nullptr && kg == 0) a.lse[((int64_t)b * a.H + h) * a.Nq + qi] = mq * a.scale + __logf(lt);
                }
            }
    } else {

#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) {
        const int qi = q0 + 32 * iq + l31;
        float lt;
        if constexpr (LSUM) {
            // accumulator row D = 32*(D/32) + (D%32); (D%32) is 8 or 16 -> register 4 or 8 of the lanes with hh = 0
            constexpr int RL = (D % 32) / 8 * 4;
            static_assert((D % 32) % 8 == 0 && D % 32 != 0, "row D must sit in a register of the hh = 0 lanes");
            lt = __shfl(o[iq][D / 32][RL], l31);
        } else {
            lt = l[iq] + __shfl_xor(l[iq], 32);
        }
        const float inv = 1.0f / lt;
        store_out_rows<T, D>(op + (int64_t)qi * a.o_rs, qi < a.Nq, o[iq], inv, hh);
        const int64_t row = ((int64_t)b * a.H + h) * a.Nq + qi;
        if (a.lse != nullptr && hh == 0 && qi < a.Nq) a.lse[row] = m[iq] * a.scale + __logf(lt);
        if constexpr (PCOLS) {
#pragma unroll
            for (int tt = 0; tt < MOS_MAX_PCOLS; ++tt)
                if (tt < a.n_pcols) {
                    float sv = cap[iq][tt];
                    sv = fmaxf(sv, __shfl_xor(sv, 32));
                    if (hh == 0 && qi < a.Nq)
                        a.pcols[row * a.n_pcols + tt] = __builtin_amdgcn_exp2f((sv - m[iq]) * c) * inv;
                }
        }
    }
    }
}

// ---- regional cross-attention: sum over covering sources of attention / count ---------------------
// One LDS residency per pass: the K rows and the transposed V of up to NSP sources (context prompt + regions, 65..96
// keys each: CLIP's 77-token context in three 32-key sub-tiles) are staged together -- every global load of the pass is in
// flight at once (which ones is read off the box table: no vote), ONE barrier, then each wave walks only the sources that
// cover one of ITS 32 queries, with no further
// block-wide synchronisation (the previous kernel walked the sources serially: a full memory round trip, two 64-key
// tiles and three barriers per source, and a block-wide vote before each).
//   * softmax over all keys of a source at once (no running max / rescale);
//   * keys >= Nkv: their K rows are zeros (buffer bounds check) and the S^T accumulator of the last sub-tile starts from a
//     bias of -1e30 for those rows (the C operand of its first MFMA) -> exp2 gives exactly 0 without a compare/select;
//   * d = 40 / 80: row sums from the P.V MFMA (ones row D of V^T), out += O * w/l;  d = 160 / 64 (no spare V^T row):
//     probabilities are normalised and weighted BEFORE P.V, which then accumulates straight into the output;
//   * only the V^T rows that produce kept output rows are staged (D, + the ones row): the A-operand reads of the MFMA row
//     padding run into the next image (the K images sit behind the V^T images) and feed output rows nobody stores.
template <int D> struct RG {
    static constexpr int KEYS = 96;
    static constexpr int TS3 = KEYS + 8;                        // 104 el = 13 x 16 B: odd -> conflict-free b128 reads
    static constexpr bool PRENORM = HD<D>::DV == D;
    static constexpr int VR = PRENORM ? D : D + 1;
    static constexpr int K_ELEMS = KEYS * HD<D>::RS;
    static constexpr int V_ELEMS = VR * TS3;
    // sources resident per pass. Round 6: 2 at d = 40 (was 4): with 2-D query tiles a workgroup rarely needs more than two sources,
    // and 38.6 KB of LDS + 166 VGPRs let THREE workgroups share a CU -- the level-0 grid (768 workgroups at 512x768) runs in one
    // round instead of one and a half: 20.5 -> 17.0 us same box (profiles/r06c9_region_occupancy.txt; the ablation
    // profiles/r06c8_region_ablation.txt puts 14 of the 20 us in the launch / q-load / store skeleton of a 1.5-round grid).
    // d = 80 at two per CU (2 x 67 KB) measured level (15.1 vs 14.6 us) and stays at one.
    static constexpr int NSP = D <= 80 ? 2 : 1;
    static constexpr bool PREFETCH = NSP < 4;                   // next pass's loads fly under this pass's MFMAs
    static constexpr int NK = (KEYS * HD<D>::DCH + 255) / 256;  // 16-byte chunks per thread: K rows
    static constexpr int NV = (KEYS / 2 * HD<D>::DCH + 255) / 256;   // row PAIRS x chunks per thread: V
    static constexpr size_t lds_bytes(size_t es) { return (size_t)NSP * (K_ELEMS + V_ELEMS) * es; }
};

template <typename T, int D>
struct RegionStage {     // registers of one source in flight
    u32x4 k[RG<D>::NK], v0[RG<D>::NV], v1[RG<D>::NV];
};

template <typename T, int D>
__global__ __launch_bounds__(256, (D <= 40 ? 3 : 1)) void region_attn_kernel(AttnArgs a, mos_region_desc reg,
                                                                          const unsigned char* __restrict__ total_count,
                                                                          int accumulate) {
    // total_count / accumulate: a region list longer than one launch holds (MOS_MAX_SOURCES - 1) is walked in chunks. The
    // blend is additive over regions with ONE divisor -- the number of boxes of the WHOLE list that cover the query
    // (pipeline_regionally_t2iadapter.py:60-83: += per region, / count at the end) -- so every chunk weighs its regions with
    // 1 / total_count[q] (a per-query byte map the caller built from all boxes), the first chunk also serves the queries no box
    // covers from the context prompt, and the later chunks add into o.
    typedef typename MT<T>::v8 v8;
    typedef RG<D> G;
    constexpr int KS = HD<D>::KS, DT = HD<D>::DT, RS = HD<D>::RS, DCH = HD<D>::DCH, TS3 = G::TS3, NSP = G::NSP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Vt_ = reinterpret_cast<T*>(smem_raw);                  // [NSP] V^T images, then [NSP] K images
    T* Ks_ = Vt_ + NSP * G::V_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int h = blockIdx.x % a.H;
    const int rest = blockIdx.x / a.H;
    const int qb = rest % a.nqb, b = rest / a.nqb;
    // Round 6: a workgroup's 128 queries are a 2-D TILE of the feature map (8 rows x 16 columns; a wave = 2 rows), not 128
    // consecutive tokens. Regions are boxes: a run of consecutive tokens spans whole rows and so touches every box that shares
    // those rows (the shipped example's three vertical strips: all of them, plus the context prompt -- 49 KB of K / V staged per
    // workgroup at d = 40 for 20 KB of its own q / o), whereas a tile lies inside one or two boxes: which sources a workgroup
    // stages is an exact rectangle test, the context prompt is staged only if some cell of the tile is uncovered, and the small
    // levels (d = 160: one source resident at a time) make one or two passes instead of one per source.
    const int ntx = (reg.feat_w + 15) >> 4;
    const int ty0 = (qb / ntx) * 8, tx0 = (qb - (qb / ntx) * ntx) * 16;
    const int y = ty0 + wave * 2 + (l31 >> 4), x = tx0 + (l31 & 15);
    const bool qvalid = y < reg.feat_h && x < reg.feat_w;
    const int qi = qvalid ? y * reg.feat_w + x : a.Nq;      // (>= Nq: masked everywhere below)
    const int S = reg.n_regions + 1;

    // ---- which sources does this lane / wave / block use ------------------------------------------------------------
    unsigned inbits = 0;
    int cnt = 0;
    for (int r = 0; r < reg.n_regions; ++r) {
        const bool in = (y >= reg.box[r][0] && y < reg.box[r][2] && x >= reg.box[r][1] && x < reg.box[r][3]);
        inbits |= in ? (2u << r) : 0u;
        cnt += in ? 1 : 0;
    }
    if (total_count != nullptr) cnt = total_count[min(qi, a.Nq - 1)];
    if (cnt == 0) inbits = accumulate ? 0u : 1u;
    if (qi >= a.Nq) inbits = 0u;
    const float wreg = cnt > 0 ? 1.f / (float)cnt : 1.f;      // weight of every source this query uses
    unsigned wave_need = 0;
    for (int j = 0; j < S; ++j) wave_need |= (__ballot((inbits >> j) & 1u) != 0ull) ? (1u << j) : 0u;
    // Sources this BLOCK stages: decided from the box table alone (scalar arithmetic, no vote / barrier before the loads are
    // issued): a region is staged when its box meets the tile's rectangle (exact), the context prompt unless ONE box of this
    // launch contains the whole tile (then no cell of it is uncovered; a superset otherwise -- the waves skip per wave below).
    unsigned need = accumulate ? 0u : 1u;
    {
        const int ty1 = min(ty0 + 8, reg.feat_h), tx1 = min(tx0 + 16, reg.feat_w);      // tile = [ty0, ty1) x [tx0, tx1)
        bool covered = false;
        for (int r = 0; r < reg.n_regions; ++r) {
            const bool hit = reg.box[r][0] < ty1 && reg.box[r][2] > ty0 && reg.box[r][1] < tx1 && reg.box[r][3] > tx0;
            need |= hit ? (2u << r) : 0u;
            covered = covered || (reg.box[r][0] <= ty0 && reg.box[r][2] >= ty1 && reg.box[r][1] <= tx0 && reg.box[r][3] >= tx1);
        }
        if (covered) need &= ~1u;      // (also with a whole-list count map: covered by a box of this launch => total count >= 1)
    }
    // constant parts of the images: K pad columns [D, DK) = 0, V^T ones row (row sums), for every resident slot (disjoint
    // from what the staging stores write: made visible by the barrier that follows the first staging)
    for (int sl = 0; sl < NSP; ++sl) {
        T* Ki = Ks_ + sl * G::K_ELEMS;
        if constexpr (HD<D>::DK > D) {
            constexpr int PC = (HD<D>::DK - D) / 8;
            for (int c = tid; c < G::KEYS * PC; c += 256) st16(Ki + (c / PC) * RS + D + (c % PC) * 8, u32x4{0, 0, 0, 0});
        }
        if constexpr (!G::PRENORM) {
            T* ones = Vt_ + sl * G::V_ELEMS + D * TS3;
            if (tid < G::KEYS) ones[tid] = (T)1.0f;
        }
    }

    const T* qp = (const T*)a.q + (int64_t)b * a.q_bs + h * D;
    T* op = (T*)a.o + (int64_t)b * a.o_bs + h * D;
    v8 qf[KS];
    load_row_frags<T, D>(qf, qp + (int64_t)min(qi, a.Nq - 1) * a.q_rs, qi < a.Nq, hh);

    // bias of the last key sub-tile (keys 64..95): -1e30 on the accumulator rows whose key is past Nkv
    f32x16 bias2;
#pragma unroll
    for (int r = 0; r < 16; ++r) bias2[r] = (64 + acc_row(r, hh) >= a.Nkv) ? NEG_BIG : 0.f;

    f32x16 fin[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) fin[dt][r] = 0.f;
    const float c = a.scale * LOG2E;

    // ---- staging --------------------------------------------------------------------------------------------------
    const uint32_t kbytes = slice_bytes<T, D>(a.Nkv, a.k_rs), vbytes = slice_bytes<T, D>(a.Nkv, a.v_rs);
    auto load_src = [&](RegionStage<T, D>& st, int src) __attribute__((always_inline)) {
        const rsrc_t ksrc = make_rsrc((const T*)a.k + (int64_t)src * reg.src_stride + (int64_t)b * a.k_bs + h * D, kbytes);
        const rsrc_t vsrc = make_rsrc((const T*)a.v + (int64_t)src * reg.src_stride + (int64_t)b * a.v_bs + h * D, vbytes);
#pragma unroll
        for (int i = 0; i < G::NK; ++i) {
            const int cc = min(tid + 256 * i, G::KEYS * DCH - 1);
            const int row = cc / DCH, ch = cc - row * DCH;
            st.k[i] = ldbuf16(ksrc, (row * (int)a.k_rs + ch * 8) * (int)sizeof(T));
        }
#pragma unroll
        for (int i = 0; i < G::NV; ++i) {
            const int it = min(tid + 256 * i, G::KEYS / 2 * DCH - 1);
            const int pr = it % (G::KEYS / 2), ch = it / (G::KEYS / 2);
            const int off = (2 * pr * (int)a.v_rs + ch * 8) * (int)sizeof(T);
            st.v0[i] = ldbuf16(vsrc, off);
            st.v1[i] = ldbuf16(vsrc, off + (int)a.v_rs * (int)sizeof(T));
        }
    };
    auto store_src = [&](const RegionStage<T, D>& st, int slot) __attribute__((always_inline)) {
        T* Ki = Ks_ + slot * G::K_ELEMS;
        T* Vi = Vt_ + slot * G::V_ELEMS;
#pragma unroll
        for (int i = 0; i < G::NK; ++i) {
            const int cc = tid + 256 * i;
            const int row = cc / DCH, ch = cc - row * DCH;
            if (G::NK * 256 <= G::KEYS * DCH || cc < G::KEYS * DCH) st16(Ki + row * RS + ch * 8, st.k[i]);
        }
#pragma unroll
        for (int i = 0; i < G::NV; ++i) {
            const int it = tid + 256 * i;
            const int pr = it % (G::KEYS / 2), ch = it / (G::KEYS / 2);
            if (G::NV * 256 <= G::KEYS / 2 * DCH || it < G::KEYS / 2 * DCH) {
                uint32_t* dst = reinterpret_cast<uint32_t*>(Vi + (ch * 8) * TS3 + tr_col(2 * pr));
#pragma unroll
                for (int w = 0; w < 4; ++w) {     // word e = {row 2p+1 elem e (high half), row 2p elem e (low half)}
                    dst[(2 * w) * (TS3 / 2)] = __builtin_amdgcn_perm(st.v1[i][w], st.v0[i][w], 0x05040100u);
                    dst[(2 * w + 1) * (TS3 / 2)] = __builtin_amdgcn_perm(st.v1[i][w], st.v0[i][w], 0x07060302u);
                }
            }
        }
    };
    // ---- one source of the resident pass, for this wave's 32 queries ------------------------------------------------
    auto attend_src = [&](int slot, int src) __attribute__((always_inline)) {
        const T* Ki = Ks_ + slot * G::K_ELEMS;
        const T* Vi = Vt_ + slot * G::V_ELEMS;
        f32x16 sT[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v8 af = as_v8<T>(ld16(Ki + (32 * t + l31) * RS + ks * 16 + hh * 8));
                if (ks == 0) {
                    if (t == 2) sT[t] = MT<T>::mfma32(af, qf[ks], bias2);
                    else {
                        f32x16 z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                        sT[t] = MT<T>::mfma32(af, qf[ks], z);
                    }
                } else {
                    sT[t] = MT<T>::mfma32(af, qf[ks], sT[t]);
                }
            }
        }
        float mx = NEG_BIG;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sT[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mc = mx * c;
        float ls = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pexp = __builtin_amdgcn_exp2f(sT[t][r] * c - mc);
                sT[t][r] = pexp;
                if constexpr (G::PRENORM) ls += pexp;
            }
        const float w = ((inbits >> src) & 1u) ? wreg : 0.f;
        if constexpr (G::PRENORM) {
            ls += __shfl_xor(ls, 32);
            const float mul = w / ls;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) sT[t][r] *= mul;
        }
        v8 pf[3][2];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) pf[t][s2] = acc_to_bfrag<T>(sT[t], s2);
        f32x16 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const T* vrow = Vi + (32 * dt + l31) * TS3;
            bool first = true;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    if (t == 2 && s2 == 1 && a.Nkv <= 80) continue;   // keys 80..95: all past Nkv (wave-uniform)
                    const v8 af = tr_afrag<T>(vrow, t, s2, hh);
                    if constexpr (G::PRENORM) {
                        fin[dt] = MT<T>::mfma32(af, pf[t][s2], fin[dt]);
                    } else {
                        if (first) {
                            f32x16 z;
#pragma unroll
                            for (int r = 0; r < 16; ++r) z[r] = 0.f;
                            o[dt] = MT<T>::mfma32(af, pf[t][s2], z);
                            first = false;
                        } else {
                            o[dt] = MT<T>::mfma32(af, pf[t][s2], o[dt]);
                        }
                    }
                }
        }
        if constexpr (!G::PRENORM) {
            constexpr int RL = (D % 32) / 8 * 4;      // accumulator row D (the ones row) = register RL of the hh = 0 lanes
            static_assert(G::PRENORM || ((D % 32) % 8 == 0 && D % 32 != 0), "ones row must sit in a register of hh = 0");
            const float lt = __shfl(o[D / 32][RL], l31);
            const float mul = w / lt;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) fin[dt][r] += o[dt][r] * mul;
        }
    };

    // ---- passes over the sources this block needs, NSP at a time ----------------------------------------------------
    RegionStage<T, D> stg[NSP];
    int cur_src[NSP], nxt_src[NSP];
    auto take = [&](int (&dst)[NSP]) __attribute__((always_inline)) {      // pop the next <= NSP needed sources
        int n = 0;
#pragma unroll
        for (int i = 0; i < NSP; ++i) {
            dst[i] = -1;
            if (need) { dst[i] = __builtin_ctz(need); need &= need - 1; ++n; }
        }
        return n;
    };
    int ncur = take(cur_src);
#pragma unroll
    for (int i = 0; i < NSP; ++i)
        if (cur_src[i] >= 0) load_src(stg[i], cur_src[i]);
#pragma unroll
    for (int i = 0; i < NSP; ++i)
        if (cur_src[i] >= 0) store_src(stg[i], i);
    __syncthreads();
    while (ncur > 0) {
        int nnxt = 0;
        if constexpr (G::PREFETCH) {
            nnxt = take(nxt_src);
#pragma unroll
            for (int i = 0; i < NSP; ++i)
                if (nxt_src[i] >= 0) load_src(stg[i], nxt_src[i]);
        }
#pragma unroll
        for (int i = 0; i < NSP; ++i)
            if (cur_src[i] >= 0 && ((wave_need >> cur_src[i]) & 1u)) attend_src(i, cur_src[i]);
        if constexpr (!G::PREFETCH) {
            nnxt = take(nxt_src);
            if (nnxt > 0) {
#pragma unroll
                for (int i = 0; i < NSP; ++i)
                    if (nxt_src[i] >= 0) load_src(stg[i], nxt_src[i]);
            }
        }
        if (nnxt > 0) {
            __syncthreads();          // every wave is done reading the resident images
#pragma unroll
            for (int i = 0; i < NSP; ++i)
                if (nxt_src[i] >= 0) store_src(stg[i], i);
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < NSP; ++i) cur_src[i] = nxt_src[i];
        ncur = nnxt;
    }
    if (accumulate) {
        if (qi < a.Nq && inbits != 0u) {      // queries none of this chunk's boxes covers keep what the earlier chunks wrote
            T* orow = op + (int64_t)qi * a.o_rs;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int db = 32 * dt + 8 * r4 + 4 * hh;
                    if (db < D) {
                        const u32x2 old = ld8(orow + db);
                        const T* oh = reinterpret_cast<const T*>(&old);
                        st8(orow + db, pack4<T>((float)oh[0] + fin[dt][4 * r4], (float)oh[1] + fin[dt][4 * r4 + 1],
                                                 (float)oh[2] + fin[dt][4 * r4 + 2], (float)oh[3] + fin[dt][4 * r4 + 3]));
                    }
                }
        }
        return;
    }
    store_out_rows<T, D>(op + (int64_t)qi * a.o_rs, qi < a.Nq, fin, 1.0f, hh);
}

// ---- backward dQ: one wave = 32 queries, loop over key tiles -----------------------------------------
//   S^T = K Q^T ; P^T = exp(scale*S^T - lse) ; dP^T = V dO^T ; dS^T = P^T o (dP^T - D) ; dQ^T += K^T dS^T
// NW waves per block share each staged K / V / K^T tile. NW = 8 (512 threads, two blocks per CU = 4 waves per
// SIMD): a wave's MFMA -> exp/VALU -> MFMA phases are serialised by data dependence, so the pipes only overlap across
// waves; the kernels need < 128 VGPRs at d = 40, and the per-tile staging cost is shared by twice as many rows.
template <typename T, int D, bool PCOLS, int NW, bool CAUSAL = false>
__global__ __launch_bounds__(64 * NW, (D <= 40 ? MOS_DQ_OCC : (D <= 80 && !PCOLS) ? 2 : 1)) void attn_bwd_dq_kernel(AttnBwdArgs a) {
    // (d = 80 with exported probability columns -- 77-key cross attention, latency-bound -- spilled at two waves per SIMD)
    constexpr int NT = 64 * NW;
    constexpr bool PIN = D <= 80 && NW == 4;   // explicit fragment prefetch where the registers allow it
    typedef typename MT<T>::v8 v8;
    constexpr int KS = HD<D>::KS, DT = HD<D>::DT, RS = HD<D>::RS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int RT = HD<D>::ROW_TILE_ELEMS, TT = HD<D>::TR_TILE_ELEMS;
    T* Ks_ = reinterpret_cast<T*>(smem_raw);   // double buffered: [2] K rows, [2] V rows, [2] K^T
    T* Vs_ = Ks_ + 2 * RT;
    T* Kt_ = Vs_ + 2 * RT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int h = blockIdx.x % a.H;
    const int rest = blockIdx.x / a.H;
    const int qb = rest % a.nqb, b = rest / a.nqb;
    const int qi = qb * (32 * NW) + wave * 32 + l31;
    const bool qvalid = qi < a.Nq;
    const int qc = min(qi, a.Nq - 1);

#pragma unroll
    for (int bf = 0; bf < 2; ++bf) {
        zero_row_pads<T, D, NT>(Ks_ + bf * RT, tid);
        zero_row_pads<T, D, NT>(Vs_ + bf * RT, tid);
        zero_tr_pads<T, D, NT>(Kt_ + bf * TT, tid);
    }

    const T* kp = (const T*)a.k + (int64_t)b * a.k_bs + h * D;
    const T* vp = (const T*)a.v + (int64_t)b * a.v_bs + h * D;
    v8 qf[KS], dof[KS];
    load_row_frags<T, D>(qf, (const T*)a.q + (int64_t)b * a.q_bs + (int64_t)qc * a.q_rs + h * D, qvalid, hh);
    load_row_frags<T, D>(dof, (const T*)a.dO + (int64_t)b * a.do_bs + (int64_t)qc * a.do_rs + h * D, qvalid, hh);
    const int64_t row = ((int64_t)b * a.H + h) * a.Nq + qc;
    const float lse2 = a.lse[row] * LOG2E;
    const float c = a.scale * LOG2E;
    int tok[MOS_MAX_PCOLS] = {-1, -1, -1, -1};
    float dpc[MOS_MAX_PCOLS] = {0.f, 0.f, 0.f, 0.f};
    // D = rowsum(dO o O) + sum_t dpcols pcols of this lane's query, formed HERE (round 6: it was a launch of its own -- 44 per SD-1.5
    // training step) from the dO row the lane holds anyway and the O row beside it; the two half-waves of a query hold alternate
    // 8-element chunks of the row. Left in Dvec for the dK/dV kernel, which runs after this one on the same stream.
    float Dq = 0.f;
    {
        v8 of[KS];
        load_row_frags<T, D>(of, (const T*)a.o + (int64_t)b * a.o_bs + (int64_t)qc * a.o_rs + h * D, qvalid, hh);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) Dq += (float)of[ks][e] * (float)dof[ks][e];
        Dq += __shfl_xor(Dq, 32);
    }
    if constexpr (PCOLS) {
#pragma unroll
        for (int tt = 0; tt < MOS_MAX_PCOLS; ++tt)
            if (tt < a.n_pcols) {
                tok[tt] = a.tok_idx[b * a.n_pcols + tt];
                dpc[tt] = a.dpcols[row * a.n_pcols + tt];
                Dq += a.pcols[row * a.n_pcols + tt] * dpc[tt];
            }
    }
    if (qvalid && hh == 0) a.Dvec[row] = Dq;
    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

    RowStage<T, D, NT> vst;
    TrStage<T, D, NT> ktst;      // K: one fetch, both LDS images
    vst.init(a.v_rs, tid);
    ktst.init(a.k_rs, tid);
    const rsrc_t ksrc = make_rsrc(kp, slice_bytes<T, D>(a.Nkv, a.k_rs));
    const rsrc_t vsrc = make_rsrc(vp, slice_bytes<T, D>(a.Nkv, a.v_rs));
    const int k_tile_bytes = KV_TILE * (int)a.k_rs * (int)sizeof(T), v_tile_bytes = KV_TILE * (int)a.v_rs * (int)sizeof(T);
    __syncthreads();
    vst.load(vsrc, 0);
    ktst.load(ksrc, 0);
    ktst.store_rows(Ks_, tid);
    vst.store(Vs_, tid);
    ktst.store(Kt_, tid);
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see the dK/dV kernel
    int cur = 0;
    f32x16 negD16;      // -D of this lane's query in all 16 rows: the C operand of the first dP MFMA of every half tile
#pragma unroll
    for (int r = 0; r < 16; ++r) negD16[r] = -Dq;
    // ragged last key tile = peeled copy of the body (TAIL), causal mask = template parameter: see attend()
    auto tile = [&](const int kv0, auto tail_c) __attribute__((always_inline)) {
        constexpr bool TAIL = decltype(tail_c)::value;
        const T* Ks = Ks_ + cur * RT;
        const T* Vs = Vs_ + cur * RT;
        const T* Kt = Kt_ + cur * TT;
        const bool more = !TAIL && kv0 + KV_TILE < a.Nkv;
        if (more) {  // prefetch the next key tile into registers; written to the other LDS buffer after the MFMAs
            const int nt = kv0 / KV_TILE + 1;
            vst.load(vsrc, nt * v_tile_bytes);
            ktst.load(ksrc, nt * k_tile_bytes);
        }
        // fragment reads one stage ahead of their MFMAs, pinned with scheduling fences (see the dK/dV kernel)
        v8 ak[KS], av[KS];
        auto read_rows = [&](int t) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = (32 * t + l31) * RS + ks * 16 + hh * 8;
                ak[ks] = as_v8<T>(ld16(Ks + off));
                av[ks] = as_v8<T>(ld16(Vs + off));
            }
        };
        read_rows(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            v8 tk[DT][2];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) tk[dt][s2] = tr_afrag<T>(Kt + (32 * dt + l31) * TS, t, s2, hh);
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
            // dP accumulates on top of -D (this lane's query): the MFMA does the subtraction of dS = P o (dP - D)
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = MT<T>::mfma32(ak[ks], qf[ks], s);
                dp = MT<T>::mfma32(av[ks], dof[ks], ks == 0 ? negD16 : dp);
            }
            if constexpr (TAIL) {  // ragged last tile only: keys past Nkv get probability 0
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kv0 + 32 * t + acc_row(r, hh) >= a.Nkv) s[r] = NEG_BIG;
            }
            if constexpr (CAUSAL) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kv0 + 32 * t + acc_row(r, hh) > qi) s[r] = NEG_BIG;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[r] * c - lse2);
                float g = dp[r];
                if constexpr (PCOLS) {
                    const int kvi = kv0 + 32 * t + acc_row(r, hh);
#pragma unroll
                    for (int tt = 0; tt < MOS_MAX_PCOLS; ++tt)
                        if (kvi == tok[tt]) g += dpc[tt];
                }
                s[r] = p * g;
            }
            v8 dsf[2];
            dsf[0] = acc_to_bfrag<T>(s, 0);
            dsf[1] = acc_to_bfrag<T>(s, 1);
            if (t == 0) read_rows(1);
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) dq[dt] = MT<T>::mfma32(tk[dt][s2], dsf[s2], dq[dt]);
        }
        if (more) {
            ktst.store_rows(Ks_ + (cur ^ 1) * RT, tid);
            vst.store(Vs_ + (cur ^ 1) * RT, tid);
            ktst.store(Kt_ + (cur ^ 1) * TT, tid);
        }
        __syncthreads();
        cur ^= 1;
    };
    int kv0 = 0;
    for (; kv0 + KV_TILE <= a.Nkv; kv0 += KV_TILE) tile(kv0, std::false_type{});
    if (kv0 < a.Nkv) tile(kv0, std::true_type{});
    T* dqp = (T*)a.dq + (int64_t)b * a.dq_bs + (int64_t)qi * a.dq_rs + h * D;
    store_out_rows<T, D>(dqp, qvalid, dq, a.scale, hh);
}

// ---- backward dK/dV: one wave = 32 keys, loop over query tiles of this split ------------------------
//   S = Q K^T ; P = exp(scale*S - lse) ; dV^T += dO^T P ; dP = dO V^T ; dS = P o (dP - D) ; dK^T += Q^T dS
template <typename T, int D, bool PCOLS, int NW, bool CAUSAL = false>
__global__ __launch_bounds__(64 * NW, (D <= 40 ? 2 : 1)) void attn_bwd_dkdv_kernel(AttnBwdArgs a) {
    constexpr int NT = 64 * NW;
    typedef typename MT<T>::v8 v8;
    constexpr int KS = HD<D>::KS, DT = HD<D>::DT, RS = HD<D>::RS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int RT = HD<D>::ROW_TILE_ELEMS, TT = HD<D>::TR_TILE_ELEMS;
    constexpr int NB = DKDV_NBUF<D>::value;    // 2 = double buffered (fits in 160 KB for d <= 80), 1 = single
    constexpr int ST = KV_TILE * (2 + MOS_MAX_PCOLS);   // floats per stats buffer: lse[64], D[64], dpc[64][4]
    T* Qs_ = reinterpret_cast<T*>(smem_raw);
    T* dOs_ = Qs_ + NB * RT;
    T* Qt_ = dOs_ + NB * RT;
    T* dOt_ = Qt_ + NB * TT;
    float* stat_ = reinterpret_cast<float*>(dOt_ + NB * TT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int h = blockIdx.x % a.H;
    const int rest = blockIdx.x / a.H;
    const int kb = rest % a.nkb, b = rest / a.nkb;
    const int split = blockIdx.y;
    const int kvi = kb * (32 * NW) + wave * 32 + l31;
    const bool kvalid = kvi < a.Nkv;
    const int kc = min(kvi, a.Nkv - 1);

#pragma unroll
    for (int bf = 0; bf < NB; ++bf) {
        zero_row_pads<T, D, NT>(Qs_ + bf * RT, tid);
        zero_row_pads<T, D, NT>(dOs_ + bf * RT, tid);
        zero_tr_pads<T, D, NT>(Qt_ + bf * TT, tid);
        zero_tr_pads<T, D, NT>(dOt_ + bf * TT, tid);
    }

    v8 kf[KS], vf[KS];
    load_row_frags<T, D>(kf, (const T*)a.k + (int64_t)b * a.k_bs + (int64_t)kc * a.k_rs + h * D, kvalid, hh);
    load_row_frags<T, D>(vf, (const T*)a.v + (int64_t)b * a.v_bs + (int64_t)kc * a.v_rs + h * D, kvalid, hh);
    const float c = a.scale * LOG2E;
    int mytok = -1;  // index t of the exported column this lane's key corresponds to, if any
    if constexpr (PCOLS) {
#pragma unroll
        for (int tt = 0; tt < MOS_MAX_PCOLS; ++tt)
            if (tt < a.n_pcols && a.tok_idx[b * a.n_pcols + tt] == kvi) mytok = tt;
    }
    f32x16 dkT[DT], dvT[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkT[dt][r] = 0.f; dvT[dt][r] = 0.f; }

    const T* qp = (const T*)a.q + (int64_t)b * a.q_bs + h * D;
    const T* dop = (const T*)a.dO + (int64_t)b * a.do_bs + h * D;
    const int64_t rowbase = ((int64_t)b * a.H + h) * a.Nq;
    const int qbeg = split * a.q_per_split;
    const int qend = min(qbeg + a.q_per_split, a.Nq);

    TrStage<T, D, NT> qtst, dotst;   // one fetch per tensor, both LDS images
    qtst.init(a.q_rs, tid);
    dotst.init(a.do_rs, tid);
    float st_lse = 0.f, st_D = 0.f, st_dpc[MOS_MAX_PCOLS] = {0.f, 0.f, 0.f, 0.f};
    int st_nv = 0;
    // rows in [qend, tile end) are always >= Nq (splits are whole tiles), i.e. outside the descriptors: zeros
    const rsrc_t qsrc = make_rsrc(qp, slice_bytes<T, D>(a.Nq, a.q_rs));
    const rsrc_t dosrc = make_rsrc(dop, slice_bytes<T, D>(a.Nq, a.do_rs));
    const int q_row_bytes = (int)a.q_rs * (int)sizeof(T), do_row_bytes = (int)a.do_rs * (int)sizeof(T);
    auto load_tile = [&](int q0) {
        const int nv = qend - q0;
        qtst.load(qsrc, q0 * q_row_bytes);
        dotst.load(dosrc, q0 * do_row_bytes);
        st_nv = nv;
        const int64_t rr = rowbase + q0 + min(tid & (KV_TILE - 1), nv - 1);   // unconditional, clamped
        st_lse = a.lse[rr];
        st_D = a.Dvec[rr];
        if constexpr (PCOLS) {
#pragma unroll
            for (int tt = 0; tt < MOS_MAX_PCOLS; ++tt)
                st_dpc[tt] = a.dpcols[rr * a.n_pcols + min(tt, a.n_pcols - 1)];
        }
    };
    auto store_tile = [&](int bf) {
        qtst.store_rows(Qs_ + bf * RT, tid);
        dotst.store_rows(dOs_ + bf * RT, tid);
        qtst.store(Qt_ + bf * TT, tid);
        dotst.store(dOt_ + bf * TT, tid);
        if (tid < KV_TILE) {
            float* sb = stat_ + bf * ST;
            const bool ok = tid < st_nv;
            sb[tid] = ok ? st_lse * LOG2E : 0.f;
            sb[KV_TILE + tid] = ok ? -st_D : 0.f;     // stored negated: it is the initial value of the dP accumulator
            if constexpr (PCOLS) {
#pragma unroll
                for (int tt = 0; tt < MOS_MAX_PCOLS; ++tt)
                    sb[2 * KV_TILE + tid * MOS_MAX_PCOLS + tt] = (ok && tt < a.n_pcols) ? st_dpc[tt] : 0.f;
            }
        }
    };
    __syncthreads();
    if (qbeg < qend) {
        load_tile(qbeg);
        store_tile(0);
    }
    __syncthreads();
    // nothing may be pending on the vector-memory counter when the loop is entered: with the K/V (Q/dO) fragment
    // loads of the prologue still "in flight" on some path, hipcc's waitcnt pass guards their first use INSIDE the loop
    // with vmcnt(N), and because that counter retires in order, the wait also covers the tile prefetch issued just
    // before it -- a full memory round trip in front of the first MFMAs of every tile
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) (expcnt / lgkmcnt untouched)
    int cur = 0;
    for (int q0 = qbeg; q0 < qend; q0 += KV_TILE) {
        const T* Qs = Qs_ + cur * RT;
        const T* dOs = dOs_ + cur * RT;
        const T* Qt = Qt_ + cur * TT;
        const T* dOt = dOt_ + cur * TT;
        const float* lse_s = stat_ + cur * ST;
        const float* D_s = lse_s + KV_TILE;
        const float* dpc_s = D_s + KV_TILE;
        const bool more = q0 + KV_TILE < qend;
        if (more) load_tile(q0 + KV_TILE);
        // Fragment reads are issued one stage ahead of their MFMAs and pinned there with scheduling fences: left
        // alone, hipcc sinks each ds_read next to its consumer (`ds_read; s_waitcnt lgkmcnt(0); v_mfma`), which puts
        // one LDS round trip in front of nearly every MFMA.
        v8 aq[KS], ado[KS];
        f32x4 l4[4];
        f32x16 d16;     // -D of this half tile's 16 accumulator rows: the C operand of the first dP MFMA
        auto read_rows = [&](int t) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = (32 * t + l31) * RS + ks * 16 + hh * 8;
                aq[ks] = as_v8<T>(ld16(Qs + off));
                ado[ks] = as_v8<T>(ld16(dOs + off));
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int ql = 32 * t + 8 * r4 + 4 * hh;
                l4[r4] = *reinterpret_cast<const f32x4*>(lse_s + ql);
                const f32x4 dd = *reinterpret_cast<const f32x4*>(D_s + ql);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) d16[4 * r4 + rr] = dd[rr];
            }
        };
        read_rows(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // transposed fragments of this half: consumed after the softmax, in flight during S / dP / exp
            v8 tdo[DT][2], tq[DT][2];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    tdo[dt][s2] = tr_afrag<T>(dOt + (32 * dt + l31) * TS, t, s2, hh);
                    tq[dt][s2] = tr_afrag<T>(Qt + (32 * dt + l31) * TS, t, s2, hh);
                }
            if constexpr (D <= 80) __builtin_amdgcn_sched_barrier(0);   // d = 160: the fragments would spill
            // dP accumulates on top of -D (rows = queries: d4 holds exactly this lane's 16 accumulator rows), so the MFMA
            // does the subtraction of dS = P o (dP - D)
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = MT<T>::mfma32(aq[ks], kf[ks], s);
                dp = MT<T>::mfma32(ado[ks], vf[ks], ks == 0 ? d16 : dp);
            }
            // accumulator rows are query-local indices 32t + acc_row(r, hh); column = this lane's key
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = 4 * r4 + rr;
                    // keys past Nkv (lanes of the last key block) need no masking: each lane's key is one column of
                    // dK^T / dV^T, columns never mix, and the store skips invalid keys
                    float p, g = dp[r];
                    p = __builtin_amdgcn_exp2f(s[r] * c - l4[r4][rr]);
                    if constexpr (CAUSAL) {
                        if (kvi > q0 + 32 * t + 8 * r4 + 4 * hh + rr) p = 0.f;   // this lane's key vs the row's query
                    }
                    if constexpr (PCOLS) {
                        if (mytok >= 0) g += dpc_s[(32 * t + 8 * r4 + 4 * hh + rr) * MOS_MAX_PCOLS + mytok];
                    }
                    s[r] = p;
                    dp[r] = p * g;
                }
            }
            v8 pf[2], dsf[2];
            pf[0] = acc_to_bfrag<T>(s, 0); pf[1] = acc_to_bfrag<T>(s, 1);
            dsf[0] = acc_to_bfrag<T>(dp, 0); dsf[1] = acc_to_bfrag<T>(dp, 1);
            if (t == 0) read_rows(1);   // next half's row fragments fly under the 8 MFMAs below
            if constexpr (D <= 80) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    dvT[dt] = MT<T>::mfma32(tdo[dt][s2], pf[s2], dvT[dt]);
                    dkT[dt] = MT<T>::mfma32(tq[dt][s2], dsf[s2], dkT[dt]);
                }
        }
        if constexpr (NB == 1) __syncthreads();   // single buffer: every wave is done reading before the overwrite
        if (more) store_tile(NB == 2 ? (cur ^ 1) : 0);
        __syncthreads();
        if constexpr (NB == 2) cur ^= 1;
    }
    if (a.nsplit == 1) {
        T* dkp = (T*)a.dk + (int64_t)b * a.dk_bs + (int64_t)kvi * a.dk_rs + h * D;
        T* dvp = (T*)a.dv + (int64_t)b * a.dv_bs + (int64_t)kvi * a.dv_rs + h * D;
        store_out_rows<T, D>(dkp, kvalid, dkT, a.scale, hh);
        store_out_rows<T, D>(dvp, kvalid, dvT, 1.0f, hh);
    } else if (kvalid) {
        const int64_t bh = (int64_t)b * a.H + h;
        const int64_t slab = (int64_t)a.B * a.H * a.Nkv * D;
        float* pk = a.part + ((int64_t)split * slab) + (bh * a.Nkv + kvi) * D;
        float* pv = pk + (int64_t)a.nsplit * slab;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int db = 32 * dt + 8 * r4 + 4 * hh;
                if (db < D) {
                    *reinterpret_cast<f32x4*>(pk + db) = f32x4{dkT[dt][4 * r4] * a.scale, dkT[dt][4 * r4 + 1] * a.scale,
                                                              dkT[dt][4 * r4 + 2] * a.scale, dkT[dt][4 * r4 + 3] * a.scale};
                    *reinterpret_cast<f32x4*>(pv + db) = f32x4{dvT[dt][4 * r4], dvT[dt][4 * r4 + 1], dvT[dt][4 * r4 + 2],
                                                              dvT[dt][4 * r4 + 3]};
                }
            }
    }
}

// sum split partials -> dK, dV in T with output strides
template <typename T, int D>
__global__ void attn_bwd_reduce_kernel(const float* __restrict__ part, int nsplit, T* __restrict__ dk, int64_t dk_bs,
                                       int64_t dk_rs, T* __restrict__ dv, int64_t dv_bs, int64_t dv_rs, int B, int H,
                                       int Nkv) {
    const int64_t slab = (int64_t)B * H * Nkv * D;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over slab/4
    if (idx * 4 >= slab) return;
    const int64_t e = idx * 4;
    const int d = e % D;
    const int64_t r = e / D;
    const int kv = r % Nkv;
    const int64_t bh = r / Nkv;
    const int h = bh % H, b = bh / H;
    f32x4 sk = {0.f, 0.f, 0.f, 0.f}, sv = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsplit; ++s) {
        sk += *reinterpret_cast<const f32x4*>(part + (int64_t)s * slab + e);
        sv += *reinterpret_cast<const f32x4*>(part + ((int64_t)nsplit + s) * slab + e);
    }
    st8(dk + (int64_t)b * dk_bs + (int64_t)kv * dk_rs + h * D + d, pack4<T>(sk[0], sk[1], sk[2], sk[3]));
    st8(dv + (int64_t)b * dv_bs + (int64_t)kv * dv_rs + h * D + d, pack4<T>(sv[0], sv[1], sv[2], sv[3]));
}

// ---- dK/dV, slot-interleaved (round 4) ---------------------------------------------------------------------------------------
// The ablation of attn_bwd_dkdv_kernel (profiles/r03_attention_ablation.txt: 185 us of MFMA + LDS reads, 62 us of softmax VALU,
// 55 us of staging, 17 us of barriers that ADD UP) names the fault: per 32-query half the kernel runs 6 MFMAs, then ~300 cycles
// of VALU (exp2, p * (dP - D), two roundings), then 8 MFMAs, each phase waiting for the previous one. Only VALU issued by the
// SAME wave behind an MFMA overlaps with it, so the tile is laid out as 28 MFMA slots in program order
//     A0 (6)            S, dP of queries 0-31
//     A1 (6)  || P, dS of queries 0-31, accumulator rows 0-7           (needs A0)
//     C0a (4) || P, dS of queries 0-31, rows 8-15                      (needs the first eight)
//     C0b (4) || P, dS of queries 32-63, rows 0-7                      (needs A1)
//     C1a (4) || P, dS of queries 32-63, rows 8-15
//     C1b (4) || the transposing LDS stores of the next tile
// with 1-2 elements of softmax work behind every MFMA and an empty volatile asm after each slot that the accumulators and the
// in-place S / dP vectors pass through (the compiler can neither bunch the MFMAs nor move the exponentials out of their slot).
// d = 40, no probability columns, no causal mask, whole 64-query tiles: the level-0 self-attention of the UNet; everything else
// takes attn_bwd_dkdv_kernel.
template <typename T, int D>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_pipe_kernel(AttnBwdArgs a) {
    constexpr int NT = 256, NW = 4;
    typedef typename MT<T>::v8 v8;
    constexpr int KS = HD<D>::KS, DT = HD<D>::DT, RS = HD<D>::RS;
    static_assert(KS == 3 && DT == 2, "the slot table is written for d = 40");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int RT = HD<D>::ROW_TILE_ELEMS, TT = HD<D>::TR_TILE_ELEMS;
    constexpr int ST = KV_TILE * (2 + MOS_MAX_PCOLS);   // (same LDS layout as attn_bwd_dkdv_kernel: dkdv_lds<D>)
    T* Qs_ = reinterpret_cast<T*>(smem_raw);
    T* dOs_ = Qs_ + 2 * RT;
    T* Qt_ = dOs_ + 2 * RT;
    T* dOt_ = Qt_ + 2 * TT;
    float* stat_ = reinterpret_cast<float*>(dOt_ + 2 * TT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int h = blockIdx.x % a.H;
    const int rest = blockIdx.x / a.H;
    const int kb = rest % a.nkb, b = rest / a.nkb;
    const int split = blockIdx.y;
    const int kvi = kb * (32 * NW) + wave * 32 + l31;
    const bool kvalid = kvi < a.Nkv;
    const int kc = min(kvi, a.Nkv - 1);
#pragma unroll
    for (int bf = 0; bf < 2; ++bf) {
        zero_row_pads<T, D, NT>(Qs_ + bf * RT, tid);
        zero_row_pads<T, D, NT>(dOs_ + bf * RT, tid);
        zero_tr_pads<T, D, NT>(Qt_ + bf * TT, tid);
        zero_tr_pads<T, D, NT>(dOt_ + bf * TT, tid);
    }
    v8 kf[KS], vf[KS];
    load_row_frags<T, D>(kf, (const T*)a.k + (int64_t)b * a.k_bs + (int64_t)kc * a.k_rs + h * D, kvalid, hh);
    load_row_frags<T, D>(vf, (const T*)a.v + (int64_t)b * a.v_bs + (int64_t)kc * a.v_rs + h * D, kvalid, hh);
    const float c = a.scale * LOG2E;
    f32x16 dkT[DT], dvT[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkT[dt][r] = 0.f; dvT[dt][r] = 0.f; }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const T* qp = (const T*)a.q + (int64_t)b * a.q_bs + h * D;
    const T* dop = (const T*)a.dO + (int64_t)b * a.do_bs + h * D;
    const int64_t rowbase = ((int64_t)b * a.H + h) * a.Nq;
    const int qbeg = split * a.q_per_split;
    const int qend = min(qbeg + a.q_per_split, a.Nq);
    TrStage<T, D, NT> qtst, dotst;
    qtst.init(a.q_rs, tid);
    dotst.init(a.do_rs, tid);
    float st_lse = 0.f, st_D = 0.f;
    int st_nv = 0;
    const rsrc_t qsrc = make_rsrc(qp, slice_bytes<T, D>(a.Nq, a.q_rs));
    const rsrc_t dosrc = make_rsrc(dop, slice_bytes<T, D>(a.Nq, a.do_rs));
    const int q_row_bytes = (int)a.q_rs * (int)sizeof(T), do_row_bytes = (int)a.do_rs * (int)sizeof(T);
    auto load_tile = [&](int q0) {
        const int nv = qend - q0;
        qtst.load(qsrc, q0 * q_row_bytes);
        dotst.load(dosrc, q0 * do_row_bytes);
        st_nv = nv;
        const int64_t rr = rowbase + q0 + min(tid & (KV_TILE - 1), nv - 1);
        st_lse = a.lse[rr];
        st_D = a.Dvec[rr];
    };
    auto store_tile = [&](int bf) {
        qtst.store_rows(Qs_ + bf * RT, tid);
        dotst.store_rows(dOs_ + bf * RT, tid);
        qtst.store(Qt_ + bf * TT, tid);
        dotst.store(dOt_ + bf * TT, tid);
        if (tid < KV_TILE) {
            float* sb = stat_ + bf * ST;
            const bool ok = tid < st_nv;
            sb[tid] = ok ? st_lse * LOG2E : 0.f;
            sb[KV_TILE + tid] = ok ? -st_D : 0.f;     // negated: the initial value of the dP accumulator
        }
    };
    __syncthreads();
    if (qbeg < qend) {
        load_tile(qbeg);
        store_tile(0);
    }
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), see attn_bwd_dkdv_kernel
    int cur = 0;
    for (int q0 = qbeg; q0 < qend; q0 += KV_TILE) {
        const T* Qs = Qs_ + cur * RT;
        const T* dOs = dOs_ + cur * RT;
        const T* Qt = Qt_ + cur * TT;
        const T* dOt = dOt_ + cur * TT;
        const float* lse_s = stat_ + cur * ST;
        const float* D_s = lse_s + KV_TILE;
        const bool more = q0 + KV_TILE < qend;
        if (more) load_tile(q0 + KV_TILE);
        f32x4 l0[4], l1[4];
        f32x16 nd0, nd1;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int ql = 8 * r4 + 4 * hh;
            l0[r4] = *reinterpret_cast<const f32x4*>(lse_s + ql);
            l1[r4] = *reinterpret_cast<const f32x4*>(lse_s + 32 + ql);
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(D_s + ql), d1 = *reinterpret_cast<const f32x4*>(D_s + 32 + ql);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) { nd0[4 * r4 + rr] = d0[rr]; nd1[4 * r4 + rr] = d1[rr]; }
        }
        f32x16 s0, dp0, s1, dp1;
        v8 pf0[2], dsf0[2], pf1[2], dsf1[2];
        // A-operand fragments are read a whole stage ahead of the MFMAs that consume them and ride through the fences (read inside
        // their slot, every MFMA would wait on LDS): the row fragments of both halves now, the transposed ones stage by stage.
        v8 aq0[KS], ado0[KS], aq1[KS], ado1[KS], fdo00[DT], fq00[DT], fdo01[DT], fq01[DT], fdo10[DT], fq10[DT], fdo11[DT], fq11[DT];
        #pragma unroll
        for (int ks = 0; ks < KS; ++ks) { aq0[ks] = as_v8<T>(ld16(Qs + (32 * 0 + l31) * RS + ks * 16 + hh * 8)); ado0[ks] = as_v8<T>(ld16(dOs + (32 * 0 + l31) * RS + ks * 16 + hh * 8)); }
        #pragma unroll
        for (int ks = 0; ks < KS; ++ks) { aq1[ks] = as_v8<T>(ld16(Qs + (32 * 1 + l31) * RS + ks * 16 + hh * 8)); ado1[ks] = as_v8<T>(ld16(dOs + (32 * 1 + l31) * RS + ks * 16 + hh * 8)); }
        asm volatile("" : "+v"(aq0[0]), "+v"(ado0[0]), "+v"(aq0[1]), "+v"(ado0[1]), "+v"(aq0[2]), "+v"(ado0[2]));
        // A0: S and dP of queries 0-31 (nothing to hide behind them yet)
        s0 = MT<T>::mfma32(aq0[0], kf[0], zero16);
        dp0 = MT<T>::mfma32(ado0[0], vf[0], nd0);
        s0 = MT<T>::mfma32(aq0[1], kf[1], s0);
        dp0 = MT<T>::mfma32(ado0[1], vf[1], dp0);
        s0 = MT<T>::mfma32(aq0[2], kf[2], s0);
        dp0 = MT<T>::mfma32(ado0[2], vf[2], dp0);
        asm volatile("" : "+v"(s0), "+v"(dp0), "+v"(aq1[0]), "+v"(ado1[0]), "+v"(aq1[1]), "+v"(ado1[1]), "+v"(aq1[2]), "+v"(ado1[2]));
        // A1: S and dP of queries 32-63   || P, dS of queries 0-31, accumulator rows 0-7
        #pragma unroll
        for (int dt = 0; dt < DT; ++dt) { fdo00[dt] = tr_afrag<T>(dOt + (32 * dt + l31) * TS, 0, 0, hh); fq00[dt] = tr_afrag<T>(Qt + (32 * dt + l31) * TS, 0, 0, hh); }
        s1 = MT<T>::mfma32(aq1[0], kf[0], zero16);
        { const float p = __builtin_amdgcn_exp2f(s0[0] * c - l0[0][0]); s0[0] = p; dp0[0] = p * dp0[0]; }
        { const float p = __builtin_amdgcn_exp2f(s0[1] * c - l0[0][1]); s0[1] = p; dp0[1] = p * dp0[1]; }
        asm volatile("" : "+v"(s1), "+v"(s0), "+v"(dp0));
        dp1 = MT<T>::mfma32(ado1[0], vf[0], nd1);
        { const float p = __builtin_amdgcn_exp2f(s0[2] * c - l0[0][2]); s0[2] = p; dp0[2] = p * dp0[2]; }
        asm volatile("" : "+v"(s1), "+v"(dp1), "+v"(s0), "+v"(dp0));
        s1 = MT<T>::mfma32(aq1[1], kf[1], s1);
        { const float p = __builtin_amdgcn_exp2f(s0[3] * c - l0[0][3]); s0[3] = p; dp0[3] = p * dp0[3]; }
        asm volatile("" : "+v"(s1), "+v"(dp1), "+v"(s0), "+v"(dp0));
        dp1 = MT<T>::mfma32(ado1[1], vf[1], dp1);
        { const float p = __builtin_amdgcn_exp2f(s0[4] * c - l0[1][0]); s0[4] = p; dp0[4] = p * dp0[4]; }
        { const float p = __builtin_amdgcn_exp2f(s0[5] * c - l0[1][1]); s0[5] = p; dp0[5] = p * dp0[5]; }
        asm volatile("" : "+v"(s1), "+v"(dp1), "+v"(s0), "+v"(dp0));
        s1 = MT<T>::mfma32(aq1[2], kf[2], s1);
        { const float p = __builtin_amdgcn_exp2f(s0[6] * c - l0[1][2]); s0[6] = p; dp0[6] = p * dp0[6]; }
        asm volatile("" : "+v"(s1), "+v"(dp1), "+v"(s0), "+v"(dp0));
        dp1 = MT<T>::mfma32(ado1[2], vf[2], dp1);
        { const float p = __builtin_amdgcn_exp2f(s0[7] * c - l0[1][3]); s0[7] = p; dp0[7] = p * dp0[7]; }
        pf0[0] = acc_to_bfrag<T>(s0, 0); dsf0[0] = acc_to_bfrag<T>(dp0, 0);
        asm volatile("" : "+v"(s1), "+v"(dp1), "+v"(s0), "+v"(dp0), "+v"(pf0[0]), "+v"(dsf0[0]), "+v"(fdo00[0]), "+v"(fq00[0]), "+v"(fdo00[1]), "+v"(fq00[1]));
        // C0a: dV^T += dO^T P, dK^T += Q^T dS over queries 0-31, contraction rows 0-15   || P, dS rows 8-15 of queries 0-31
        #pragma unroll
        for (int dt = 0; dt < DT; ++dt) { fdo01[dt] = tr_afrag<T>(dOt + (32 * dt + l31) * TS, 0, 1, hh); fq01[dt] = tr_afrag<T>(Qt + (32 * dt + l31) * TS, 0, 1, hh); }
        dvT[0] = MT<T>::mfma32(fdo00[0], pf0[0], dvT[0]);
        { const float p = __builtin_amdgcn_exp2f(s0[8] * c - l0[2][0]); s0[8] = p; dp0[8] = p * dp0[8]; }
        { const float p = __builtin_amdgcn_exp2f(s0[9] * c - l0[2][1]); s0[9] = p; dp0[9] = p * dp0[9]; }
        asm volatile("" : "+v"(dvT[0]), "+v"(dkT[0]), "+v"(dvT[1]), "+v"(dkT[1]), "+v"(s0), "+v"(dp0));
        dkT[0] = MT<T>::mfma32(fq00[0], dsf0[0], dkT[0]);
        { const float p = __builtin_amdgcn_exp2f(s0[10] * c - l0[2][2]); s0[10] = p; dp0[10] = p * dp0[10]; }
        { const float p = __builtin_amdgcn_exp2f(s0[11] * c - l0[2][3]); s0[11] = p; dp0[11] = p * dp0[11]; }
        asm volatile("" : "+v"(dvT[0]), "+v"(dkT[0]), "+v"(dvT[1]), "+v"(dkT[1]), "+v"(s0), "+v"(dp0));
        dvT[1] = MT<T>::mfma32(fdo00[1], pf0[0], dvT[1]);
        { const float p = __builtin_amdgcn_exp2f(s0[12] * c - l0[3][0]); s0[12] = p; dp0[12] = p * dp0[12]; }
        { const float p = __builtin_amdgcn_exp2f(s0[13] * c - l0[3][1]); s0[13] = p; dp0[13] = p * dp0[13]; }
        asm volatile("" : "+v"(dvT[0]), "+v"(dkT[0]), "+v"(dvT[1]), "+v"(dkT[1]), "+v"(s0), "+v"(dp0));
        dkT[1] = MT<T>::mfma32(fq00[1], dsf0[0], dkT[1]);
        { const float p = __builtin_amdgcn_exp2f(s0[14] * c - l0[3][2]); s0[14] = p; dp0[14] = p * dp0[14]; }
        { const float p = __builtin_amdgcn_exp2f(s0[15] * c - l0[3][3]); s0[15] = p; dp0[15] = p * dp0[15]; }
        pf0[1] = acc_to_bfrag<T>(s0, 1); dsf0[1] = acc_to_bfrag<T>(dp0, 1);
        asm volatile("" : "+v"(dvT[0]), "+v"(dkT[0]), "+v"(dvT[1]), "+v"(dkT[1]), "+v"(s0), "+v"(dp0), "+v"(pf0[1]), "+v"(dsf0[1]), "+v"(s1), "+v"(dp1), "+v"(fdo01[0]), "+v"(fq01[0]), "+v"(fdo01[1]), "+v"(fq01[1]));
        // C0b: the second contraction half of queries 0-31   || P, dS rows 0-7 of queries 32-63
        #pragma unroll
        for (int dt = 0; dt < DT; ++dt) { fdo10[dt] = tr_afrag<T>(dOt + (32 * dt + l31) * TS, 1, 0, hh); fq10[dt] = tr_afrag<T>(Qt + (32 * dt + l31) * TS, 1, 0, hh); }
        dvT[0] = MT<T>::mfma32(fdo01[0], pf0[1], dvT[0]);
        { const float p = __builtin_amdgcn_exp2f(s1[0] * c - l1[0][0]); s1[0] = p; dp1[0] = p * dp1[0]; }
        { const float p = __builtin_amdgcn_exp2f(s1[1] * c - l1[0][1]); s1[1] = p; dp1[1] = p * dp1[1]; }
        asm volatile("" : "+v"(dvT[0]), "+v"(dkT[0]), "+v"(dvT[1]), "+v"(dkT[1]), "+v"(s1), "+v"(dp1));
        dkT[0] = MT<T>::mfma32(fq01[0], dsf0[1], dkT[0]);
        { const float p = __builtin_amdgcn_exp2f(s1[2] * c - l1[0][2]); s1[2] = p; dp1[2] = p * dp1[2]; }
        { const float p = __builtin_amdgcn_exp2f(s1[3] * c - l1[0][3]); s1[3] = p; dp1[3] = p * dp1[3]; }
        asm volatile("" : "+v"(dvT[0]), "+v"(dkT[0]), "+v"(dvT[1]), "+v"(dkT[1]), "+v"(s1), "+v"(dp1));
        dvT[1] = MT<T>::mfma32(fdo01[1], pf0[1], dvT[1]);
        { const float p = __builtin_amdgcn_exp2f(s1[4] * c - l1[1][0]); s1[4] = p; dp1[4] = p * dp1[4]; }
        { const float p = __builtin_amdgcn_exp2f(s1[5] * c - l1[1][1]); s1[5] = p; dp1[5] = p * dp1[5]; }
        asm volatile("" : "+v"(dvT[0]), "+v"(dkT[0]), "+v"(dvT[1]), "+v"(dkT[1]), "+v"(s1), "+v"(dp1));
        dkT[1] = MT<T>::mfma32(fq01[1], dsf0[1], dkT[1]);
        { const float p = __builtin_amdgcn_exp2f(s1[6] * c - l1[1][2]); s1[6] = p; dp1[6] = p * dp1[6]; }
        { const float p = __builtin_amdgcn_exp2f(s1[7] * c - l1[1][3]); s1[7] = p; dp1[7] = p * dp1[7]; }
        pf1[0] = acc_to_bfrag<T>(s1, 0); dsf1[0] = acc_to_bfrag<T>(dp1, 0);
        asm volatile("" : "+v"(dvT[0]), "+v"(dkT[0]), "+v"(dvT[1]), "+v"(dkT[1]), "+v"(s1), "+v"(dp1), "+v"(pf1[0]), "+v"(dsf1[0]), "+v"(fdo10[0]), "+v"(fq10[0]), "+v"(fdo10[1]), "+v"(fq10[1]));
        // C1a: queries 32-63, contraction rows 0-15   || P, dS rows 8-15 of queries 32-63
        #pragma unroll
        for (int dt = 0; dt < DT; ++dt) { fdo11[dt] = tr_afrag<T>(dOt + (32 * dt + l31) * TS, 1, 1, hh); fq11[dt] = tr_afrag<T>(Qt + (32 * dt + l31) * TS, 1, 1, hh); }
        dvT[0] = MT<T>::mfma32(fdo10[0], pf1[0], dvT[0]);
        { const float p = __builtin_amdgcn_exp2f(s1[8] * c - l1[2][0]); s1[8] = p; dp1[8] = p * dp1[8]; }
        { const float p = __builtin_amdgcn_exp2f(s1[9] * c - l1[2][1]); s1[9] = p; dp1[9] = p * dp1[9]; }
        asm volatile("" : "+v"(dvT[0]), "+v"(dkT[0]), "+v"(dvT[1]), "+v"(dkT[1]), "+v"(s1), "+v"(dp1));
        dkT[0] = MT<T>::mfma32(fq10[0], dsf1[0], dkT[0]);
        { const float p = __builtin_amdgcn_exp2f(s1[10] * c - l1[2][2]); s1[10] = p; dp1[10] = p * dp1[10]; }
        { const float p = __builtin_amdgcn_exp2f(s1[11] * c - l1[2][3]); s1[11] = p; dp1[11] = p * dp1[11]; }
        asm volatile("" : "+v"(dvT[0]), "+v"(dkT[0]), "+v"(dvT[1]), "+v"(dkT[1]), "+v"(s1), "+v"(dp1));
        dvT[1] = MT<T>::mfma32(fdo10[1], pf1[0], dvT[1]);
        { const float p = __builtin_amdgcn_exp2f(s1[12] * c - l1[3][0]); s1[12] = p; dp1[12] = p * dp1[12]; }
        { const float p = __builtin_amdgcn_exp2f(s1[13] * c - l1[3][1]); s1[13] = p; dp1[13] = p * dp1[13]; }
        asm volatile("" : "+v"(dvT[0]), "+v"(dkT[0]), "+v"(dvT[1]), "+v"(dkT[1]), "+v"(s1), "+v"(dp1));
        dkT[1] = MT<T>::mfma32(fq10[1], dsf1[0], dkT[1]);
        { const float p = __builtin_amdgcn_exp2f(s1[14] * c - l1[3][2]); s1[14] = p; dp1[14] = p * dp1[14]; }
        { const float p = __builtin_amdgcn_exp2f(s1[15] * c - l1[3][3]); s1[15] = p; dp1[15] = p * dp1[15]; }
        pf1[1] = acc_to_bfrag<T>(s1, 1); dsf1[1] = acc_to_bfrag<T>(dp1, 1);
        asm volatile("" : "+v"(dvT[0]), "+v"(dkT[0]), "+v"(dvT[1]), "+v"(dkT[1]), "+v"(s1), "+v"(dp1), "+v"(pf1[1]), "+v"(dsf1[1]), "+v"(fdo11[0]), "+v"(fq11[0]), "+v"(fdo11[1]), "+v"(fq11[1]));
        // C1b: the last four MFMAs run under the LDS stores of the next tile below
        dvT[0] = MT<T>::mfma32(fdo11[0], pf1[1], dvT[0]);
        dkT[0] = MT<T>::mfma32(fq11[0], dsf1[1], dkT[0]);
        dvT[1] = MT<T>::mfma32(fdo11[1], pf1[1], dvT[1]);
        dkT[1] = MT<T>::mfma32(fq11[1], dsf1[1], dkT[1]);
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    if (a.nsplit == 1) {
        T* dkp = (T*)a.dk + (int64_t)b * a.dk_bs + (int64_t)kvi * a.dk_rs + h * D;
        T* dvp = (T*)a.dv + (int64_t)b * a.dv_bs + (int64_t)kvi * a.dv_rs + h * D;
        store_out_rows<T, D>(dkp, kvalid, dkT, a.scale, hh);
        store_out_rows<T, D>(dvp, kvalid, dvT, 1.0f, hh);
    } else if (kvalid) {
        const int64_t bh = (int64_t)b * a.H + h;
        const int64_t slab = (int64_t)a.B * a.H * a.Nkv * D;
        float* pk = a.part + ((int64_t)split * slab) + (bh * a.Nkv + kvi) * D;
        float* pv = pk + (int64_t)a.nsplit * slab;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int db = 32 * dt + 8 * r4 + 4 * hh;
                if (db < D) {
                    *reinterpret_cast<f32x4*>(pk + db) = f32x4{dkT[dt][4 * r4] * a.scale, dkT[dt][4 * r4 + 1] * a.scale,
                                                              dkT[dt][4 * r4 + 2] * a.scale, dkT[dt][4 * r4 + 3] * a.scale};
                    *reinterpret_cast<f32x4*>(pv + db) = f32x4{dvT[dt][4 * r4], dvT[dt][4 * r4 + 1], dvT[dt][4 * r4 + 2],
                                                              dvT[dt][4 * r4 + 3]};
                }
            }
    }
}

// ---- host dispatch -----------------------------------------------------------------------------
template <int D> constexpr int fwd_qw() { return D <= 80 ? 64 : 32; }

template <int D> constexpr size_t fwd_lds(size_t es) { return 2 * (HD<D>::ROW_TILE_ELEMS + HD<D>::TR_TILE_ELEMS) * es; }
template <int D> constexpr size_t dq_lds(size_t es) { return 2 * (2 * HD<D>::ROW_TILE_ELEMS + HD<D>::TR_TILE_ELEMS) * es; }
template <int D> constexpr size_t dkdv_lds(size_t es) {
    return DKDV_NBUF<D>::value * ((2 * HD<D>::ROW_TILE_ELEMS + 2 * HD<D>::TR_TILE_ELEMS) * es +
                                  KV_TILE * (2 + MOS_MAX_PCOLS) * sizeof(float));
}

template <typename K>
void set_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

int check_shape(const mos_attn_shape* s, const char* who) {
    if (!s) return mos_set_error(MOS_ERR_BAD_ARG, "%s: NULL shape", who);
    if (s->B <= 0 || s->H <= 0 || s->Nq <= 0 || s->Nkv <= 0)
        return mos_set_error(MOS_ERR_BAD_ARG, "%s: B=%d H=%d Nq=%d Nkv=%d", who, s->B, s->H, s->Nq, s->Nkv);
    if (s->d != 40 && s->d != 64 && s->d != 80 && s->d != 160)
        return mos_set_error(MOS_ERR_UNSUPPORTED, "%s: head dim %d not in {40, 64, 80, 160}", who, s->d);
    if (s->causal && s->Nq != s->Nkv)
        return mos_set_error(MOS_ERR_BAD_ARG, "%s: causal masking needs Nq == Nkv (got %d, %d)", who, s->Nq, s->Nkv);
    if (s->q_rs % 8 || s->k_rs % 8 || s->v_rs % 8 || s->o_rs % 4 || s->q_bs % 8 || s->k_bs % 8 || s->v_bs % 8 || s->o_bs % 4)
        return mos_set_error(MOS_ERR_BAD_ARG, "%s: strides must be multiples of 8 elements", who);
    return MOS_OK;
}

AttnArgs make_args(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* tok, int np,
                   float* pcols, const mos_attn_shape* s, int q_tile) {
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse; a.tok_idx = tok; a.pcols = pcols; a.n_pcols = np;
    a.B = s->B; a.H = s->H; a.Nq = s->Nq; a.Nkv = s->Nkv; a.nqb = (s->Nq + q_tile - 1) / q_tile;
    a.q_bs = s->q_bs; a.q_rs = s->q_rs; a.k_bs = s->k_bs; a.k_rs = s->k_rs;
    a.v_bs = s->v_bs; a.v_rs = s->v_rs; a.o_bs = s->o_bs; a.o_rs = s->o_rs; a.scale = s->scale;
    a.causal = s->causal;
    return a;
}

template <typename T>
const char* tname() { return sizeof(T) == 2 && std::is_same<T, f16_t>::value ? "f16" : "bf16"; }

struct AttnKey {
    char s[96];
    double flops, bytes;
    AttnKey(const char* dt, const mos_attn_shape* sh, double gemms) {
        snprintf(s, sizeof(s), "%s d%d B%d H%d Nq%d Nkv%d", dt, sh->d, sh->B, sh->H, sh->Nq, sh->Nkv);
        // one "gemm" = 2*Nq*Nkv*d flops per head; forward has 2 (QK^T, PV), dq pass 3, dk/dv pass 4
        flops = gemms * 2.0 * sh->B * sh->H * (double)sh->Nq * sh->Nkv * sh->d;
        bytes = 2.0 * sh->B * sh->H * sh->d * (2.0 * sh->Nq + 2.0 * sh->Nkv);  // q,o + k,v at 2 B/element
    }
};

// causal masking (the CLIP text tower's d = 64 is the only user on the path; the other head dims keep it for the tests)
template <int D> constexpr bool has_causal() { return true; }

template <typename T, int D, int QW, bool PCOLS, bool CAUSAL>
void launch_fwd_one(const AttnArgs& a, size_t lds, hipStream_t st) {
    const dim3 grid((unsigned)(a.H * a.nqb * a.B));
    set_lds(&attn_fwd_kernel<T, D, QW, PCOLS, CAUSAL>, lds);
    hipLaunchKernelGGL((attn_fwd_kernel<T, D, QW, PCOLS, CAUSAL>), grid, dim3(256), lds, st, a);
}
template <typename T, int D, int QW>
void launch_fwd_qw(const AttnArgs& a, int np, size_t lds, hipStream_t st) {
    if constexpr (has_causal<D>()) {
        if (a.causal) {
            if (np > 0) launch_fwd_one<T, D, QW, true, true>(a, lds, st);
            else launch_fwd_one<T, D, QW, false, true>(a, lds, st);
            return;
        }
    }
    if (np > 0) launch_fwd_one<T, D, QW, true, false>(a, lds, st);
    else launch_fwd_one<T, D, QW, false, false>(a, lds, st);
}

template <typename T, int D>
int launch_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* tok, int np,
               float* pcols, const mos_attn_shape* s, hipStream_t st) {
    constexpr int QW = fwd_qw<D>();
    const size_t lds = fwd_lds<D>(sizeof(T));
    AttnKey key(tname<T>(), s, 2.0);
    MosProfScope prof(st, np > 0 ? "attn_fwd_pcols" : "attn_fwd", key.s, key.flops, key.bytes);
    // 64 queries per wave halve the LDS operand traffic, but with few workgroups (small batch at inference, short
    // sequences) they leave CUs idle / unbalanced: fall back to 32 queries per wave when the 256-query grid would
    // not give every CU at least two workgroups.
    const int64_t wg_big = (int64_t)s->H * s->B * ((s->Nq + 4 * QW - 1) / (4 * QW));
    // (a software-pipelined d = 40 forward -- S(j+1) under the exponentials of S(j), lazily moved softmax reference -- was built in
    //  round 4, measured 185-191 us against 161-164 us for this kernel at B4 H8 N4096, and removed in round 5;
    //  profiles/r04_kernel_bench_attn_pipe_*.txt)
    if (QW == 64 && wg_big < 512) {
        launch_fwd_qw<T, D, 32>(make_args(q, k, v, o, lse, tok, np, pcols, s, 128), np, lds, st);
    } else {
        launch_fwd_qw<T, D, QW>(make_args(q, k, v, o, lse, tok, np, pcols, s, 4 * QW), np, lds, st);
    }
    return mos_check_launch("attn_fwd");
}

template <typename T, int D>
int launch_region(const void* q, const void* k, const void* v, void* o, const mos_attn_shape* s,
                  const mos_region_desc* reg, const unsigned char* total_count, int accumulate, hipStream_t st) {
    AttnArgs a = make_args(q, k, v, o, nullptr, nullptr, 0, nullptr, s, 128);
    a.nqb = ((reg->feat_w + 15) / 16) * ((reg->feat_h + 7) / 8);        // 8 x 16 query tiles of the feature map
    const dim3 grid((unsigned)(a.H * a.nqb * a.B));
    const size_t lds = RG<D>::lds_bytes(sizeof(T));
    set_lds(&region_attn_kernel<T, D>, lds);
    // algorithmic work: every query attends to the context OR to its covering regions (box areas)
    double cover = 0.0, area = 0.0;
    for (int r = 0; r < reg->n_regions; ++r) {
        const double bh = reg->box[r][2] - reg->box[r][0], bw = reg->box[r][3] - reg->box[r][1];
        if (bh > 0 && bw > 0) area += bh * bw;
    }
    cover = area + (double)s->Nq;  // upper bound of (query, source) pairs: uncovered queries use the context
    AttnKey key(tname<T>(), s, 2.0);
    char rk[128];
    snprintf(rk, sizeof(rk), "%s R%d", key.s, reg->n_regions);
    MosProfScope prof(st, "region_attn", rk, key.flops * cover / (double)s->Nq, key.bytes);
    hipLaunchKernelGGL((region_attn_kernel<T, D>), grid, dim3(256), lds, st, a, *reg, total_count, accumulate);
    return mos_check_launch("region_attn");
}

struct BwdPlan { int nw_q, nw_k, nqb, nkb, nsplit, q_per_split; };
// 4 waves per block in both backward kernels (8-wave dQ blocks measured +47 % slower at d = 40)
BwdPlan plan_bwd(const mos_attn_shape* s) {
    BwdPlan p;
    p.nw_q = 4;
    p.nw_k = 4;
    p.nqb = (s->Nq + 32 * p.nw_q - 1) / (32 * p.nw_q);
    p.nkb = (s->Nkv + 32 * p.nw_k - 1) / (32 * p.nw_k);
    const int64_t base = (int64_t)p.nkb * s->B * s->H;
    const int qtiles = (s->Nq + KV_TILE - 1) / KV_TILE;
    int ns = (int)((512 + base - 1) / base);
    if (ns > qtiles) ns = qtiles;
    if (ns < 1) ns = 1;
    int tps = (qtiles + ns - 1) / ns;  // tiles per split
    ns = (qtiles + tps - 1) / tps;
    p.nsplit = ns;
    p.q_per_split = tps * KV_TILE;
    return p;
}

template <typename T, int D>
int launch_bwd(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dO,
               const int32_t* tok, int np, const float* pcols, const float* dpcols, void* dq, void* dk, void* dv,
               void* ws, const mos_attn_shape* s, const mos_attn_grad_strides* g, hipStream_t st) {
    const BwdPlan p = plan_bwd(s);
    float* Dvec = (float*)ws;
    const int64_t nrows = (int64_t)s->B * s->H * s->Nq;
    float* part = Dvec + ((nrows + 3) / 4) * 4;
    AttnBwdArgs a;
    a.q = q; a.k = k; a.v = v; a.dO = dO; a.lse = lse; a.Dvec = Dvec; a.tok_idx = tok; a.dpcols = dpcols;
    a.o = o; a.o_bs = s->o_bs; a.o_rs = s->o_rs; a.pcols = pcols;
    a.n_pcols = (dpcols != nullptr && pcols != nullptr) ? np : 0;
    a.dq = dq; a.dk = dk; a.dv = dv; a.part = part;
    a.B = s->B; a.H = s->H; a.Nq = s->Nq; a.Nkv = s->Nkv; a.nqb = p.nqb; a.nkb = p.nkb;
    a.nsplit = p.nsplit; a.q_per_split = p.q_per_split;
    a.q_bs = s->q_bs; a.q_rs = s->q_rs; a.k_bs = s->k_bs; a.k_rs = s->k_rs; a.v_bs = s->v_bs; a.v_rs = s->v_rs;
    a.do_bs = g->do_bs; a.do_rs = g->do_rs; a.dq_bs = g->dq_bs; a.dq_rs = g->dq_rs;
    a.dk_bs = g->dk_bs; a.dk_rs = g->dk_rs; a.dv_bs = g->dv_bs; a.dv_rs = g->dv_rs; a.scale = s->scale;
    a.causal = s->causal;
    const bool pc = a.n_pcols > 0;
    {
        const dim3 grid((unsigned)(a.H * a.nqb * a.B));
        const size_t lds = dq_lds<D>(sizeof(T));
        AttnKey key(tname<T>(), s, 3.0);
        MosProfScope prof(st, "attn_bwd_dq", key.s, key.flops, key.bytes * 1.5);
        auto go = [&](auto pc_c, auto ca_c) {
            constexpr bool PC = decltype(pc_c)::value, CA = decltype(ca_c)::value;
            set_lds(&attn_bwd_dq_kernel<T, D, PC, 4, CA>, lds);
            hipLaunchKernelGGL((attn_bwd_dq_kernel<T, D, PC, 4, CA>), grid, dim3(256), lds, st, a);
        };
        bool done = false;
        if constexpr (has_causal<D>()) {
            if (a.causal) {
                if (pc) go(std::true_type{}, std::true_type{}); else go(std::false_type{}, std::true_type{});
                done = true;
            }
        }
        if (!done) { if (pc) go(std::true_type{}, std::false_type{}); else go(std::false_type{}, std::false_type{}); }
        int rc = mos_check_launch("attn_bwd_dq");
        if (rc) return rc;
    }
    {
        const dim3 grid((unsigned)(a.H * a.nkb * a.B), (unsigned)a.nsplit);
        const size_t lds = dkdv_lds<D>(sizeof(T));
        AttnKey key(tname<T>(), s, 4.0);
        MosProfScope prof(st, "attn_bwd_dkdv", key.s, key.flops, key.bytes * 1.5);
        auto go = [&](auto pc_c, auto ca_c) {
            constexpr bool PC = decltype(pc_c)::value, CA = decltype(ca_c)::value;
            set_lds(&attn_bwd_dkdv_kernel<T, D, PC, 4, CA>, lds);
            hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, D, PC, 4, CA>), grid, dim3(256), lds, st, a);
        };
        bool done = false;
        if constexpr (D == 40) {
            // the slot-interleaved form: 297.6-304.4 us against 332-334 us for attn_bwd_dkdv_kernel at B4 H8 N4096 on the same box
            // (profiles/r04_kernel_bench_attn_pipe_*.txt); whole query tiles, no probability columns, no causal mask
            if (!pc && !a.causal && s->Nq % KV_TILE == 0 && a.q_per_split % KV_TILE == 0) {
                set_lds(&attn_bwd_dkdv_pipe_kernel<T, D>, lds);
                hipLaunchKernelGGL((attn_bwd_dkdv_pipe_kernel<T, D>), grid, dim3(256), lds, st, a);
                done = true;
            }
        }
        if constexpr (has_causal<D>()) {
            if (!done && a.causal) {
                if (pc) go(std::true_type{}, std::true_type{}); else go(std::false_type{}, std::true_type{});
                done = true;
            }
        }
        if (!done) { if (pc) go(std::true_type{}, std::false_type{}); else go(std::false_type{}, std::false_type{}); }
        int rc = mos_check_launch("attn_bwd_dkdv");
        if (rc) return rc;
    }
    if (a.nsplit > 1) {
        const int64_t slab4 = (int64_t)s->B * s->H * s->Nkv * D / 4;
        hipLaunchKernelGGL((attn_bwd_reduce_kernel<T, D>), dim3((unsigned)((slab4 + 255) / 256)), dim3(256), 0, st, part,
                           a.nsplit, (T*)dk, g->dk_bs, g->dk_rs, (T*)dv, g->dv_bs, g->dv_rs, s->B, s->H, s->Nkv);
        return mos_check_launch("attn_bwd_reduce");
    }
    return MOS_OK;
}

#define MOS_DISPATCH_TD(dtype, d, CALL)                                                     \
    do {                                                                                    \
        if ((dtype) == MOS_F16) {                                                           \
            typedef f16_t TT;                                                               \
            if ((d) == 40) { constexpr int DD = 40; return CALL; }                          \
            if ((d) == 64) { constexpr int DD = 64; return CALL; }                          \
            if ((d) == 80) { constexpr int DD = 80; return CALL; }                          \
            if ((d) == 160) { constexpr int DD = 160; return CALL; }                        \
        } else if ((dtype) == MOS_BF16) {                                                   \
            typedef bf16_t TT;                                                              \
            if ((d) == 40) { constexpr int DD = 40; return CALL; }                          \
            if ((d) == 64) { constexpr int DD = 64; return CALL; }                          \
            if ((d) == 80) { constexpr int DD = 80; return CALL; }                          \
            if ((d) == 160) { constexpr int DD = 160; return CALL; }                        \
        }                                                                                   \
        return mos_set_error(MOS_ERR_UNSUPPORTED, "unsupported dtype %d / head dim %d", (int)(dtype), (int)(d)); \
    } while (0)

}  // namespace

extern "C" {

int mos_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* tok_idx, int n_pcols,
                 float* pcols, const mos_attn_shape* s, int dtype, void* stream) {
    int rc = check_shape(s, "mos_attn_fwd");
    if (rc) return rc;
    MOS_REQUIRE(q && k && v && o, "mos_attn_fwd: NULL tensor");
    MOS_REQUIRE(n_pcols >= 0 && n_pcols <= MOS_MAX_PCOLS, "mos_attn_fwd: n_pcols=%d (max %d)", n_pcols, MOS_MAX_PCOLS);
    MOS_REQUIRE(n_pcols == 0 || (tok_idx && pcols), "mos_attn_fwd: n_pcols>0 needs tok_idx and pcols");
    hipStream_t st = (hipStream_t)stream;
    MOS_DISPATCH_TD(dtype, s->d, (launch_fwd<TT, DD>(q, k, v, o, lse, tok_idx, n_pcols, pcols, s, st)));
}

int mos_self_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const mos_attn_shape* s,
                      int dtype, void* stream) {
    return mos_attn_fwd(q, k, v, o, lse, nullptr, 0, nullptr, s, dtype, stream);
}

int mos_cross_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* tok_idx,
                       int n_pcols, float* pcols, const mos_attn_shape* s, int dtype, void* stream) {
    return mos_attn_fwd(q, k, v, o, lse, tok_idx, n_pcols, pcols, s, dtype, stream);
}

int64_t mos_attn_bwd_workspace_bytes(const mos_attn_shape* s) {
    if (!s) return 0;
    const BwdPlan p = plan_bwd(s);
    const int64_t nrows = (int64_t)s->B * s->H * s->Nq;
    int64_t fl = ((nrows + 3) / 4) * 4;
    if (p.nsplit > 1) fl += 2 * (int64_t)p.nsplit * s->B * s->H * s->Nkv * s->d;
    return fl * (int64_t)sizeof(float);
}

int mos_attn_bwd(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dO,
                 const int32_t* tok_idx, int n_pcols, const float* pcols, const float* dpcols, void* dq, void* dk,
                 void* dv, void* ws, const mos_attn_shape* s, const mos_attn_grad_strides* g, int dtype, void* stream) {
    int rc = check_shape(s, "mos_attn_bwd");
    if (rc) return rc;
    MOS_REQUIRE(q && k && v && o && lse && dO && dq && dk && dv && ws && g, "mos_attn_bwd: NULL argument");
    MOS_REQUIRE(n_pcols >= 0 && n_pcols <= MOS_MAX_PCOLS, "mos_attn_bwd: n_pcols=%d", n_pcols);
    MOS_REQUIRE(dpcols == nullptr || (n_pcols > 0 && tok_idx && pcols), "mos_attn_bwd: dpcols needs tok_idx, pcols");
    MOS_REQUIRE(g->do_rs % 8 == 0 && g->dq_rs % 4 == 0 && g->dk_rs % 4 == 0 && g->dv_rs % 4 == 0,
                "mos_attn_bwd: gradient strides misaligned");
    hipStream_t st = (hipStream_t)stream;
    MOS_DISPATCH_TD(dtype, s->d,
                    (launch_bwd<TT, DD>(q, k, v, o, lse, dO, tok_idx, n_pcols, pcols, dpcols, dq, dk, dv, ws, s, g, st)));
}

int mos_self_attn_bwd(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dO, void* dq,
                      void* dk, void* dv, void* ws, const mos_attn_shape* s, const mos_attn_grad_strides* g, int dtype,
                      void* stream) {
    return mos_attn_bwd(q, k, v, o, lse, dO, nullptr, 0, nullptr, nullptr, dq, dk, dv, ws, s, g, dtype, stream);
}

int mos_cross_attn_bwd(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dO,
                       const int32_t* tok_idx, int n_pcols, const float* pcols, const float* dpcols, void* dq, void* dk,
                       void* dv, void* ws, const mos_attn_shape* s, const mos_attn_grad_strides* g, int dtype, void* stream) {
    return mos_attn_bwd(q, k, v, o, lse, dO, tok_idx, n_pcols, pcols, dpcols, dq, dk, dv, ws, s, g, dtype, stream);
}

int mos_region_cross_attn_fwd(const void* q, const void* k_src, const void* v_src, void* o, const mos_attn_shape* s,
                              const mos_region_desc* reg, int dtype, void* stream) {
    return mos_region_cross_attn_fwd_chunk(q, k_src, v_src, o, s, reg, nullptr, 0, dtype, stream);
}

int mos_region_cross_attn_fwd_chunk(const void* q, const void* k_src, const void* v_src, void* o, const mos_attn_shape* s,
                                    const mos_region_desc* reg, const void* total_count, int accumulate, int dtype,
                                    void* stream) {
    int rc = check_shape(s, "mos_region_cross_attn_fwd");
    MOS_REQUIRE(!accumulate || total_count, "mos_region_cross_attn_fwd_chunk: accumulate needs the total count map");
    if (rc) return rc;
    MOS_REQUIRE(q && k_src && v_src && o && reg, "mos_region_cross_attn_fwd: NULL argument");
    MOS_REQUIRE(reg->n_regions >= 0 && reg->n_regions <= MOS_MAX_SOURCES - 1, "mos_region_cross_attn_fwd: n_regions=%d",
                reg->n_regions);
    MOS_REQUIRE(reg->feat_h > 0 && reg->feat_w > 0 && reg->feat_h * reg->feat_w == s->Nq,
                "mos_region_cross_attn_fwd: feat %dx%d != Nq %d", reg->feat_h, reg->feat_w, s->Nq);
    if (s->Nkv <= 64 || s->Nkv > 96)
        return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_region_cross_attn_fwd: built for CLIP's 77-token context "
                             "(65..96 keys per source), got Nkv=%d", s->Nkv);
    hipStream_t st = (hipStream_t)stream;
    MOS_DISPATCH_TD(dtype, s->d, (launch_region<TT, DD>(q, k_src, v_src, o, s, reg, (const unsigned char*)total_count,
                                                        accumulate ? 1 : 0, st)));
}

}  // extern "C"

// mos_conv.hip — 3x3 stride-1 pad-1 convolution on channels-last half activations as an implicit GEMM (gfx950).
//
// A CALLER of the attention path: the ResnetBlock2D / Upsample2D / VAE convolutions around every
// transformer block. In ED-LoRA training their weights are frozen, so only forward and backward-DATA are needed, and the
// latter is the same operator with flipped, transposed taps:
//     y[b,y,x,co]  = sum_{ky,kx,ci} x[b, y+ky-1, x+kx-1, ci] * W[co][ky][kx][ci]   (+ bias[co] + tbias[b][co] + res[b,y,x,co])
//     dx[b,y,x,ci] = sum_{ky,kx,co} dy[b, y+ky-1, x+kx-1, co] * Wb[ci][ky][kx][co],  Wb[ci][ky][kx][co] = W[co][2-ky][2-kx][ci]
// GEMM view: M = B*H*W pixels, N = Cout, K = 9*Cin with k = (tap, ci). In NHWC a pixel's channels are one contiguous run and
// pixel (b,y,x) has linear index m, so the "im2col row" of tap (dy,dx) is just the run of pixel m + dy*W + dx — or zeros
// outside the image, which the buffer-load bounds check delivers for free when the offset is pushed past the descriptor's
// range. No im2col buffer, no layout transposes, no zero-fill of split-K outputs; the per-sample time-embedding bias and
// the ResNet residual add ride in the epilogue (MIOpen: separate transpose / SubTensorOp / add kernels around each call).
//
// Staging: the tiles go global -> LDS directly (`buffer_load_dwordx4 ... lds`, LDS-DMA): no staging VGPRs and, above all,
// no ds_write_b128 pass — the VGPR->LDS store path moves ~79 B/clk/CU against 256 B/clk for ds_read_b128
// (MI355X_MICROARCH.md, LDS), so at 128x128x64 register-staged stores cost ~415 clk per 515 clk of MFMA. Measured
// against the register-staged form of this kernel (same box): 64x64 maps 83 -> 49 us, VAE 512-px stage 536 -> 463 us,
// 128-px stage 447 -> 377 us. The DMA destination is lane-linear (M0 base + lane * 16 B), so rows cannot be padded: the
// tile image is [row][8 chunks of 16 B] with the chunk index XOR-ed with (row & 7), applied on the SOURCE side (which
// chunk a lane fetches) and undone in the fragment reads — conflict-free for the ds_read_b128 lane groups (checked
// exhaustively). Out-of-image taps and rows past M still come from the bounds check: the DMA deposits the zeros.
//
// Two pipelines: NS = 2 — double buffer, one tile ahead, `vmcnt(0)` + barrier per K tile — for grids with several blocks
// per CU (block-level overlap hides the DMA latency); NS >= 3 — a ring with NS - 1 tiles in flight and a COUNTED
// `s_waitcnt vmcnt((NS-2) * loads-per-tile)` in front of a raw `s_barrier` (a `__syncthreads()` would drain the queue) —
// for the low-resolution levels, where ~1 block per CU walks a 9*Cin-deep K loop and every K tile otherwise pays a full
// L2/HBM round trip. Tiles past the end of K are issued as out-of-range (zero) DMAs so the count is the same every
// iteration; the ring is drained before the epilogue reuses the LDS.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "mos_common.h"

namespace {

constexpr int CBK = 64;

struct ConvArgs {
    const void* X; const void* W; const float* bias; const void* tbias; const void* R; void* Y;
    int B, H, Wd, Cin, Cout;      // Wd = image width
    int M, mt, nt, cpt;           // M = B*H*W ; cpt = Cin / 64 channel chunks per tap
    int up;                       // 1: the input is read through a nearest 2x upsample (x is (B, H/2, W/2, Cin))
    int ldx;                      // PIXEL stride of x in elements: Cin for a dense map, wider when x is a channel slice of a wider
                                  // channels-last tensor (a concatenation's gradient read in place by the dX convolution: round 6)
    int Hin, Win;                 // stride-2 form only: the source image size (H, Wd are the OUTPUT size)
    float* gn_part;               // halo form only, or NULL: per-(pixel tile, output channel) sum and sum of squares of the stored values
    int ksplit, kt_per;           // split-K form: K tiles [z * kt_per, (z + 1) * kt_per) per workgroup, z < ksplit
    float* partial;               // split-K form: fp32 partial sums [ksplit][M][Cout]
};

template <int N>
__device__ __forceinline__ void wait_vmcnt_then_barrier() {
    // counted wait on this wave's DMAs (and on its LDS reads: the scheduler may sink MFMAs, with the wait for their
    // operands, below this point), then the workgroup barrier; one asm block with a memory clobber so that no LDS access
    // moves across it (s_barrier itself does not drain VMEM: later tiles stay in flight)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// SPLIT (low-resolution levels: 32x32 .. 8x8 maps, M = 192 .. 4096 pixels against K = 9 * Cin = 5.8 k .. 23 k): few output
// tiles, each walking a 90 - 360 tile K loop, while the operator streams up to 59 MB of weights (see conv_ksplit). The K loop is cut into `ksplit` ranges (tap-major, so a range is a few taps of a channel
// slab); each workgroup leaves its fp32 partial tile in `partial[z]`, conv_splitk_reduce_kernel sums them in z order
// (deterministic) and applies the epilogue. All m-tiles of one (n-tile, K range) run on ONE XCD, back to back: its weight
// slice is fetched into that L2 once.
// S2 (round 6): the 3x3 / STRIDE 2 convolutions of the down-samplers on the same kernel -- 1: padding 1 (diffusers Downsample2D of
// the UNet), 2: padding 0 on a map padded by one row / column of zeros at the bottom / right (the VAE encoder's asymmetric pad,
// F.pad(x, (0, 1, 0, 1)): folded into the bounds, the padded copy is never made). Output pixel (y, x) reads source pixel
// (2y + ky - P, 2x + kx - P), P = 1 / 0: only the per-pixel base offset, the nine validity bits and the tap delta change.
template <typename T, int BM, int BN, bool UP, int NS, bool SPLIT = false, int S2 = 0>
__global__ __launch_bounds__(256) void conv3x3_nhwc_kernel(const ConvArgs a) {
    static_assert(!(UP && S2), "upsampled input and stride 2 exclude each other");
    constexpr int ST = S2 ? 2 : 1, PD = S2 == 2 ? 0 : 1;
    typedef typename MT<T>::v8 v8;
    constexpr int MI = BM / 32, NJ = BN / 32;
    constexpr int XCH = BM * 8 / 256, WCH = BN * 8 / 256;
    constexpr int CS = BN + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Xs = reinterpret_cast<T*>(smem_raw);        // [NS][BM][64], chunk-swizzled
    T* Ws = Xs + NS * BM * CBK;                    // [NS][BN][64]

    const int w = blockIdx.x;                      // XCD-aware map: the 8 m-tiles of a slot share one weight column block
    int n_tile, m_tile, kt0 = 0, kt1 = 9 * a.cpt, zsplit = 0;
    if constexpr (SPLIT) {
        const int r = w >> 3;
        m_tile = r % a.mt;
        const int u = (r / a.mt) * 8 + (w & 7);    // unit = (n-tile, K range); its mt workgroups share XCD w & 7
        if (u >= a.nt * a.ksplit) return;
        n_tile = u % a.nt;
        zsplit = u / a.nt;
        kt0 = zsplit * a.kt_per;
        kt1 = min(kt1, kt0 + a.kt_per);
    } else {
        const int slot = w >> 3;
        n_tile = slot % a.nt;
        m_tile = (slot / a.nt) * 8 + (w & 7);
        if (m_tile >= a.mt) return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably uniform: the LDS-DMA destinations (M0) become SALU values
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lg = lane >> 4;
    const int n0 = n_tile * BN, m0 = m_tile * BM;
    const int M = a.M, N = a.Cout, C = a.Cin, LX = a.ldx, H = a.H, Wd = a.Wd;
    const int K = 9 * C;
    const int Hs = S2 ? a.Hin : (UP ? H / 2 : H), Ws_ = S2 ? a.Win : (UP ? Wd / 2 : Wd);     // source image size

    const rsrc_t xsrc = make_rsrc(a.X, (uint32_t)((((int64_t)a.B * Hs * Ws_ - 1) * LX + C) * (int64_t)sizeof(T)));
    const rsrc_t wsrc = make_rsrc(a.W, (uint32_t)((((int64_t)N - 1) * K + K) * (int64_t)sizeof(T)));
    constexpr int OOB = 0x7FFFFF00;

    // this lane's 16-byte slot in a wave-instruction: row = slot / 8, physical chunk = slot % 8 -> logical chunk
    const int cc8 = ((tid & 7) ^ ((tid >> 3) & 7)) * 8;
    int py[XCH], px[XCH], rowoff[XCH], woff[WCH];
    // tapmask[i] bit t: tap t of this thread's pixel row i lies inside the image (and the row inside M). Round 4: the nine
    // validity tests per pixel are made ONCE here instead of per K tile -- the main loop of the 64 x 64 configuration carried 34
    // VALU (4 compares + selects + an exec-masked branch per staged chunk, a readfirstlane per DMA destination) for 8 MFMAs, and
    // on this chip VALU and MFMA time add up.
    [[maybe_unused]] uint32_t tapmask[XCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int m = m0 + ((tid + 256 * i) >> 3);
        tapmask[i] = 0u;
        if (m < M) {
            const int b = m / (H * Wd), p = m - b * (H * Wd);
            py[i] = p / Wd; px[i] = p - py[i] * Wd;
            rowoff[i] = UP ? b * Hs : (((b * Hs + ST * py[i]) * Ws_ + ST * px[i]) * LX + cc8) * (int)sizeof(T);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = ST * py[i] + t / 3 - PD, xx = ST * px[i] + t % 3 - PD;
                if (yy >= 0 && yy < (S2 ? Hs : H) && xx >= 0 && xx < (S2 ? Ws_ : Wd)) tapmask[i] |= 1u << t;
            }
        } else {
            py[i] = -100000; px[i] = 0; rowoff[i] = 0;
        }
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i)
        woff[i] = (int)((((int64_t)(n0 + ((tid + 256 * i) >> 3))) * K + cc8) * (int64_t)sizeof(T));

    f32x4 acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = kt1;
    int tap = kt0 / a.cpt, cch = kt0 - tap * a.cpt;   // tap / channel chunk of the NEXT tile to issue

    auto issue_tile = [&](int kt, int buf) {   // tile kt -> LDS buffer buf; wave w's i-th piece = slots (4 i + w) * 64 ..
        const bool live = kt < nk;             // ring only: tiles past the end are issued out of range (zeros, no traffic)
        const int dy = tap / 3 - PD, dx = tap - (tap / 3) * 3 - PD;
        T* xs = Xs + buf * BM * CBK + wave * 512;
        T* ws = Ws + buf * BN * CBK + wave * 512;
        const int delta = ((dy * Ws_ + dx) * LX + cch * CBK) * (int)sizeof(T);      // non-upsampled source: linear in the tap
        if constexpr (UP) {
#pragma unroll
            for (int i = 0; i < XCH; ++i) {
                const int yy = py[i] + dy, xx = px[i] + dx;
                const bool ok = live && yy >= 0 && yy < H && xx >= 0 && xx < Wd;
                const int off = (((rowoff[i] + (yy >> 1)) * Ws_ + (xx >> 1)) * LX + cch * CBK + cc8) * (int)sizeof(T);
                dma16(xsrc, xs + i * 2048, ok ? off : OOB);
            }
        } else {
            const uint32_t tbit = live ? (1u << tap) : 0u;          // uniform: tap and `live` are scalars
#pragma unroll
            for (int i = 0; i < XCH; ++i) dma16(xsrc, xs + i * 2048, (tapmask[i] & tbit) ? rowoff[i] + delta : OOB);
        }
        const int kb = kt * CBK * (int)sizeof(T);
#pragma unroll
        for (int i = 0; i < WCH; ++i) dma16(wsrc, ws + i * 2048, live ? woff[i] + kb : OOB);
        if (++cch == a.cpt) { cch = 0; ++tap; }
    };

    // fragment reads: row l15 (+16 i), logical chunk kk * 4 + lg -> physical chunk ^ (row & 7)
    const int sw0 = ((lg ^ (l15 & 7)) * 8), sw1 = sw0 ^ 32;
    auto compute_tile = [&](int buf) {
        const T* xs = Xs + buf * BM * CBK + (wm * (BM / 2) + l15) * CBK;
        const T* ws = Ws + buf * BN * CBK + (wn * (BN / 2) + l15) * CBK;
#pragma unroll
        for (int kk = 0; kk < CBK / 32; ++kk) {
            const int sw = kk ? sw1 : sw0;
            v8 bfrag[MI], afrag[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) bfrag[i] = as_v8<T>(ld16(xs + i * 16 * CBK + sw));
#pragma unroll
            for (int j = 0; j < NJ; ++j) afrag[j] = as_v8<T>(ld16(ws + j * 16 * CBK + sw));
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[j][i] = MT<T>::mfma16(afrag[j], bfrag[i], acc[j][i]);
        }
    };

    if constexpr (NS == 2) {
        issue_tile(kt0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = kt0; kt < nk; ++kt) {
            const int cur = (kt - kt0) & 1;
            if (kt + 1 < nk) issue_tile(kt + 1, cur ^ 1);
            compute_tile(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        constexpr int L = XCH + WCH;           // DMAs per wave per tile
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) issue_tile(kt0 + s, s);
        int cur = 0, nxt = NS - 1;
        for (int kt = kt0; kt < nk; ++kt) {
            // tile kt has landed for this wave once at most the NS-2 younger tiles are outstanding; past the barrier it has
            // landed for all waves, and all of them have finished reading tile kt-1, whose buffer the next issue overwrites
            wait_vmcnt_then_barrier<(NS - 2) * L>();
            issue_tile(kt + NS - 1, nxt);
            compute_tile(cur);
            cur = (cur + 1 == NS) ? 0 : cur + 1;
            nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the zero tiles issued past the end
        __syncthreads();
    }

    if constexpr (SPLIT) {                     // raw fp32 partial tile; the reduce kernel owns the epilogue
        float* P = a.partial + (int64_t)zsplit * M * N;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 16 + lg * 4;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = m0 + wm * (BM / 2) + i * 16 + l15;
                if (m < M && n < N) *reinterpret_cast<f32x4*>(P + (int64_t)m * N + n) = acc[j][i];   // N % 8 == 0
            }
        }
        return;
    }
    // epilogue: + bias[n] + tbias[b(m)][n], round, stage in LDS, then coalesced rows (+ residual)
    T* Cs = reinterpret_cast<T*>(smem_raw);
    const T* tb = reinterpret_cast<const T*>(a.tbias);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int nl = wn * (BN / 2) + j * 16 + lg * 4;
        const int nb = min(n0 + nl, N - 4);
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        if (a.bias != nullptr) { b0 = a.bias[nb]; b1 = a.bias[nb + 1]; b2 = a.bias[nb + 2]; b3 = a.bias[nb + 3]; }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ml = wm * (BM / 2) + i * 16 + l15;
            f32x4 v = acc[j][i];
            if (tb != nullptr) {
                const int bimg = min(m0 + ml, M - 1) / (H * Wd);
                const typename MT<T>::v4 t4 = __builtin_bit_cast(typename MT<T>::v4, ld8(tb + (int64_t)bimg * N + nb));
                v[0] += (float)t4[0]; v[1] += (float)t4[1]; v[2] += (float)t4[2]; v[3] += (float)t4[3];
            }
            st8(Cs + ml * CS + nl, pack4<T>(v[0] + b0, v[1] + b1, v[2] + b2, v[3] + b3));
        }
    }
    __syncthreads();
    T* Y = reinterpret_cast<T*>(a.Y);
    const T* R = reinterpret_cast<const T*>(a.R);
    constexpr int OCH = BM * (BN / 8) / 256;
#pragma unroll
    for (int i = 0; i < OCH; ++i) {
        const int c = tid + 256 * i;
        const int row = c / (BN / 8), col = (c % (BN / 8)) * 8;
        if (m0 + row < M && n0 + col < N) {
            const int64_t o = (int64_t)(m0 + row) * N + n0 + col;
            u32x4 val = ld16(Cs + row * CS + col);
            if (R != nullptr) {
                const v8 r = as_v8<T>(ld16(R + o)), s = as_v8<T>(val);
                v8 q;
#pragma unroll
                for (int e = 0; e < 8; ++e) q[e] = (T)((float)s[e] + (float)r[e]);
                val = from_v8<T>(q);
            }
            st16(Y + o, val);
        }
    }
}

// ---- halo-staged form: the input HALO of a TH x 16 pixel tile is staged ONCE per channel chunk and the nine taps walk it in
// LDS; only the weight tile changes per K step. The raster form above re-fetches the activations of every tap: per wave and K step
// 4 (BM 128) activation pieces + BN/32 weight pieces, here BN/32 weight pieces + NPW/9 halo pieces (TH 8: 6/9) -- and a piece costs
// 60..185 issue cycles next to MFMAs (MI355X_MICROARCH.md), more than the step's MFMAs at these tiles.
//   tile      : image b, rows y0 .. y0+TH, columns x0 .. x0+16; output pixel ml = ty * 16 + tx
//   halo image: rows hr = hy * 18 + hx, hy < TH + 2, hx < 18 <-> image pixel (y0 + hy - 1, x0 + hx - 1) (zeros outside the image),
//               [hr][CK / 8 chunks of 16 B], chunk index swizzled by the row on the source side like the raster form
//   K order   : (channel chunk, tap) -- the raster form walks (tap, chunk); same sum, other fp32 association
//   pipeline  : weight tiles in a ring of 3 (two in flight), halo double-buffered: H(c+1) is issued at tap 0 of chunk c, in front
//               of that step's weight tile, so the counted wait of tap 1 allows NPW more pieces in flight than the others
//   main loop : MFMAs, ds_read_b128 with immediate offsets, LDS-DMA issues with scalar offsets, waits -- no VALU instruction
// CK = channels per chunk: 64 (rows of 8 x 16 B, chunk ^ (row & 7)) or 32 (rows of 4 x 16 B, chunk ^ ((row >> 1) & 3); a 1-KiB DMA
// piece is then 16 rows) -- both conflict-free for the ds_read_b128 lane groups with 16 consecutive rows per k-group
template <int CK>
__device__ __forceinline__ int halo_swz(int row) { return CK == 64 ? (row & 7) : ((row >> 1) & 3); }

template <typename T, int TH, int BN, bool UP, int CK>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const ConvArgs a) {
    typedef typename MT<T>::v8 v8;
    constexpr int BM = TH * 16, MI = TH / 2, NJ = BN / 32;
    constexpr int RPP = 512 / CK, CPR = CK / 8;    // rows per 1-KiB DMA piece, 16-B chunks per row
    constexpr int WCH = BN * CK / 2048;            // weight pieces per wave and K step
    // NPW pieces per wave, ALL issued by every wave (pieces past the halo deposit zeros in the buffer's tail): the counted waits
    // below bound what may stay in flight, so every wave must put the same number of pieces behind a weight tile
    constexpr int HW18 = 18, HR = (TH + 2) * HW18, NPW = ((HR + RPP - 1) / RPP + 3) / 4, HB = NPW * 4 * 512;
    constexpr int CS = BN + 8;
    static_assert(TH % 2 == 0 && BN % 32 == 0 && (CK == 32 || CK == 64) && WCH >= 1, "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Hs = reinterpret_cast<T*>(smem_raw);        // [2][NPW * 4 pieces][512], chunk-swizzled halo images
    T* Ws = Hs + 2 * HB;                           // [3][BN][CK]

    const int w = blockIdx.x;
    const int slot = w >> 3;
    const int n_tile = slot % a.nt;
    const int m_tile = (slot / a.nt) * 8 + (w & 7);
    if (m_tile >= a.mt) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lg = lane >> 4;
    const int N = a.Cout, C = a.Cin, LX = a.ldx, H = a.H, Wd = a.Wd;
    const int K = 9 * C;
    const int tx_n = (Wd + 15) / 16, ty_n = (H + TH - 1) / TH;
    const int b = m_tile / (tx_n * ty_n), tr = m_tile - b * (tx_n * ty_n);
    const int y0 = (tr / tx_n) * TH, x0 = (tr - (tr / tx_n) * tx_n) * 16;
    const int n0 = n_tile * BN;

    // UP: the input is read through a nearest 2x upsample -- halo pixel (yy, xx) of the H x Wd image the taps walk is source
    // pixel (yy >> 1, xx >> 1) of the (H/2) x (Wd/2) tensor; the zero border is the border of the UPSAMPLED image
    const int Hsrc = UP ? H / 2 : H, Wsrc = UP ? Wd / 2 : Wd;
    const rsrc_t xsrc = make_rsrc(a.X, (uint32_t)((((int64_t)a.B * Hsrc * Wsrc - 1) * LX + C) * (int64_t)sizeof(T)));
    const rsrc_t wsrc = make_rsrc(a.W, (uint32_t)((((int64_t)N - 1) * K + K) * (int64_t)sizeof(T)));
    // Addressing without per-step VALU work (the unrolled tap loop used to spend ~1 v_add per MFMA on it):
    //   * DMA sources: the lane part (pixel / weight row, swizzled chunk) is a loop-invariant VGPR -- 0x80000000, past any
    //     descriptor, for halo pixels outside the image --, the channel-chunk / tap part rides in the instruction's SCALAR offset,
    //     and the pieces issued past the last chunk use a descriptor of zero records (everything reads as zeros, no traffic)
    //   * fragment reads: row = (wave / pixel-fragment / tap constant) + l15, and the swizzle only looks at the row's low three
    //     bits, so a read is one of 8 lane patterns (x CK / 32 k-halves) plus an IMMEDIATE offset
    constexpr int OOB = (int)0x80000000u;
    const rsrc_t xdead = make_rsrc(a.X, 0u), wdead = make_rsrc(a.W, 0u);

    int hoff[NPW], woff[WCH];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int hr = (wave + 4 * i) * RPP + lane / CPR;
        const int hy = hr / HW18, hx = hr - hy * HW18;
        const int yy = y0 + hy - 1, xx = x0 + hx - 1;
        const int lc = (lane % CPR) ^ halo_swz<CK>(hr);
        const int ys = UP ? (yy >> 1) : yy, xs = UP ? (xx >> 1) : xx;
        hoff[i] = (hr < HR && yy >= 0 && yy < H && xx >= 0 && xx < Wd) ? (((b * Hsrc + ys) * Wsrc + xs) * LX + lc * 8) * (int)sizeof(T) : OOB;
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int q = tid + 256 * i, row = q / CPR;
        woff[i] = (int)((((int64_t)(n0 + row)) * K + (((q % CPR) ^ halo_swz<CK>(row)) * 8)) * (int64_t)sizeof(T));
    }

    f32x4 acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int cpt = a.cpt;
    auto issue_halo = [&](int cch, int hb) {       // chunk cch -> halo buffer hb (past the last chunk: zeros, no traffic)
        const bool live = cch < cpt;
        const rsrc_t src = live ? xsrc : xdead;
        const int cb = live ? cch * CK * (int)sizeof(T) : 0;
#pragma unroll
        for (int i = 0; i < NPW; ++i) dma16s(src, Hs + hb * HB + (wave + 4 * i) * 512, hoff[i], cb);
    };
    auto issue_w = [&](int cch, int tap, int buf) {
        const bool live = cch < cpt;
        const rsrc_t src = live ? wsrc : wdead;
        const int kb = live ? (tap * C + cch * CK) * (int)sizeof(T) : 0;
        T* ws = Ws + buf * BN * CK + wave * 512;
#pragma unroll
        for (int i = 0; i < WCH; ++i) dma16s(src, ws + i * 2048, woff[i], kb);
    };
    // lane patterns of the halo fragment reads: the halo row of output pixel (fragment i, column l15) under tap (ty, tx) is
    // wm * MI * 18 + cst + l15 with cst = (i + ty) * 18 + tx (ty, tx in 0..2); wm * MI * 18 is a multiple of 8, so the row's
    // low three bits are those of l15 + (cst & 7)
    constexpr int KK = CK / 32;
    int hpat[8][KK];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            hpat[c][kk] = (wm * MI * HW18 + l15) * CK + (((kk * 4 + lg) ^ halo_swz<CK>(l15 + c)) * 8);
    static_assert((MI * HW18) % 8 == 0, "the wave's first halo row must keep the swizzle class");
    const T* wfrag = Ws + (wn * (BN / 2) + l15) * CK;      // (fragment row bases are multiples of 16: the swizzle is that of l15)
    const int wsw = halo_swz<CK>(l15);

    // one channel chunk: nine taps over halo buffer HBUF (a compile-time constant: the K loop below is unrolled by two)
    auto chunk = [&](const int cch, auto hbuf_c) {
        constexpr int HBUF = decltype(hbuf_c)::value;
        const T* hs = Hs + HBUF * HB;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // the weight tile of this step has landed once only the younger pieces are outstanding: one weight tile, plus -- at
            // tap 1 -- the halo of the next chunk that was issued in front of it
            if (tap == 1) wait_vmcnt_then_barrier<WCH + NPW>(); else wait_vmcnt_then_barrier<WCH>();
            if (tap == 0) issue_halo(cch + 1, 1 - HBUF);
            {
                const int t2 = tap + 2;
                issue_w(t2 >= 9 ? cch + 1 : cch, t2 >= 9 ? t2 - 9 : t2, t2 % 3);
            }
            const T* ws = wfrag + (tap % 3) * BN * CK;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                v8 bfrag[MI], afrag[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int cst = (i + tap / 3) * HW18 + tap % 3;
                    bfrag[i] = as_v8<T>(ld16(hs + hpat[cst & 7][kk] + cst * CK));
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) afrag[j] = as_v8<T>(ld16(ws + j * 16 * CK + ((kk * 4 + lg) ^ wsw) * 8));
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[j][i] = MT<T>::mfma16(afrag[j], bfrag[i], acc[j][i]);
            }
        }
    };

    issue_halo(0, 0);
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);
    for (int cch = 0; cch < cpt; cch += 2) {
        chunk(cch, std::integral_constant<int, 0>{});
        if (cch + 1 < cpt) chunk(cch + 1, std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the zero pieces issued past the end
    __syncthreads();

    // epilogue: + bias[n] + tbias[b][n], round, stage in LDS, then coalesced rows (+ residual)
    T* Cs = reinterpret_cast<T*>(smem_raw);
    const T* tb = reinterpret_cast<const T*>(a.tbias);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int nl = wn * (BN / 2) + j * 16 + lg * 4;
        const int nb = min(n0 + nl, N - 4);
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        if (a.bias != nullptr) { b0 = a.bias[nb]; b1 = a.bias[nb + 1]; b2 = a.bias[nb + 2]; b3 = a.bias[nb + 3]; }
        if (tb != nullptr) {
            const typename MT<T>::v4 t4 = __builtin_bit_cast(typename MT<T>::v4, ld8(tb + (int64_t)b * N + nb));
            b0 += (float)t4[0]; b1 += (float)t4[1]; b2 += (float)t4[2]; b3 += (float)t4[3];
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ml = (wm * MI + i) * 16 + l15;
            const f32x4 v = acc[j][i];
            st8(Cs + ml * CS + nl, pack4<T>(v[0] + b0, v[1] + b1, v[2] + b2, v[3] + b3));
        }
    }
    __syncthreads();
    T* Y = reinterpret_cast<T*>(a.Y);
    const T* R = reinterpret_cast<const T*>(a.R);
    constexpr int OCH = BM * (BN / 8) / 256;
    // Round 6: the GroupNorm that consumes this map takes its statistics from HERE instead of re-reading the
    // map: per output channel the sum and the sum of squares of the values AS STORED (rounded, residual added) over the tile's
    // valid pixels. A thread's channel chunk is the same in every iteration of the store loop (256 % (BN / 8) == 0), so the
    // sums ride in 16 registers; one LDS pass folds the 256 / (BN / 8) threads of a chunk. The consumer combines tiles in a
    // fixed order (deterministic) -- gn_chan_finalize_kernel, mos_norm.hip.
    const bool want_gn = a.gn_part != nullptr;        // uniform
    float g0[8], g1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { g0[e] = 0.f; g1[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < OCH; ++i) {
        const int c = tid + 256 * i;
        const int row = c / (BN / 8), col = (c % (BN / 8)) * 8;
        const int yy = y0 + (row >> 4), xx = x0 + (row & 15);
        if (yy < H && xx < Wd && n0 + col < N) {
            const int64_t o = ((int64_t)(b * H + yy) * Wd + xx) * N + n0 + col;
            u32x4 val = ld16(Cs + row * CS + col);
            if (R != nullptr) {
                const v8 r = as_v8<T>(ld16(R + o)), s = as_v8<T>(val);
                v8 q;
#pragma unroll
                for (int e = 0; e < 8; ++e) q[e] = (T)((float)s[e] + (float)r[e]);
                val = from_v8<T>(q);
            }
            st16(Y + o, val);
            if (want_gn) {
                const v8 s = as_v8<T>(val);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)s[e]; g0[e] += f; g1[e] = fmaf(f, f, g1[e]); }
            }
        }
    }
    if (want_gn) {
        constexpr int RG = 256 / (BN / 8);                 // threads that share a channel chunk
        static_assert(256 % (BN / 8) == 0 && (size_t)RG * BN * 2 * sizeof(float) <= (size_t)BM * CS * sizeof(T), "statistics fit the staging tile");
        __syncthreads();                                   // every thread is done reading the staged tile
        float* red = reinterpret_cast<float*>(smem_raw);   // [RG][BN][2]
        const int rg = tid / (BN / 8), col = (tid % (BN / 8)) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[((rg * BN) + col + e) * 2] = g0[e];
            red[((rg * BN) + col + e) * 2 + 1] = g1[e];
        }
        __syncthreads();
        for (int idx = tid; idx < BN * 2; idx += 256) {    // idx = channel * 2 + moment
            float t = 0.f;
#pragma unroll 8
            for (int r = 0; r < RG; ++r) t += red[r * BN * 2 + idx];
            const int ch = n0 + (idx >> 1);
            if (ch < N) a.gn_part[((int64_t)m_tile * N + ch) * 2 + (idx & 1)] = t;
        }
    }
}

template <typename T, int TH, int BN, bool UP, int CK>
int launch_conv_halo2(ConvArgs a, hipStream_t st) {
    constexpr int NPW = (((TH + 2) * 18 + 512 / CK - 1) / (512 / CK) + 3) / 4;
    size_t lds = (size_t)2 * NPW * 4 * 512 * sizeof(T) + (size_t)3 * BN * CK * sizeof(T);
    const size_t stage = (size_t)TH * 16 * (BN + 8) * sizeof(T);
    if (stage > lds) lds = stage;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<T, TH, BN, UP, CK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    a.mt = a.B * ((a.H + TH - 1) / TH) * ((a.Wd + 15) / 16);
    a.nt = (a.Cout + BN - 1) / BN;
    a.cpt = a.Cin / CK;
    const int mt8 = (a.mt + 7) / 8 * 8;
    hipLaunchKernelGGL((conv3x3_halo_kernel<T, TH, BN, UP, CK>), dim3(mt8 * a.nt), dim3(256), lds, st, a);
    return mos_check_launch("conv3x3_nhwc(halo)");
}

template <typename T, int TH, int BN, int CK = CBK>
int launch_conv_halo(const ConvArgs& a, hipStream_t st) {
    return a.up ? launch_conv_halo2<T, TH, BN, true, CK>(a, st) : launch_conv_halo2<T, TH, BN, false, CK>(a, st);
}

// y = round(sum_z partial[z] + tbias + bias) (+ residual): the epilogue of the unsplit kernel on the summed partials
template <typename T>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvArgs a) {
    typedef typename MT<T>::v8 v8;
    const int N = a.Cout, nc = N / 8;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)a.M * nc) return;
    const int m = (int)(idx / nc), n = (int)(idx - (int64_t)m * nc) * 8;
    const float* p = a.partial + (int64_t)m * N + n;
    const int64_t zs = (int64_t)a.M * N;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int z = 0; z < a.ksplit; ++z) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p + z * zs), hi = *reinterpret_cast<const f32x4*>(p + z * zs + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += lo[e]; v[4 + e] += hi[e]; }
    }
    if (a.tbias != nullptr) {
        const v8 t = as_v8<T>(ld16(reinterpret_cast<const T*>(a.tbias) + (int64_t)(m / (a.H * a.Wd)) * N + n));
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)t[e];
    }
    v8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (T)(v[e] + (a.bias != nullptr ? a.bias[n + e] : 0.f));
    const int64_t off = (int64_t)m * N + n;
    if (a.R != nullptr) {
        const v8 r = as_v8<T>(ld16(reinterpret_cast<const T*>(a.R) + off));
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (T)((float)o[e] + (float)r[e]);
    }
    st16(reinterpret_cast<T*>(a.Y) + off, from_v8<T>(o));
}

// split-K plan: 64 x 64 tiles (3 workgroups per CU), split where that tiling alone gives <= 320 tiles, ~640 workgroups, >= 6 K
// tiles each: batch-4 16x16 67.7 us, 8x8 25.7 us; 512x768 sample 16x24 49.8 us, 8x12 23.1 us (MIOpen: 77.1 / 38.1 / 57.9 / 34.0;
// profiles/r04_kernel_bench_ff_gn_conv.txt). A 128-row x 128-wide tiling (half the L2 -> LDS operand traffic per flop, one
// workgroup per CU) was measured and lost (76.8 / 30.0 / 60.7 / 22.9 us, profiles/r04_kernel_bench_conv_splitk_128wide_tiles.txt:
// occupancy beats operand reuse here) -- removed in round 5, like the 256-row tiles of the unsplit form (profiles/r05c1_wide_tiles.txt).
inline int conv_ksplit(int M, int Cout, int Cin, int* kt_per) {
    const int nk = 9 * (Cin / 64);
    if (nk < 36) return 1;
    const int64_t tiles = (int64_t)((M + 63) / 64) * ((Cout + 63) / 64);
    if (tiles > 320) return 1;
    int64_t ks = (640 + tiles - 1) / tiles;
    if (ks > nk / 6) ks = nk / 6;               // at least 6 K tiles per workgroup
    if (ks < 2) return 1;
    const int per = (nk + (int)ks - 1) / (int)ks;
    *kt_per = per;
    return (nk + per - 1) / per;
}

template <typename T, int BM, int BN, bool UP, int S2 = 0>
int launch_conv_split(ConvArgs a, hipStream_t st) {
    constexpr int NS = 3;
    const size_t lds = (size_t)NS * (BM + BN) * CBK * sizeof(T);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_nhwc_kernel<T, BM, BN, UP, NS, true, S2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    a.mt = (a.M + BM - 1) / BM;
    a.nt = (a.Cout + BN - 1) / BN;
    const int units8 = (a.nt * a.ksplit + 7) / 8;
    hipLaunchKernelGGL((conv3x3_nhwc_kernel<T, BM, BN, UP, NS, true, S2>), dim3(8 * units8 * a.mt), dim3(256), lds, st, a);
    int rc = mos_check_launch("conv3x3_nhwc(split-K)");
    if (rc) return rc;
    const int64_t chunks = (int64_t)a.M * (a.Cout / 8);
    hipLaunchKernelGGL((conv_splitk_reduce_kernel<T>), dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, a);
    return mos_check_launch("conv_splitk_reduce");
}

template <typename T, int BM, int BN, bool UP, int NS, int S2 = 0>
int launch_conv_cfg2(ConvArgs a, hipStream_t st) {
    size_t lds = (size_t)NS * (BM + BN) * CBK * sizeof(T);
    const size_t stage = (size_t)BM * (BN + 8) * sizeof(T);
    if (stage > lds) lds = stage;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_nhwc_kernel<T, BM, BN, UP, NS, false, S2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    a.mt = (a.M + BM - 1) / BM;
    a.nt = (a.Cout + BN - 1) / BN;
    const int mt8 = (a.mt + 7) / 8 * 8;
    hipLaunchKernelGGL((conv3x3_nhwc_kernel<T, BM, BN, UP, NS, false, S2>), dim3(mt8 * a.nt), dim3(256), lds, st, a);
    return mos_check_launch("conv3x3_nhwc");
}

// stride-2 form (down-samplers): raster kernel only (the halo form's LDS image assumes unit stride). Tiles as the unsplit
// stride-1 dispatch chooses them; the low-resolution UNet down-samplers (<= 320 tiles, deep K) take the split-K form.
template <typename T, int S2>
int launch_conv_s2(ConvArgs a, hipStream_t st) {
    char key[112];
    int kt_per = 0;
    const int ks = a.partial != nullptr ? conv_ksplit(a.M, a.Cout, a.Cin, &kt_per) : 1;
    snprintf(key, sizeof(key), "%s B%d %dx%d Cin%d Cout%d stride2%s%s", std::is_same<T, f16_t>::value ? "f16" : "bf16", a.B, a.H, a.Wd,
             a.Cin, a.Cout, S2 == 2 ? " pad(0,1,0,1)" : "", ks > 1 ? " splitK" : "");
    MosProfScope prof(st, "conv3x3", key, 2.0 * a.M * (double)a.Cout * 9.0 * a.Cin,
                      2.0 * ((double)a.B * a.Hin * a.Win * a.Cin + 9.0 * a.Cin * a.Cout + (double)a.M * a.Cout));
    if (ks > 1) {
        a.ksplit = ks; a.kt_per = kt_per;
        return launch_conv_split<T, 64, 64, false, S2>(a, st);
    }
    int bn = (a.Cout % 128 == 0) ? 128 : 64, bm = 128;
    auto tiles = [&](int m, int n) { return (int64_t)((a.M + m - 1) / m) * ((a.Cout + n - 1) / n); };
    if (tiles(bm, bn) < 384) bm = 64;
    if (tiles(bm, bn) < 256 && bn == 128) bn = 64;
    if (bm == 128 && bn == 128) return launch_conv_cfg2<T, 128, 128, false, 2, S2>(a, st);
    if (bm == 128) return launch_conv_cfg2<T, 128, 64, false, 2, S2>(a, st);
    if (bn == 128) return launch_conv_cfg2<T, 64, 128, false, 2, S2>(a, st);
    return launch_conv_cfg2<T, 64, 64, false, 2, S2>(a, st);
}

template <typename T, int BM, int BN, int NS>
int launch_conv_cfg(const ConvArgs& a, hipStream_t st) {
    return a.up ? launch_conv_cfg2<T, BM, BN, true, NS>(a, st) : launch_conv_cfg2<T, BM, BN, false, NS>(a, st);
}

// Workgroup count below which the K loop is latency-bound and the DMA ring replaces the double buffer.
constexpr int RING_MAX_WG = 640;

// Dispatch (round 5, profiles/r05c1_wide_tiles.txt, same box, forward / backward-data in us):
//   * low-resolution levels (<= 320 tiles of 64 x 64, K >= 36 tiles): split-K raster form;
//   * maps at least 16 wide and 8 high: the HALO-staged form, 8 x 16 pixels x 64 outputs per workgroup (two workgroups per CU):
//     B4 320->320 64x64 46/48 -> 38/40, 960->320 133/129 -> 109/104, 640->640 32x32 55/58 -> 43/43, 1920->640 162/101 -> 123/104,
//     128->128 512x512 416/425 -> 388/354, B2 640->640 32x48 42/46 -> 32/32; the 16 x 16 x 128 tile only where >= 512 input channels
//     meet >= 256 such tiles (the VAE's 512-channel stages: 324/305 -> 295/281 at 128x128, 82/86 -> 74/77 at 64x64);
//     8 x 16 x 128 and 16 x 16 x 64 tiles, and 256-row tiles of the raster form, lost everywhere and are gone; so did an
//     8 x 16 x 160 tile (one round of 256 workgroups at level 0 instead of 640 on 512 slots: 41.0 vs 39.3 us, one wave per SIMD
//     leaves the fragment reads exposed; profiles/r05c3_kernel_bench_conv_forms.txt);
//     same-box whole step, raster form -> this dispatch: 36.5 -> 35.3 ms (109.6 -> 113.3 images/s), conv3x3 11.0 -> 9.9 ms / step,
//     regional sample (latent out) 406.2 -> 386 ms, conv3x3 134.5 -> 112.6 ms / sample (profiles/r05c3_ab_same_box_conv_forms.txt);
//     later in round 5 (profiles/r05c11_*, r05c12_*; same box each): the tap loop's addressing moved off the VALU (scalar-offset
//     DMA, zero-record descriptors past the end, fragment reads = 8 lane patterns + immediates: ~1 v_add per MFMA -> none, 207 -> 90
//     VGPRs): conv3x3 10.05 -> 9.65 ms / step, 113.7 -> 115.0 images/s; then the 128-multiple-output maps (VAE stages) on
//     16 x 16 x 128 tiles with 32-CHANNEL chunks (73.7 KB of LDS: two workgroups per CU, where the 64-channel 16 x 16 x 128 tile
//     had one): B4 128->128 512x512 395/347 -> 330/289, 256->256 256x256 299/286 -> 244/237, 512->512 128x128 290/260 -> 261/266,
//     B1 512->512 256x384 424/376 -> 390/346; whole step 114.3 -> 116.8 images/s, conv3x3 9.66 -> 8.92 ms. 32-channel chunks on the
//     8 x 16 x 64 / 8 x 16 x 128 tiles of the UNet maps (3-4 workgroups per CU) measured level with or behind the 64-channel
//     8 x 16 x 64 tile (B2 640->640 32x48 30 -> 38 / 47 us) and are not built;
//   * what is left (maps narrower than 16 pixels that are not split): raster form.
// which halo tile (if any) launch_conv gives a shape: 2 = 16 x 16 x 128 on 32-channel chunks, 1 = 8 x 16 x 64, 0 = not the halo form
inline int conv_halo_form(int B, int H, int Wd, int Cin, int Cout, bool has_ws) {
    int kt_per = 0;
    if (has_ws && conv_ksplit(B * H * Wd, Cout, Cin, &kt_per) > 1) return 0;
#ifdef MOS_CONV_NO_HALO
    return 0;
#endif
    if (!(Wd >= 16 && H >= 8)) return 0;
    const int64_t t16 = (int64_t)B * ((H + 15) / 16) * ((Wd + 15) / 16) * (Cout / 128);
    if (Cout % 128 == 0 && H >= 16 && ((Cin >= 512 && t16 >= 256) || (Cin <= 256 && t16 >= 512))) return 2;
    return 1;
}

template <typename T>
int launch_conv(ConvArgs a, hipStream_t st) {
    char key[112];
    int kt_per = 0;
    const int ks = a.partial != nullptr ? conv_ksplit(a.M, a.Cout, a.Cin, &kt_per) : 1;
    if (a.gn_part != nullptr && conv_halo_form(a.B, a.H, a.Wd, a.Cin, a.Cout, a.partial != nullptr) == 0)
        return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_conv3x3_nhwc_gn: this shape does not take the halo form (ask mos_conv3x3_gn_tiles first)");
    snprintf(key, sizeof(key), "%s B%d %dx%d Cin%d Cout%d%s%s%s%s", std::is_same<T, f16_t>::value ? "f16" : "bf16", a.B, a.H, a.Wd,
             a.Cin, a.Cout, a.up ? " up2x" : "", a.tbias ? " +tbias" : "", a.R ? " +res" : "", ks > 1 ? " splitK" : "");
    MosProfScope prof(st, "conv3x3", key, 2.0 * a.M * (double)a.Cout * 9.0 * a.Cin,
                      2.0 * ((double)a.M * a.Cin / (a.up ? 4 : 1) + 9.0 * a.Cin * a.Cout + (double)a.M * a.Cout * (a.R ? 2 : 1)));
    if (ks > 1) {
        a.ksplit = ks; a.kt_per = kt_per;
        return a.up ? launch_conv_split<T, 64, 64, true>(a, st) : launch_conv_split<T, 64, 64, false>(a, st);
    }
#ifndef MOS_CONV_NO_HALO          // (variant build for the same-box A/B against the raster form, csrc/build.sh)
    if (a.Wd >= 16 && a.H >= 8) {
        // 128-multiples of output channels on big maps (the VAE's stages; level 1 of a 1024 x 2048 sample): 16 x 16 x 128 tiles on
        // 32-channel chunks -- two workgroups per CU, 32 MFMAs per wave and barrier, 0.375 fragment reads per MFMA
        const int64_t t16 = (int64_t)a.B * ((a.H + 15) / 16) * ((a.Wd + 15) / 16) * (a.Cout / 128);
        if (a.Cout % 128 == 0 && a.H >= 16 && ((a.Cin >= 512 && t16 >= 256) || (a.Cin <= 256 && t16 >= 512)))
            return launch_conv_halo<T, 16, 128, 32>(a, st);
        return launch_conv_halo<T, 8, 64>(a, st);
    }
#endif
    int bn = (a.Cout % 128 == 0) ? 128 : 64, bm = 128;
    auto tiles = [&](int m, int n) { return (int64_t)((a.M + m - 1) / m) * ((a.Cout + n - 1) / n); };
    if (tiles(bm, bn) < 384) bm = 64;
    if (tiles(bm, bn) < 256 && bn == 128) bn = 64;
    if (bm == 128 && bn == 128) return launch_conv_cfg<T, 128, 128, 2>(a, st);
    if (bm == 128) return launch_conv_cfg<T, 128, 64, 2>(a, st);
    const bool ring = tiles(bm, bn) <= RING_MAX_WG;
    if (bn == 128) return ring ? launch_conv_cfg<T, 64, 128, 3>(a, st) : launch_conv_cfg<T, 64, 128, 2>(a, st);
    return ring ? launch_conv_cfg<T, 64, 64, 4>(a, st) : launch_conv_cfg<T, 64, 64, 2>(a, st);
}

}  // namespace

extern "C" {

int64_t mos_conv3x3_nhwc_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 8) return 0;
    int kt_per = 0;
    const int64_t M = (int64_t)B * H * W;
    if (M > (1 << 20)) return 0;
    const int ks = conv_ksplit((int)M, Cout, Cin, &kt_per);
    return ks > 1 ? (int64_t)ks * M * Cout * (int64_t)sizeof(float) : 0;
}

int mos_conv3x3_nhwc(const void* x, const void* w, const float* bias, const void* tbias, const void* residual, void* y,
                     int B, int H, int W, int Cin, int Cout, int upsample2x, int dtype, void* stream) {
    return mos_conv3x3_nhwc_ws(x, w, bias, tbias, residual, y, B, H, W, Cin, Cout, upsample2x, dtype, nullptr, stream);
}

int mos_conv3x3_nhwc_ws(const void* x, const void* w, const float* bias, const void* tbias, const void* residual, void* y,
                        int B, int H, int W, int Cin, int Cout, int upsample2x, int dtype, void* ws, void* stream) {
    return mos_conv3x3_nhwc_gn(x, w, bias, tbias, residual, y, B, H, W, Cin, Cout, upsample2x, dtype, ws, nullptr, stream);
}

int mos_conv3x3_nhwc_gn(const void* x, const void* w, const float* bias, const void* tbias, const void* residual, void* y,
                        int B, int H, int W, int Cin, int Cout, int upsample2x, int dtype, void* ws, void* gn_part, void* stream) {
    return mos_conv3x3_nhwc_px(x, (int64_t)Cin, w, bias, tbias, residual, y, B, H, W, Cin, Cout, upsample2x, dtype, ws, gn_part, stream);
}

/* The same with x read in place from a channel slice of a wider channels-last tensor (round 6): x_pixel_stride = elements between
 * consecutive pixels of x (>= Cin, a multiple of 8; rows and images follow at W and H * W pixels). This is how the dX convolution
 * of a layer whose output went into a torch.cat reads its slice of the concatenation's gradient without a contiguous copy. */
int mos_conv3x3_nhwc_px(const void* x, int64_t x_pixel_stride, const void* w, const float* bias, const void* tbias,
                        const void* residual, void* y, int B, int H, int W, int Cin, int Cout, int upsample2x, int dtype, void* ws,
                        void* gn_part, void* stream) {
    MOS_REQUIRE(x && w && y, "mos_conv3x3_nhwc: NULL argument");
    MOS_REQUIRE(x_pixel_stride >= Cin && x_pixel_stride % 8 == 0 && ((uint64_t)x & 15) == 0 &&
                    (int64_t)B * H * W * x_pixel_stride * 2 < (1ll << 31),
                "mos_conv3x3_nhwc_px: x pixel stride %lld (need >= Cin, a multiple of 8, a 16-byte aligned base, < 2 GiB in all)",
                (long long)x_pixel_stride);
    MOS_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 64 == 0 && Cout % 8 == 0,
                "mos_conv3x3_nhwc: B=%d H=%d W=%d Cin=%d Cout=%d (need Cin %% 64 == 0, Cout %% 8 == 0)", B, H, W, Cin, Cout);
    MOS_REQUIRE(!upsample2x || (H % 2 == 0 && W % 2 == 0), "mos_conv3x3_nhwc: upsample2x needs even output H, W");
    MOS_REQUIRE((int64_t)B * H * W * (int64_t)(Cin > Cout ? Cin : Cout) * 2 < (1ll << 31) && (int64_t)Cout * 9 * Cin * 2 < (1ll << 31),
                "mos_conv3x3_nhwc: tensor exceeds the 2 GiB range of one buffer descriptor");
    ConvArgs a;
    a.X = x; a.W = w; a.bias = bias; a.tbias = tbias; a.R = residual; a.Y = y;
    a.B = B; a.H = H; a.Wd = W; a.Cin = Cin; a.Cout = Cout; a.M = B * H * W; a.cpt = Cin / 64; a.up = upsample2x ? 1 : 0;
    a.mt = a.nt = 0; a.Hin = H; a.Win = W; a.gn_part = (float*)gn_part; a.ldx = (int)x_pixel_stride;
    a.ksplit = 1; a.kt_per = 0; a.partial = (float*)ws;
    if (dtype == MOS_F16) return launch_conv<f16_t>(a, (hipStream_t)stream);
    if (dtype == MOS_BF16) return launch_conv<bf16_t>(a, (hipStream_t)stream);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_conv3x3_nhwc: dtype %d", dtype);
}

int mos_conv3x3_gn_tiles(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 8) return 0;
    const int form = conv_halo_form(B, H, W, Cin, Cout, true);
    if (form == 0) return 0;
    const int th = form == 2 ? 16 : 8;
    return ((H + th - 1) / th) * ((W + 15) / 16);
}

int mos_conv3x3_s2_nhwc(const void* x, const void* w, const float* bias, void* y, int B, int Hin, int Win, int Cin, int Cout,
                        int pad_mode, int dtype, void* ws, void* stream) {
    MOS_REQUIRE(x && w && y, "mos_conv3x3_s2_nhwc: NULL argument");
    MOS_REQUIRE(B > 0 && Hin > 1 && Win > 1 && Cin > 0 && Cout > 0 && Cin % 64 == 0 && Cout % 8 == 0,
                "mos_conv3x3_s2_nhwc: B=%d Hin=%d Win=%d Cin=%d Cout=%d (need Cin %% 64 == 0, Cout %% 8 == 0)", B, Hin, Win, Cin, Cout);
    MOS_REQUIRE(pad_mode == 1 || pad_mode == 2, "mos_conv3x3_s2_nhwc: pad_mode %d (1: padding 1; 2: zero row / column at bottom / right)",
                pad_mode);
    // output size of a 3x3 / stride-2 window: padding 1 -> floor((Hin - 1) / 2) + 1; (0, 1, 0, 1) pad -> floor((Hin - 2) / 2) + 1
    const int H = pad_mode == 1 ? (Hin - 1) / 2 + 1 : (Hin - 2) / 2 + 1, W = pad_mode == 1 ? (Win - 1) / 2 + 1 : (Win - 2) / 2 + 1;
    MOS_REQUIRE((int64_t)B * Hin * Win * (int64_t)Cin * 2 < (1ll << 31) && (int64_t)B * H * W * (int64_t)Cout * 2 < (1ll << 31) &&
                    (int64_t)Cout * 9 * Cin * 2 < (1ll << 31),
                "mos_conv3x3_s2_nhwc: tensor exceeds the 2 GiB range of one buffer descriptor");
    ConvArgs a;
    a.X = x; a.W = w; a.bias = bias; a.tbias = nullptr; a.R = nullptr; a.Y = y;
    a.B = B; a.H = H; a.Wd = W; a.Cin = Cin; a.Cout = Cout; a.M = B * H * W; a.cpt = Cin / 64; a.up = 0;
    a.mt = a.nt = 0; a.Hin = Hin; a.Win = Win; a.gn_part = nullptr; a.ldx = Cin;
    a.ksplit = 1; a.kt_per = 0; a.partial = (float*)ws;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MOS_F16) return pad_mode == 1 ? launch_conv_s2<f16_t, 1>(a, st) : launch_conv_s2<f16_t, 2>(a, st);
    if (dtype == MOS_BF16) return pad_mode == 1 ? launch_conv_s2<bf16_t, 1>(a, st) : launch_conv_s2<bf16_t, 2>(a, st);
    return mos_set_error(MOS_ERR_UNSUPPORTED, "mos_conv3x3_s2_nhwc: dtype %d", dtype);
}

}  // extern "C"

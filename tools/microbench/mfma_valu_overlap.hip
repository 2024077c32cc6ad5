// Micro-benchmark: do the MFMA and VALU phases of two waves on one SIMD overlap?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
// Each wave alternates an M phase (NM dependent-chain v_mfma_f32_32x32x16_f16 on 2 accumulators) and a V phase (NE v_exp_f32
// + NV v_fma_f32 on the accumulator values, as a softmax would). Variants:
//   free256 : 256-thread blocks, 2 per CU (two waves per SIMD from DIFFERENT blocks, free running)
//   free512 : 512-thread blocks, 1 per CU, no synchronisation between the halves
//   alt512  : 512-thread blocks; waves 4-7 run one phase behind waves 0-3 and an s_barrier closes every phase, so that on each
//             SIMD one wave is in its M phase while its partner is in its V phase
//   m_only / v_only : the phases alone (256-thread blocks, 2 per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f16v;

#ifndef ACC_AGPR
#define ACC_AGPR 0
#endif
template <int NM>
__device__ __forceinline__ void phase_m(f16v& a0, f16v& a1, h8 x, h8 y) {
#pragma unroll
    for (int i = 0; i < NM / 2; ++i) {
#if ACC_AGPR      // accumulators in the AccVGPR half of the register file ("a" constraint)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(a0) : "v"(x), "v"(y));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(a1) : "v"(y), "v"(x));
#else
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
#endif
    }
}
template <int NE, int NV>
__device__ __forceinline__ void phase_v(f16v& a0, f16v& a1, float c) {
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        if (i & 16) a1[i & 15] = __builtin_amdgcn_exp2f(a1[i & 15] * c - 1.0f);
        else a0[i & 15] = __builtin_amdgcn_exp2f(a0[i & 15] * c - 1.0f);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (i & 16) a1[i & 15] = a1[i & 15] * c + 0.25f;
        else a0[i & 15] = a0[i & 15] * c + 0.25f;
    }
}

template <int NM, int NE, int NV>
__device__ __forceinline__ void phase_mix(f16v& a0, f16v& a1, f16v& b0, f16v& b1, h8 x, h8 y, float c) {
    // one wave: NM MFMAs with (NE + NV) / NM independent VALU ops placed after each (sched_group_barrier pins the interleave)
    constexpr int PER = (NE + NV + NM - 1) / NM;
    constexpr int EPER = (NE + NM - 1) / NM;
    int vi = 0, ei = 0;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        if (i & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int r = (i * PER + j) & 31;
            if (j < EPER && ei < NE) {
                if (r & 16) b1[r & 15] = __builtin_amdgcn_exp2f(b1[r & 15] * c - 1.0f); else b0[r & 15] = __builtin_amdgcn_exp2f(b0[r & 15] * c - 1.0f);
                ++ei;
            } else if (vi < NV) {
                if (r & 16) b1[r & 15] = b1[r & 15] * c + 0.25f; else b0[r & 15] = b0[r & 15] * c + 0.25f;
                ++vi;
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, 2 * PER, 0);
    }
}

template <int MODE, int NM, int NE, int NV>   // MODE 0 free, 1 alternate, 2 M only, 3 V only, 4 split roles, 5 in-wave interleave
__global__ void k(float* out, int iters, float c) {
    f16v a0, a1;
    for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f + i; a1[i] = i * 0.5f; }
    h8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(0.01f * (threadIdx.x & 7) + 0.001f * i); y[i] = (_Float16)(0.02f * i); }
    const bool second = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;
    if (MODE == 1) {
        if (second) { __builtin_amdgcn_s_barrier(); }        // one phase behind
        for (int it = 0; it < iters; ++it) {
            phase_m<NM>(a0, a1, x, y);
            __builtin_amdgcn_s_barrier();
            phase_v<NE, NV>(a0, a1, c);
            __builtin_amdgcn_s_barrier();
        }
        if (!second) { __builtin_amdgcn_s_barrier(); }
    } else if (MODE == 6) {          // one wave per SIMD, M only, both waves' work
        for (int it = 0; it < 2 * iters; ++it) phase_m<NM>(a0, a1, x, y);
    } else if (MODE == 7) {          // one wave per SIMD, V only, both waves' work
        for (int it = 0; it < 2 * iters; ++it) phase_v<NE, NV>(a0, a1, c);
    } else if (MODE == 4) {
        if (second) { for (int it = 0; it < 2 * iters; ++it) phase_v<NE, NV>(a0, a1, c); }
        else { for (int it = 0; it < 2 * iters; ++it) phase_m<NM>(a0, a1, x, y); }
    } else if (MODE == 5) {
        f16v b0 = a0, b1 = a1;
        for (int it = 0; it < 2 * iters; ++it) phase_mix<NM, NE, NV>(a0, a1, b0, b1, x, y, c);
        for (int i = 0; i < 16; ++i) { a0[i] += b0[i]; a1[i] += b1[i]; }
    } else {
        for (int it = 0; it < iters; ++it) {
            if (MODE != 3) phase_m<NM>(a0, a1, x, y);
            if (MODE != 2) phase_v<NE, NV>(a0, a1, c);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NM, int NE, int NV>
float run(int threads, int blocks, int iters, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NM, NE, NV>), dim3(blocks), dim3(threads), 0, 0, out, 10, 0.999f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, NM, NE, NV>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.999f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    return best * 1e3f;
}

template <int NM, int NE, int NV>
void suite(float* out) {
    const int iters = 2000;
    // work per SIMD is identical in every variant: 2 waves x iters phases
    const float m = run<2, NM, NE, NV>(256, 512, iters, out), v = run<3, NM, NE, NV>(256, 512, iters, out);
    const float f256 = run<0, NM, NE, NV>(256, 512, iters, out), f512 = run<0, NM, NE, NV>(512, 256, iters, out);
    const float a512 = run<1, NM, NE, NV>(512, 256, iters, out);
    const float split = run<4, NM, NE, NV>(512, 256, iters, out);     // same total work per SIMD: 2*iters M phases + 2*iters V phases
    const float mix1 = run<5, NM, NE, NV>(256, 256, iters, out);      // one wave per SIMD doing both waves' work, interleaved
    const float m1 = run<6, NM, NE, NV>(256, 256, iters, out), v1 = run<7, NM, NE, NV>(256, 256, iters, out);
    printf("   one wave per SIMD doing both waves' work: M %7.1f  V %7.1f\n", m1, v1);
    const double clk = 2.4e3;  // cycles per us at the nominal clock (for orientation only)
    printf("NM=%2d NE=%2d NV=%2d | us: M-only %7.1f  V-only %7.1f  free256 %7.1f  free512 %7.1f  alt512 %7.1f  split-roles %7.1f  1wave-interleaved %7.1f | per-iteration cycles@2.4GHz per SIMD pair: "
           "M %5.0f V %5.0f free256 %5.0f alt512 %5.0f\n", NM, NE, NV, m, v, f256, f512, a512, split, mix1, m * clk / iters, v * clk / iters,
           f256 * clk / iters, a512 * clk / iters);
}

int main() {
    printf("accumulators in %s\n", ACC_AGPR ? "AccVGPRs (inline asm)" : "arch VGPRs (compiler's choice)");
    float* out; hipMalloc(&out, 512 * 512 * sizeof(float));
    suite<14, 16, 48>(out);    // one 32-query half tile of the d = 40 dK/dV kernel
    suite<14, 32, 64>(out);
    suite<14, 0, 96>(out);
    suite<28, 32, 96>(out);
    suite<8, 16, 32>(out);
    hipFree(out);
    return 0;
}

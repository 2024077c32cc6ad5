// Micro-benchmark: how many independent VALU instructions hide behind one v_mfma_f32_32x32x16_f16 of the SAME wave?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_filler mfma_filler.hip && ./mfma_filler
// One wave per SIMD (256-thread blocks, one per CU). The loop body is NM MFMAs (two accumulators, alternating), each followed by
// K filler instructions on registers the MFMAs never touch, written as ONE inline-asm block so that the placement is
// exactly what is written. Fillers: v_fma_f32 (plain VALU), v_exp_f32 (transcendental), v_cvt_pk_f16_f32, v_max3_f32,
// v_pk_mul_f32, and a softmax-like mix (1 exp + 1 fma + 1/2 max3 + 1/2 cvt per element).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f16v;

#define REP4(x) x x x x
#define MFMA0 "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\t"
#define MFMA1 "v_mfma_f32_32x32x16_f16 %1, %3, %2, %1\n\t"
#define FMA "v_fma_f32 %4, %4, %6, %7\n\t"
#define FMB "v_fma_f32 %5, %5, %6, %7\n\t"
#define EXA "v_exp_f32 %4, %4\n\t"
#define EXB "v_exp_f32 %5, %5\n\t"
#define CVT "v_cvt_pk_f16_f32 %8, %4, %5\n\t"
#define MX3 "v_max3_f32 %9, %4, %5, %9\n\t"
#define PKM "v_pk_mul_f32 %10, %10, %11\n\t"

template <int KIND, int K>
__global__ void k(float* out, int iters) {
    f16v a0, a1;
    for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f + i; a1[i] = i * 0.5f; }
    h8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(0.01f * (threadIdx.x & 7) + 0.001f * i); y[i] = (_Float16)(0.02f * i); }
    float f0 = threadIdx.x * 1e-4f, f1 = 0.5f, c = 0.999f, d = 1e-3f, mx = 0.f;
    unsigned pk = 0;
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 p0 = {1.f, 2.f}, p1 = {0.999f, 1.001f};
#define BODY(FILL) asm volatile(MFMA0 FILL MFMA1 FILL MFMA0 FILL MFMA1 FILL \
        : "+v"(a0), "+v"(a1) : "v"(x), "v"(y), "v"(f0), "v"(f1), "v"(c), "v"(d), "v"(pk), "v"(mx), "v"(p0), "v"(p1));
    for (int it = 0; it < iters; ++it) {
        if constexpr (K == 0) { BODY("") }
        else if constexpr (KIND == 0) {           // K plain v_fma
            if constexpr (K == 1) { BODY(FMA) } else if constexpr (K == 2) { BODY(FMA FMB) } else if constexpr (K == 3) { BODY(FMA FMB FMA) }
            else if constexpr (K == 4) { BODY(FMA FMB FMA FMB) } else if constexpr (K == 5) { BODY(FMA FMB FMA FMB FMA) }
            else if constexpr (K == 6) { BODY(FMA FMB FMA FMB FMA FMB) } else if constexpr (K == 8) { BODY(FMA FMB FMA FMB FMA FMB FMA FMB) }
            else if constexpr (K == 12) { BODY(FMA FMB FMA FMB FMA FMB FMA FMB FMA FMB FMA FMB) }
        } else if constexpr (KIND == 1) {         // K v_exp
            if constexpr (K == 1) { BODY(EXA) } else if constexpr (K == 2) { BODY(EXA EXB) } else if constexpr (K == 3) { BODY(EXA EXB EXA) }
            else if constexpr (K == 4) { BODY(EXA EXB EXA EXB) } else if constexpr (K == 6) { BODY(EXA EXB EXA EXB EXA EXB) }
        } else if constexpr (KIND == 2) {         // softmax-like mix per MFMA: K elements: exp + fma each, + K/2 max3 + K/2 cvt
            if constexpr (K == 2) { BODY(FMA EXA FMB EXB MX3 CVT) }
            else if constexpr (K == 4) { BODY(FMA EXA FMB EXB MX3 CVT FMA EXA FMB EXB MX3 CVT) }
        } else if constexpr (KIND == 3) {         // K v_pk_mul_f32
            if constexpr (K == 2) { BODY(PKM PKM) } else if constexpr (K == 4) { BODY(PKM PKM PKM PKM) }
        }
    }
    float s = f0 + f1 + mx + p0[0] + p0[1] + (float)pk;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int K>
void run(float* out, const char* what, float base) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(256), 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    const double ns_per_mfma = best * 1e6 / (iters * 4.0);
    printf("%-34s %8.1f us   %6.2f ns per MFMA+fillers  (MFMA alone %.2f ns -> +%.2f ns for the fillers)\n", what, best * 1e3,
           ns_per_mfma, base, ns_per_mfma - base);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 256 * sizeof(float));
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<0, 0>), dim3(256), dim3(256), 0, 0, out, 10); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL((k<0, 0>), dim3(256), dim3(256), 0, 0, out, iters); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const float base = ms * 1e6 / (iters * 4.0);
    printf("one wave per SIMD, v_mfma_f32_32x32x16_f16 alone: %.2f ns each (32 cycles at %.2f GHz)\n", base, 32.0 / base);
    run<0, 1>(out, "1 v_fma per MFMA", base); run<0, 2>(out, "2 v_fma", base); run<0, 3>(out, "3 v_fma", base); run<0, 4>(out, "4 v_fma", base);
    run<0, 5>(out, "5 v_fma", base); run<0, 6>(out, "6 v_fma", base); run<0, 8>(out, "8 v_fma", base); run<0, 12>(out, "12 v_fma", base);
    run<1, 1>(out, "1 v_exp per MFMA", base); run<1, 2>(out, "2 v_exp", base); run<1, 3>(out, "3 v_exp", base); run<1, 4>(out, "4 v_exp", base);
    run<1, 6>(out, "6 v_exp", base);
    run<2, 2>(out, "softmax mix, 2 elements per MFMA", base); run<2, 4>(out, "softmax mix, 4 elements per MFMA", base);
    run<3, 2>(out, "2 v_pk_mul_f32", base); run<3, 4>(out, "4 v_pk_mul_f32", base);
    (void)hipFree(out);
    return 0;
}

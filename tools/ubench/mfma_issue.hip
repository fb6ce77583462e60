// How many independent non-MFMA instructions fit behind one v_mfma_f32_16x16x4_f32 (8 passes = 32 cycles) of the SAME wave, one wave per
// SIMD?  Each kernel runs ITER x 32 MFMAs (32 independent accumulators) with K extra instructions of one kind after every MFMA.
// Prints cycles per MFMA (s_memtime based) — 32 = the matrix pipe never idles.   Build: hipcc --offload-arch=gfx950 -O3 mfma_issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define ITER 2000
template <int KIND, int K, int TPB>
__global__ __launch_bounds__(TPB) void k(float *out, const float *in, unsigned long long *cyc) {
    __shared__ float lds[4096];
    const int l = threadIdx.x;
    lds[l] = (float)l; lds[l + 512] = 1.f;
    __syncthreads();
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = in[l], b = in[l + 512];
    float v[8]; f32x2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = in[l + i]; p[i] = (f32x2){in[l + i], in[l + 8 + i]}; }
    const unsigned laddr = (unsigned)(size_t)((__attribute__((address_space(3))) const float *)lds) + 8u * (unsigned)(l & 63) + 1024u * (unsigned)(l >> 6);
    f32x4 g4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) g4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float *gp = in + 4 * l;
    float ld[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ld[i] = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
            if (KIND == 10) {
                if ((m & 7) == 7) {
#pragma unroll
                    for (int e = 0; e < 8 * K; ++e) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[e & 7]) : "v"(a));
                }
            }
#pragma unroll
            for (int e = 0; e < K; ++e) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[(m * K + e) & 7]) : "v"(a));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[(m * K + e) & 7]) : "v"(p[7 - ((m * K + e) & 7)]));
                if (KIND == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(p[(m * K + e) & 7]) : "v"(laddr));
                if (KIND == 5) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g4[(m * K + e) & 3]) : "v"(gp));
                if (KIND == 6) asm volatile("ds_write_b64 %0, %1" : : "v"(laddr), "v"(p[(m * K + e) & 7]));
                if (KIND == 3) asm volatile("s_nop 0");
                if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(v[(m * K + e) & 7]) : "v"(a));
            }
        }
        if (KIND == 2 || KIND == 6) asm volatile("s_waitcnt lgkmcnt(0)");
        if (KIND == 5) asm volatile("s_waitcnt vmcnt(0)");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1] + ld[i] + g4[i & 3][i & 3];
    out[blockIdx.x * TPB + l] = s;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND, int K, int TPB = 256>
void run(const char *name, float *out, float *in, unsigned long long *cyc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, K, TPB>), dim3(256), dim3(TPB), 0, 0, out, in, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, K, TPB>), dim3(256), dim3(TPB), 0, 0, out, in, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-14s waves/SIMD %d K=%d: %.3f ms  -> %.1f ns per MFMA, counter ticks per MFMA %.2f\n", name, TPB / 256, K, ms, ms * 1e6 / (ITER * 32.0), (double)c / (ITER * 32.0));
}
int main() {
    float *out, *in; unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&in, 4096 * 4); hipMalloc(&cyc, 8);
    hipMemset(in, 0, 4096 * 4);
#define ROW(KIND, NAME) run<KIND, 0>(NAME, out, in, cyc); run<KIND, 1>(NAME, out, in, cyc); run<KIND, 2>(NAME, out, in, cyc); run<KIND, 3>(NAME, out, in, cyc); \
    run<KIND, 4>(NAME, out, in, cyc); run<KIND, 6>(NAME, out, in, cyc); run<KIND, 8>(NAME, out, in, cyc);
    ROW(0, "v_fma_f32") ROW(10, "v_fma burst/8") ROW(2, "ds_read_b64") ROW(6, "ds_write_b64") ROW(5, "global_load_x4")
#define ROW2(KIND, NAME) run<KIND, 0, 512>(NAME, out, in, cyc); run<KIND, 1, 512>(NAME, out, in, cyc); run<KIND, 2, 512>(NAME, out, in, cyc); \
    run<KIND, 4, 512>(NAME, out, in, cyc); run<KIND, 8, 512>(NAME, out, in, cyc);
    ROW2(0, "v_fma_f32") ROW2(2, "ds_read_b64") ROW2(5, "global_load_x4")
    return 0;
}

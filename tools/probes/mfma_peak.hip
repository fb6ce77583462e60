// fp32 MFMA issue-rate probe: waves per SIMD x independent accumulator chains, operands in registers (no memory in the loop).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o tools/probes/mfma_peak ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS> void run(int wgs_per_cu, float *out) {
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * 8 * CHAINS * 4096.0;
    printf("chains %d, %d workgroup(s) of 4 waves per CU (= waves per SIMD): %.3f ms, %.1f TFLOP/s\n", CHAINS, wgs_per_cu, ms, flop / ms * 1e-9);
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w = 1; w <= 4; ++w) { run<1>(w, out); run<2>(w, out); run<4>(w, out); }
    return 0;
}

// Do two 1024-thread workgroups share a CU?  Each workgroup stamps the 100 MHz wall clock at its start, spins ~5 us and stamps
// its end; a grid of 313 workgroups either starts together (co-resident) or 57 of them start one spin later.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
template <int LDS, int REGS>
__global__ __launch_bounds__(1024) void probe(int *out, int spin) {
    __shared__ int s[LDS / 4];
    const int id = blockIdx.y * gridDim.x + blockIdx.x;
    float acc[REGS];
    for (int k = 0; k < REGS; ++k) acc[k] = threadIdx.x * 0.5f + k;
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) out[2 * id] = (int)t0;
    s[threadIdx.x % (LDS / 4)] = threadIdx.x;
    __syncthreads();
    while (wall_clock64() - t0 < spin)
        for (int k = 0; k < REGS; ++k) acc[k] = acc[k] * 1.0001f + s[(threadIdx.x + k) % (LDS / 4)];
    float r = 0;
    for (int k = 0; k < REGS; ++k) r += acc[k];
    if (threadIdx.x == 0) out[2 * id + 1] = (int)wall_clock64() + (r == 12345.f);
}
template <int LDS, int REGS>
void run(const char *name, dim3 grid) {
    const int n = grid.x * grid.y;
    int *d; hipMalloc(&d, n * 8);
    for (int it = 0; it < 3; ++it) { hipLaunchKernelGGL((probe<LDS, REGS>), grid, dim3(1024), 0, 0, d, 500); hipDeviceSynchronize(); }
    int *h = (int *)malloc(n * 8); hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost);
    int t0 = h[0]; for (int i = 0; i < n; ++i) t0 = std::min(t0, h[2 * i]);
    int late = 0, mx = 0; for (int i = 0; i < n; ++i) { late += (h[2 * i] - t0) > 200; mx = std::max(mx, h[2 * i] - t0); }
    int l256 = 0; for (int i = 256; i < n; ++i) l256 = std::max(l256, h[2 * i] - t0);
    printf("%-40s grid %3d x %2d: %3d of %3d workgroups start more than 2 us late; latest start %.2f us (ids >= 256: %.2f)\n", name, grid.x, grid.y, late, n, mx / 100.0, l256 / 100.0);
    hipFree(d); free(h);
}
template <int LDS, int REGS>
void run_after(const char *name, dim3 grid) {     // the same, launched right behind a 512-workgroup kernel with 76 KB of LDS each
    const int n = grid.x * grid.y;
    int *d, *d0; hipMalloc(&d, n * 8); hipMalloc(&d0, 512 * 8);
    for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL((probe<77824, 40>), dim3(512), dim3(1024), 0, 0, d0, 1500);
        hipLaunchKernelGGL((probe<LDS, REGS>), grid, dim3(1024), 0, 0, d, 500);
        hipDeviceSynchronize();
    }
    int *h = (int *)malloc(n * 8); hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost);
    int h0[1024]; hipMemcpy(h0, d0, 512 * 8, hipMemcpyDeviceToHost);
    int e0 = h0[1]; for (int i = 0; i < 512; ++i) e0 = std::max(e0, h0[2 * i + 1]);
    int t0 = h[0]; for (int i = 0; i < n; ++i) t0 = std::min(t0, h[2 * i]);
    int late = 0; for (int i = 0; i < n; ++i) late += (h[2 * i] - t0) > 200;
    printf("%-40s grid %3d x %2d behind a 512-wg kernel: gap %.2f us, %3d of %3d workgroups start more than 2 us late\n", name, grid.x, grid.y,
           (t0 - e0) / 100.0, late, n);
    hipFree(d); hipFree(d0); free(h);
}
int main() {
    run_after<16384, 40>("LDS 16K, ~56 VGPR", dim3(20, 16));
    run_after<16384, 40>("LDS 16K, ~56 VGPR", dim3(313, 1));
    run_after<16384, 40>("LDS 16K, ~56 VGPR", dim3(256, 1));
    run<16384, 24>("LDS 16K, ~40 VGPR", dim3(313, 1));
    run<16384, 24>("LDS 16K, ~40 VGPR", dim3(20, 16));
    run<16384, 24>("LDS 16K, ~40 VGPR", dim3(512, 1));
    run<16384, 40>("LDS 16K, ~56 VGPR", dim3(313, 1));
    run<16384, 40>("LDS 16K, ~56 VGPR", dim3(512, 1));
    run<65536, 24>("LDS 64K, ~40 VGPR", dim3(512, 1));
    run<4096, 8>("LDS 4K, few VGPR", dim3(512, 1));
    run<4096, 8>("LDS 4K, few VGPR", dim3(1024, 1));
    return 0;
}

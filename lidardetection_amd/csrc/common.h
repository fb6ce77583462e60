// Shared host/device helpers for the gfx950 kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define LIDAR_OK 0
#define LIDAR_ERR_ARG (-1)
#define LIDAR_ERR_LAUNCH (-2)
#define LIDAR_ERR_WORKSPACE (-3)
#define LIDAR_ERR_UNSUPPORTED (-4)   // an optional library path is not available here: the caller keeps its other path

#define LIDAR_EXPORT extern "C" __attribute__((visibility("default")))

static inline int lidar_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fprintf(stderr, "[lidar_hip] %s: %s\n", what, hipGetErrorString(e));
        return LIDAR_ERR_LAUNCH;
    }
    return LIDAR_OK;
}

static inline int divup(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}

// inclusive wave scan (sum) of an int over 64 lanes
__device__ __forceinline__ int wave_incl_scan(int v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (l >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

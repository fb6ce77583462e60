// Row order for the mask-sorted sparse GEMM (sparse_conv.hip): argsort of the per-row neighbour-offset bit masks.
// torch.argsort sends these sizes (150-430 k keys) to rocPRIM's block sort + 11 merge passes = ~30 launches per table, which
// made the sparse backbone's rulebook phase launch-bound (150 of the 220 launches of one SECOND forward).  Here rocPRIM's
// onesweep radix sort is selected explicitly (merge-sort limit 0) over the K significant bits only: histogram + scan +
// ceil(K / 8) passes = 6 launches for K = 27.  Any permutation of equal keys gives bit-identical GEMM results (every output
// row is summed by its own lane in fixed offset order), so nothing depends on the sort's tie order; it is stable anyway.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "common.h"

using ScSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

static size_t ms_align(size_t b) { return (b + 255) & ~(size_t)255; }

static hipError_t ms_sort(void *tmp, size_t &tmp_bytes, const unsigned *keys, unsigned *keys_sorted, int *order, int n, int bits,
                          hipStream_t s) {
    return rocprim::radix_sort_pairs<ScSortConfig>(tmp, tmp_bytes, keys, keys_sorted, rocprim::counting_iterator<int>(0), order,
                                                   (size_t)n, 0u, (unsigned)bits, s);
}

LIDAR_EXPORT size_t lidar_spconv_mask_order_workspace_bytes(int n_out, int K) {
    if (n_out <= 0 || K <= 0 || K > 32) return 0;
    size_t tmp = 0;
    if (ms_sort(nullptr, tmp, nullptr, nullptr, nullptr, n_out, K, nullptr) != hipSuccess) return 0;
    return ms_align((size_t)n_out * sizeof(unsigned)) + ms_align(tmp) + 256;
}

// masks (n_out) in table order -> order (n_out): order[i] = table row visited i-th (masks ascending as K-bit unsigned numbers)
LIDAR_EXPORT int lidar_spconv_mask_order(const int *masks, int n_out, int K, int *order, void *ws, size_t ws_bytes, void *stream) {
    if (n_out < 0 || K <= 0 || K > 32) return LIDAR_ERR_ARG;
    if (n_out == 0) return LIDAR_OK;
    if (!masks || !order || !ws) return LIDAR_ERR_ARG;
    if (ws_bytes < lidar_spconv_mask_order_workspace_bytes(n_out, K)) return LIDAR_ERR_WORKSPACE;
    unsigned *sorted = (unsigned *)ws;
    void *tmp = (char *)ws + ms_align((size_t)n_out * sizeof(unsigned));
    size_t tmp_bytes = 0;
    if (ms_sort(nullptr, tmp_bytes, nullptr, nullptr, nullptr, n_out, K, nullptr) != hipSuccess) return LIDAR_ERR_LAUNCH;
    if (ms_sort(tmp, tmp_bytes, (const unsigned *)masks, sorted, order, n_out, K, (hipStream_t)stream) != hipSuccess)
        return LIDAR_ERR_LAUNCH;
    return lidar_check_launch("lidar_spconv_mask_order");
}

/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into or called from the product path.
 *
 * Sequential restatement of spconv v1.2 `VoxelGeneratorV2.generate` (points_to_voxel_3d_np), the
 * voxeliser the reference calls at
 *   /root/reference/pcdet/datasets/processor/data_processor.py:48-80
 * spconv is an un-vendored third-party dependency (pinned by the reference at v1.0@8da6f96 / v1.2,
 * docs/INSTALL.md:9,28-29; setup.py:41) and is absent from /root/reference and from this image, so
 * this file restates its published algorithm.  PARITY UNPINNED: the
 * reference holds no test / golden vector for this boundary.  Self-consistency properties are
 * checked in tests/test_oracle_pins.py instead.
 *
 * Semantics (v1.2): strictly sequential over points; coordinate c_j = floor((p_j - lo_j)/vs_j) in
 * fp32; a point outside the grid in any of x,y,z is skipped; voxel ids are handed out in order of
 * first appearance; once `max_voxels` exist, points that would open a new voxel are skipped
 * (`continue`, not `break`); the first `max_points` points of each voxel are stored, later ones are
 * dropped.  Output coordinate order is (z, y, x).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* coor_to_voxelidx: caller-provided int32 map of nz*ny*nx cells, all -1 on entry; restored on exit.
 * voxels (max_voxels, max_points, C) must be zero on entry.  Returns voxel_num. */
int orc_voxelize(const float *points, int n, int c,
                 const float *range6, const float *vsize3, const int *grid3 /* nx,ny,nz */,
                 int max_points, int max_voxels,
                 float *voxels, int32_t *coors /* (max_voxels,3) z,y,x */, int32_t *num_per_voxel,
                 int32_t *coor_to_voxelidx) {
    int voxel_num = 0;
    const int nx = grid3[0], ny = grid3[1];
    for (int i = 0; i < n; i++) {
        int coor[3]; /* z,y,x */
        int failed = 0;
        for (int j = 0; j < 3; j++) {
            float q = (points[(size_t)i * c + j] - range6[j]) / vsize3[j];
            int cc = (int)floorf(q);
            if (cc < 0 || cc >= grid3[j]) { failed = 1; break; }
            coor[2 - j] = cc;
        }
        if (failed) continue;
        size_t cell = ((size_t)coor[0] * ny + coor[1]) * nx + coor[2];
        int vid = coor_to_voxelidx[cell];
        if (vid == -1) {
            if (voxel_num >= max_voxels) continue;
            vid = voxel_num++;
            coor_to_voxelidx[cell] = vid;
            coors[3 * vid + 0] = coor[0]; coors[3 * vid + 1] = coor[1]; coors[3 * vid + 2] = coor[2];
        }
        int k = num_per_voxel[vid];
        if (k < max_points) {
            memcpy(voxels + ((size_t)vid * max_points + k) * c, points + (size_t)i * c, sizeof(float) * c);
            num_per_voxel[vid] = k + 1;
        }
    }
    for (int v = 0; v < voxel_num; v++) {
        size_t cell = ((size_t)coors[3 * v] * ny + coors[3 * v + 1]) * nx + coors[3 * v + 2];
        coor_to_voxelidx[cell] = -1;
    }
    return voxel_num;
}

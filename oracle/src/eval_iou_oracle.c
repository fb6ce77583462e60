/* TEST INFRASTRUCTURE ONLY.  Literal C restatement of the reference's numba-CUDA KITTI-eval rotated IoU
 * (pcdet/datasets/kitti/kitti_object_eval_python/rotate_iou.py:17-330), following numba's typing rules: float32 arrays,
 * float32 (x) float32 -> float32, anything mixed with a Python int / float literal -> float64 (then cast back when stored
 * into a float32 array).  PARITY UNPINNED: numba is not installed in this image, so the reference kernel cannot be run;
 * NVVM's fused multiply-add contraction is not modelled (no FMA here). */
#include <math.h>
#include <stdlib.h>

static double trangle_area(const float *a, const float *b, const float *c) { /* :17-20 */
    return (double)((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / 2.0;
}

static double area(const float *int_pts, int num_of_inter) { /* :23-30 */
    double area_val = 0.0;
    for (int i = 0; i < num_of_inter - 2; ++i)
        area_val += fabs(trangle_area(int_pts, int_pts + 2 * i + 2, int_pts + 2 * i + 4));
    return area_val;
}

static void sort_vertex_in_convex_polygon(float *int_pts, int num_of_inter) { /* :33-69 */
    if (num_of_inter <= 0) return;
    float center[2] = {0.0f, 0.0f};
    for (int i = 0; i < num_of_inter; ++i) {
        center[0] += int_pts[2 * i];
        center[1] += int_pts[2 * i + 1];
    }
    center[0] = (float)((double)center[0] / (double)num_of_inter);
    center[1] = (float)((double)center[1] / (double)num_of_inter);
    float v[2], vs[16];
    for (int i = 0; i < num_of_inter; ++i) {
        v[0] = int_pts[2 * i] - center[0];
        v[1] = int_pts[2 * i + 1] - center[1];
        float d = sqrtf(v[0] * v[0] + v[1] * v[1]);
        v[0] = v[0] / d;
        v[1] = v[1] / d;
        if (v[1] < 0) v[0] = (float)(-2.0 - (double)v[0]);
        vs[i] = v[0];
    }
    for (int i = 1; i < num_of_inter; ++i) {
        if (vs[i - 1] > vs[i]) {
            float temp = vs[i], tx = int_pts[2 * i], ty = int_pts[2 * i + 1];
            int j = i;
            while (j > 0 && vs[j - 1] > temp) {
                vs[j] = vs[j - 1];
                int_pts[j * 2] = int_pts[j * 2 - 2];
                int_pts[j * 2 + 1] = int_pts[j * 2 - 1];
                --j;
            }
            vs[j] = temp;
            int_pts[j * 2] = tx;
            int_pts[j * 2 + 1] = ty;
        }
    }
}

static int line_segment_intersection(const float *pts1, const float *pts2, int i, int j, float *temp_pts) { /* :72-116 */
    float A[2] = {pts1[2 * i], pts1[2 * i + 1]};
    float B[2] = {pts1[2 * ((i + 1) % 4)], pts1[2 * ((i + 1) % 4) + 1]};
    float C[2] = {pts2[2 * j], pts2[2 * j + 1]};
    float D[2] = {pts2[2 * ((j + 1) % 4)], pts2[2 * ((j + 1) % 4) + 1]};
    float BA0 = B[0] - A[0], BA1 = B[1] - A[1], DA0 = D[0] - A[0], CA0 = C[0] - A[0], DA1 = D[1] - A[1], CA1 = C[1] - A[1];
    int acd = DA1 * CA0 > CA1 * DA0;
    int bcd = (D[1] - B[1]) * (C[0] - B[0]) > (C[1] - B[1]) * (D[0] - B[0]);
    if (acd != bcd) {
        int abc = CA1 * BA0 > BA1 * CA0;
        int abd = DA1 * BA0 > BA1 * DA0;
        if (abc != abd) {
            float DC0 = D[0] - C[0], DC1 = D[1] - C[1];
            float ABBA = A[0] * B[1] - B[0] * A[1];
            float CDDC = C[0] * D[1] - D[0] * C[1];
            float DH = BA1 * DC0 - BA0 * DC1;
            float Dx = ABBA * DC0 - BA0 * CDDC;
            float Dy = ABBA * DC1 - BA1 * CDDC;
            temp_pts[0] = Dx / DH;
            temp_pts[1] = Dy / DH;
            return 1;
        }
    }
    return 0;
}

static int point_in_quadrilateral(float pt_x, float pt_y, const float *corners) { /* :157-173 */
    float ab0 = corners[2] - corners[0], ab1 = corners[3] - corners[1];
    float ad0 = corners[6] - corners[0], ad1 = corners[7] - corners[1];
    float ap0 = pt_x - corners[0], ap1 = pt_y - corners[1];
    float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
    float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

static int quadrilateral_intersection(const float *pts1, const float *pts2, float *int_pts) { /* :176-197 */
    int num_of_inter = 0;
    for (int i = 0; i < 4; ++i) {
        if (num_of_inter < 16 && point_in_quadrilateral(pts1[2 * i], pts1[2 * i + 1], pts2)) {
            int_pts[num_of_inter * 2] = pts1[2 * i];
            int_pts[num_of_inter * 2 + 1] = pts1[2 * i + 1];
            ++num_of_inter;
        }
        if (num_of_inter < 16 && point_in_quadrilateral(pts2[2 * i], pts2[2 * i + 1], pts1)) {
            int_pts[num_of_inter * 2] = pts2[2 * i];
            int_pts[num_of_inter * 2 + 1] = pts2[2 * i + 1];
            ++num_of_inter;
        }
    }
    float temp_pts[2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (num_of_inter < 16 && line_segment_intersection(pts1, pts2, i, j, temp_pts)) {
                int_pts[num_of_inter * 2] = temp_pts[0];
                int_pts[num_of_inter * 2 + 1] = temp_pts[1];
                ++num_of_inter;
            }
    return num_of_inter;
}

static void rbbox_to_corners(float *corners, const float *rbbox) { /* :200-223 */
    float angle = rbbox[4];
    float a_cos = cosf(angle), a_sin = sinf(angle);
    float center_x = rbbox[0], center_y = rbbox[1], x_d = rbbox[2], y_d = rbbox[3];
    float corners_x[4], corners_y[4];
    corners_x[0] = (float)(-(double)x_d / 2); corners_x[1] = (float)(-(double)x_d / 2);
    corners_x[2] = (float)((double)x_d / 2);  corners_x[3] = (float)((double)x_d / 2);
    corners_y[0] = (float)(-(double)y_d / 2); corners_y[1] = (float)((double)y_d / 2);
    corners_y[2] = (float)((double)y_d / 2);  corners_y[3] = (float)(-(double)y_d / 2);
    for (int i = 0; i < 4; ++i) {
        corners[2 * i] = a_cos * corners_x[i] + a_sin * corners_y[i] + center_x;
        corners[2 * i + 1] = -a_sin * corners_x[i] + a_cos * corners_y[i] + center_y;
    }
}

static double inter(const float *rbbox1, const float *rbbox2) { /* :226-239 */
    /* the reference's local buffer holds 8 points and is not bounds-checked; coincident boxes can produce more.  Both this
     * restatement and the HIP kernel keep up to 16 points (vs[] has 16 entries in the reference too) and drop the rest. */
    float corners1[8], corners2[8], intersection_corners[32];
    rbbox_to_corners(corners1, rbbox1);
    rbbox_to_corners(corners2, rbbox2);
    int n = quadrilateral_intersection(corners1, corners2, intersection_corners);
    sort_vertex_in_convex_polygon(intersection_corners, n);
    return area(intersection_corners, n);
}

/* devRotateIoUEval (:242-254) called as in rotate_iou_kernel_eval (:256-288): rbox1 = query box, rbox2 = box */
void orc_rotate_iou_eval(const float *boxes, int N, const float *query_boxes, int K, int criterion, float *iou) {
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const float *rbox1 = query_boxes + 5 * k, *rbox2 = boxes + 5 * n;
            float area1 = rbox1[2] * rbox1[3], area2 = rbox2[2] * rbox2[3];
            double area_inter = inter(rbox1, rbox2);
            double r;
            if (criterion == -1) r = area_inter / ((double)area1 + (double)area2 - area_inter);
            else if (criterion == 0) r = area_inter / (double)area1;
            else if (criterion == 1) r = area_inter / (double)area2;
            else r = area_inter;
            iou[(size_t)n * K + k] = (float)r;
        }
}

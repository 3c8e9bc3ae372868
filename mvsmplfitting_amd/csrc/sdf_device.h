// Per-voxel pieces of the SDF voxelisation (reference sdf/sdf/csrc/sdf_cuda_kernel.cu:73-237), shared by the
// stand-alone op (sdf_voxelize.hip) and the interpenetration term of the loss (sdf_term.hip), which evaluates
// the same voxel function on the fly at the 8 grid corners each vertex samples.  Arithmetic is kept
// un-contracted (no FMA fusion): it is the expression tree of the restatement in oracle/sdf_np.py, and both
// users produce bit-identical voxel values.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvfit {

constexpr int SDF_NT = 256;
constexpr int SDF_CH = 128;

struct SdfTri {          // 32 floats = 128 B per triangle
    float v1[3], v2[3], v3[3];
    float x13[3], x23[3];
    float m13, m23, d, invdet;
    float e1[3], e2[3];          // v2 - v1, v3 - v1 (ray test: vert0 = v1)
    float d12[3], m12;           // segment invariants: x2 - x1 and |.|^2 per edge are recomputed (cheap)
    float pad[3];
};

#pragma clang fp contract(off)

__device__ __forceinline__ float sdf_dot3(const float* a, const float* b) {
    float l = 0.f;
    l += a[0] * b[0]; l += a[1] * b[1]; l += a[2] * b[2];
    return l;
}
__device__ __forceinline__ float sdf_dist(const float* x, const float* y) {
    float l = 0.f, df;
    df = x[0] - y[0]; l += df * df;
    df = x[1] - y[1]; l += df * df;
    df = x[2] - y[2]; l += df * df;
    return sqrtf(l);
}
// sdf_cuda_kernel.cu:73-92
__device__ __forceinline__ float sdf_point_segment(const float* x0, const float* x1, const float* x2, float* r) {
    const float dx[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
    const float m2 = sdf_dot3(dx, dx);
    float s12 = (sdf_dot3(x2, dx) - sdf_dot3(x0, dx)) / m2;
    if (s12 < 0.f) s12 = 0.f; else if (s12 > 1.f) s12 = 1.f;
    for (int i = 0; i < 3; ++i) r[i] = s12 * x1[i] + (1.f - s12) * x2[i];
    return sdf_dist(x0, r);
}

// per-triangle invariants of both tests from the three (normalised) vertex positions
__device__ __forceinline__ void sdf_tri_setup(SdfTri& T, const float* p1, const float* p2, const float* p3) {
    for (int q = 0; q < 3; ++q) { T.v1[q] = p1[q]; T.v2[q] = p2[q]; T.v3[q] = p3[q]; }
    for (int q = 0; q < 3; ++q) { T.x13[q] = T.v1[q] - T.v3[q]; T.x23[q] = T.v2[q] - T.v3[q]; }
    T.m13 = sdf_dot3(T.x13, T.x13);
    T.m23 = sdf_dot3(T.x23, T.x23);
    T.d = sdf_dot3(T.x13, T.x23);
    T.invdet = 1.f / fmaxf(T.m13 * T.m23 - T.d * T.d, 1e-30f);
    for (int q = 0; q < 3; ++q) { T.e1[q] = T.v2[q] - T.v1[q]; T.e2[q] = T.v3[q] - T.v1[q]; }
}

// voxel centre coordinate: "-1 + (idx + 0.5) * dx" with dx = 2/(G-1) rounded to float first (:252-256)
__device__ __forceinline__ float sdf_voxel_coord(int idx, int G) {
    const float dx = (float)(2.0 / (G - 1));
    return (float)(-1 + (idx + 0.5) * (double)dx);
}

// point_triangle_distance (:155-237) followed by the distance recomputed from the closest point (:278)
__device__ __forceinline__ float sdf_tri_distance(const SdfTri& T, const float* c) {
    const float x03[3] = {c[0] - T.v3[0], c[1] - T.v3[1], c[2] - T.v3[2]};
    const float a = sdf_dot3(T.x13, x03), b = sdf_dot3(T.x23, x03);
    const float w23 = T.invdet * (T.m23 * a - T.d * b);
    const float w31 = T.invdet * (T.m13 * b - T.d * a);
    const float w12 = 1.f - w23 - w31;
    float r[3];
    if (w23 >= 0.f && w31 >= 0.f && w12 >= 0.f) {
        for (int q = 0; q < 3; ++q) r[q] = w23 * T.v1[q] + w31 * T.v2[q] + w12 * T.v3[q];
    } else {
        float r1[3], r2[3], d1, d2;
        if (w23 > 0.f) { d1 = sdf_point_segment(c, T.v1, T.v2, r1); d2 = sdf_point_segment(c, T.v1, T.v3, r2); }
        else if (w31 > 0.f) { d1 = sdf_point_segment(c, T.v1, T.v2, r1); d2 = sdf_point_segment(c, T.v2, T.v3, r2); }
        else { d1 = sdf_point_segment(c, T.v1, T.v3, r1); d2 = sdf_point_segment(c, T.v2, T.v3, r2); }
        const bool first = d1 < d2;
        for (int q = 0; q < 3; ++q) r[q] = first ? r1[q] : r2[q];
    }
    return sdf_dist(c, r);
}

// triangle_ray_intersection / intersect_triangle (:95-150): segment from c towards (-1,-1,-1), t >= 0, unbounded
__device__ __forceinline__ bool sdf_ray_hit(const SdfTri& T, const float* c) {
    const float dir[3] = {-1.0f - c[0], -1.0f - c[1], -1.0f - c[2]};
    const float pvec[3] = {dir[1] * T.e2[2] - dir[2] * T.e2[1], dir[2] * T.e2[0] - dir[0] * T.e2[2],
                           dir[0] * T.e2[1] - dir[1] * T.e2[0]};
    const float det = T.e1[0] * pvec[0] + T.e1[1] * pvec[1] + T.e1[2] * pvec[2];
    if (det > -0.000001 && det < 0.000001) return false;
    const float inv_det = (float)(1.0 / (double)det);
    const float tvec[3] = {c[0] - T.v1[0], c[1] - T.v1[1], c[2] - T.v1[2]};
    const float u = (tvec[0] * pvec[0] + tvec[1] * pvec[1] + tvec[2] * pvec[2]) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    const float qvec[3] = {tvec[1] * T.e1[2] - tvec[2] * T.e1[1], tvec[2] * T.e1[0] - tvec[0] * T.e1[2],
                           tvec[0] * T.e1[1] - tvec[1] * T.e1[0]};
    const float v = (dir[0] * qvec[0] + dir[1] * qvec[1] + dir[2] * qvec[2]) * inv_det;
    if (v < 0.0f || (u + v) > 1.0f) return false;
    const float tt = (T.e2[0] * qvec[0] + T.e2[1] * qvec[1] + T.e2[2] * qvec[2]) * inv_det;
    return tt >= 0.f;
}

}  // namespace mvfit

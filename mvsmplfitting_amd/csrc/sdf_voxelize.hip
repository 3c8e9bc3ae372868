// SDF voxelisation op: phi[b, k, j, i] = distance from the voxel centre to the nearest triangle if the
// centre is inside the mesh (odd number of crossings of the segment towards (-1,-1,-1)), else 0.
//
// Replaces the reference's only native hot-path kernel, sdf/sdf/csrc/sdf_cuda_kernel.cu:242-335
// (+ binding sdf_cuda.cpp:14-28, module sdf/sdf/sdf.py:21-26): same voxel-centre convention
// (-1 + (idx + 0.5) * 2/(G-1), i fastest = x), same closest-point construction (:155-237), same
// Moeller-Trumbore test (eps 1e-6, t >= 0, unbounded; :95-150), min_distance start 1000 (:268).
// Deviation, documented: every voxel is computed (the reference launches G^3 / 512 blocks with integer
// division, :317, and leaves the tail at zero when G^3 is not a multiple of 512).
//
// MI355X mapping: brute force is an all-pairs (voxel x triangle) kernel - compute bound.  One thread per
// voxel (256 per workgroup, consecutive i -> coalesced phi stores); triangles are processed in chunks
// of 128: the workgroup gathers the chunk's vertices once, precomputes the per-triangle invariants of
// both tests (edge vectors, Gram terms, 1/det) and keeps them in LDS, where every lane reads the same
// triangle at the same time (broadcast reads, no bank conflicts).  Per voxel the inner loop is then
// ~60 FMA-class ops per triangle.  Arithmetic is kept un-contracted (no FMA fusion) so that it is
// the expression tree of the restatement in oracle/sdf_np.py.
#include "sdf_device.h"

namespace mvfit {

#pragma clang fp contract(off)

__global__ __launch_bounds__(SDF_NT) void sdf_voxelize_kernel(const int32_t* __restrict__ faces, int num_faces,
                                                              const float* __restrict__ vertices, int num_vertices,
                                                              int G, float* __restrict__ phi) {
    __shared__ SdfTri tri[SDF_CH];
    const int bn = blockIdx.y;
    const int nvox = G * G * G;
    const int vid = blockIdx.x * SDF_NT + threadIdx.x;
    const bool live = vid < nvox;
    const int i = vid % G, j = (vid / G) % G, k = (vid / (G * G)) % G;
    const float c[3] = {sdf_voxel_coord(i, G), sdf_voxel_coord(j, G), sdf_voxel_coord(k, G)};
    const float* vb = vertices + (size_t)bn * num_vertices * 3;
    // two passes over the face list (the reference takes both quantities from one loop, :258-287): the crossing parity of
    // every voxel first, then the minimum distance only where a voxel of the workgroup is inside - phi is 0 elsewhere, so
    // a workgroup of 256 consecutive voxels that lies outside the mesh (most of the grid) skips the second walk
    int num_intersect = 0;
    for (int f0 = 0; f0 < num_faces; f0 += SDF_CH) {
        const int nf = min(SDF_CH, num_faces - f0);
        __syncthreads();
        for (int t = threadIdx.x; t < nf; t += SDF_NT) {
            const int a = faces[3 * (f0 + t)], b = faces[3 * (f0 + t) + 1], cc = faces[3 * (f0 + t) + 2];
            sdf_tri_setup(tri[t], vb + 3 * a, vb + 3 * b, vb + 3 * cc);
        }
        __syncthreads();
        if (!live) continue;
        for (int t = 0; t < nf; ++t)
            if (sdf_ray_hit(tri[t], c)) num_intersect++;
    }
    const bool inside = live && (num_intersect % 2 != 0);
    float min_distance = 1000.f;
    if (__syncthreads_or(inside ? 1 : 0)) {
        for (int f0 = 0; f0 < num_faces; f0 += SDF_CH) {
            const int nf = min(SDF_CH, num_faces - f0);
            if (num_faces > SDF_CH) {                    // a single chunk is still staged from the first pass
                __syncthreads();
                for (int t = threadIdx.x; t < nf; t += SDF_NT) {
                    const int a = faces[3 * (f0 + t)], b = faces[3 * (f0 + t) + 1], cc = faces[3 * (f0 + t) + 2];
                    sdf_tri_setup(tri[t], vb + 3 * a, vb + 3 * b, vb + 3 * cc);
                }
                __syncthreads();
            }
            if (!inside) continue;
            for (int t = 0; t < nf; ++t) {
                const float distance = sdf_tri_distance(tri[t], c);
                if (distance < min_distance) min_distance = distance;
            }
        }
    }
    if (live) phi[(size_t)bn * nvox + vid] = inside ? min_distance : 0.f;
}

hipError_t launch_sdf_voxelize(const int32_t* faces, int num_faces, const float* vertices, int B, int num_vertices, int G,
                               float* phi, hipStream_t stream) {
    const int nvox = G * G * G;
    dim3 grid((nvox + SDF_NT - 1) / SDF_NT, B);
    hipLaunchKernelGGL(sdf_voxelize_kernel, grid, dim3(SDF_NT), 0, stream, faces, num_faces, vertices, num_vertices, G, phi);
    return hipGetLastError();
}

}  // namespace mvfit

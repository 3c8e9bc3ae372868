// Per-view pinhole projection of point sets (the full mesh, or joints): the reference's visualisation path
// `cam(verts)` per view (code/utils/utils.py:603-607 visualize_fitting, :581-583 project_to_img) with
// PerspectiveCamera.forward (code/camera.py:93-117): p = R X + t, uv = f * p_xy / p_z + c (fx == fy, code/init.py:113-119).
// This is the literal "subject x view x 6890 x 3" transform of the task statement; it is not part of the objective
// (SURVEY fact 3, section 8(f) row 4).
//
// HBM-bound and trivially so: per point 12 bytes in, 8 bytes out per view.  One thread per (problem, point): the point
// is read once, the V cameras sit in LDS (broadcast reads), the V results go out as coalesced 8-byte stores per view
// plane ([B][V][N][2]: consecutive threads = consecutive points of one view).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mvfit_device.h"

namespace mvfit {

constexpr int PJ_NT = 256;

__global__ __launch_bounds__(PJ_NT) void project_points_kernel(DevProblems Q, const float* __restrict__ pts, int N,
                                                               float* __restrict__ uv) {
    __shared__ float cam[MVFIT_MAX_VIEWS][16];          // R (9), t (3), f, cx, cy
    const int b = blockIdx.y, V = Q.V;
    const size_t cb = Q.cam_batched ? (size_t)b * V : 0;
    for (int i = threadIdx.x; i < V * 16; i += PJ_NT) {
        const int v = i >> 4, e = i & 15;
        float x = 0.f;
        if (e < 9) x = Q.cam_R[(cb + v) * 9 + e];
        else if (e < 12) x = Q.cam_t[(cb + v) * 3 + e - 9];
        else if (e == 12) x = Q.cam_f[cb + v];
        else if (e < 15) x = Q.cam_c[(cb + v) * 2 + e - 13];
        cam[v][e] = x;
    }
    __syncthreads();
    const int n = blockIdx.x * PJ_NT + threadIdx.x;
    if (n >= N) return;
    const float* p = pts + ((size_t)b * N + n) * 3;
    const float X = p[0], Y = p[1], Z = p[2];
    for (int v = 0; v < V; ++v) {
        const float* c = cam[v];
        const float px = c[0] * X + c[1] * Y + c[2] * Z + c[9];
        const float py = c[3] * X + c[4] * Y + c[5] * Z + c[10];
        const float pz = c[6] * X + c[7] * Y + c[8] * Z + c[11];
        float2 o;
        o.x = c[12] * (px / pz) + c[13];            // camera.py:112-116: divide, then scale by the focal length, then centre
        o.y = c[12] * (py / pz) + c[14];
        *reinterpret_cast<float2*>(uv + (((size_t)b * V + v) * N + n) * 2) = o;
    }
}

hipError_t launch_project_points(const DevProblems& Q, const float* pts, int N, float* uv, hipStream_t stream) {
    dim3 grid((N + PJ_NT - 1) / PJ_NT, Q.B);
    hipLaunchKernelGGL(project_points_kernel, grid, dim3(PJ_NT), 0, stream, Q, pts, N, uv);
    return hipGetLastError();
}

}  // namespace mvfit

// Multi-view linear triangulation of the 17 keypoints of a frame: the first stage of the reference's per-frame
// initial guess (code/utils/init_guess.py:80-83 -> code/utils/recompute3D.py:22-62), batched
// over frames.  One thread per (frame, joint): the per-view accumulation of the 3x3 normal equations in float64,
// AtA rounded to float32 before the solve exactly like the reference (:54), Gaussian elimination with partial
// pivoting in float64 (np.linalg.solve = LAPACK gesv).  Flop-trivial and latency-trivial; it is on the device so
// that a batched pipeline has no CPU stage between the keypoint tensors and the fit.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvfit {

__device__ __forceinline__ void inv3(const double* K, double* Ki) {          // np.linalg.inv of a 3x3
    const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C;
    const double id = 1.0 / det;
    Ki[0] = A * id;  Ki[1] = -(b * i - c * h) * id; Ki[2] = (b * f - c * e) * id;
    Ki[3] = B * id;  Ki[4] = (a * i - c * g) * id;  Ki[5] = -(a * f - c * d) * id;
    Ki[6] = C * id;  Ki[7] = -(a * h - b * g) * id; Ki[8] = (a * e - b * d) * id;
}

__global__ void triangulate_kernel(const float* __restrict__ kps, const double* __restrict__ intris,
                                   const double* __restrict__ extris, int B, int V, int J, double* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * J) return;
    const int b = idx / J, j = idx - b * J;
    double AtA[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Atb[3] = {0, 0, 0};
    for (int v = 0; v < V; ++v) {
        double Ki[9];
        inv3(intris + 9 * v, Ki);
        const double* E = extris + 16 * v;
        const double R[9] = {E[0], E[1], E[2], E[4], E[5], E[6], E[8], E[9], E[10]};
        const double t[3] = {E[3], E[7], E[11]};
        const float* kp = kps + (((size_t)b * V + v) * J + j) * 3;
        const double x = (double)kp[0], y = (double)kp[1], w = (double)kp[2] + 1e-6;       // :47-48 (conf + 1e-6)
        double n[3] = {Ki[0] * x + Ki[1] * y + Ki[2], Ki[3] * x + Ki[4] * y + Ki[5], Ki[6] * x + Ki[7] * y + Ki[8]};
        const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        n[0] /= nn; n[1] /= nn; n[2] /= nn;
        double P[9];                                                                       // I - n n^T  (:17-20)
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) P[3 * r + c] = (r == c ? 1.0 : 0.0) - n[r] * n[c];
        double N[9];                                                                       // R^T P      (:46)
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
            N[3 * r + c] = R[0 + r] * P[0 + c] + R[3 + r] * P[3 + c] + R[6 + r] * P[6 + c];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c)
                AtA[3 * r + c] += (N[3 * r] * R[c] + N[3 * r + 1] * R[3 + c] + N[3 * r + 2] * R[6 + c]) * w;
            Atb[r] += -(N[3 * r] * t[0] + N[3 * r + 1] * t[1] + N[3 * r + 2] * t[2]) * w;
        }
    }
    double M[3][4];
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) M[r][c] = (double)(float)AtA[3 * r + c]; M[r][3] = Atb[r]; }   // :54
    // Gaussian elimination with partial pivoting (np.linalg.solve)
    for (int c = 0; c < 3; ++c) {
        int p = c;
        for (int r = c + 1; r < 3; ++r) if (fabs(M[r][c]) > fabs(M[p][c])) p = r;
        if (p != c) for (int q = 0; q < 4; ++q) { const double tmp = M[c][q]; M[c][q] = M[p][q]; M[p][q] = tmp; }
        for (int r = c + 1; r < 3; ++r) {
            const double f = M[r][c] / M[c][c];
            for (int q = c; q < 4; ++q) M[r][q] -= f * M[c][q];
        }
    }
    double xs[3];
    for (int r = 2; r >= 0; --r) {
        double s = M[r][3];
        for (int q = r + 1; q < 3; ++q) s -= M[r][q] * xs[q];
        xs[r] = s / M[r][r];
    }
    out[(size_t)idx * 3 + 0] = xs[0]; out[(size_t)idx * 3 + 1] = xs[1]; out[(size_t)idx * 3 + 2] = xs[2];
}

hipError_t launch_triangulate(const float* kps, const double* intris, const double* extris, int B, int V, int J, double* out,
                              hipStream_t stream) {
    const int n = B * J;
    hipLaunchKernelGGL(triangulate_kernel, dim3((n + 127) / 128), dim3(128), 0, stream, kps, intris, extris, B, V, J, out);
    return hipGetLastError();
}

}  // namespace mvfit

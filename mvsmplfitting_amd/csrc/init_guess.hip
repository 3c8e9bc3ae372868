// Multi-view linear triangulation of the 17 keypoints of a frame: the first stage of the reference's per-frame
// initial guess (code/utils/init_guess.py:80-83 -> code/utils/recompute3D.py:22-62), batched
// over frames.  One thread per (frame, joint): the per-view accumulation of the 3x3 normal equations in float64,
// AtA rounded to float32 before the solve exactly like the reference (:54), Gaussian elimination with partial
// pivoting in float64 (np.linalg.solve = LAPACK gesv).  Flop-trivial and latency-trivial; it is on the device so
// that a batched pipeline has no CPU stage between the keypoint tensors and the fit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lapack_svd3.h"

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

// ---------------------------------------------------------------------------------------------------------
// Single-view branch of the initial guess (code/utils/init_guess.py:54-74): with one camera there is nothing to
// triangulate; the rest-pose keypoints are pushed along the camera's z axis by
//     est_d = fx * (torso height in camera space) / (torso height in the image)
// and mapped back with inv(extri).  The reference's arithmetic is kept as it is: the 3-D height is the mean of the two
// shoulder-hip distances in float64 (:62-63); the 2-D "mean" takes the LEFT shoulder-hip pair twice (:65) over the rows
// (u, v, confidence) of the float32 keypoint array - the confidence difference is inside the norm, and the norm is
// float32 (:66).  One thread per (frame, keypoint); every thread derives the frame's est_d and inv(extri) itself
// (a 4 x 4 Gauss-Jordan with partial pivoting in float64, np.linalg.inv = LAPACK getrf / getri up to rounding).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void inv4(const double* E, double* Ei) {
    double M[4][8];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { M[r][c] = E[4 * r + c]; M[r][4 + c] = r == c ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (fabs(M[r][c]) > fabs(M[p][c])) p = r;
        if (p != c) for (int q = 0; q < 8; ++q) { const double t = M[c][q]; M[c][q] = M[p][q]; M[p][q] = t; }
        const double ip = 1.0 / M[c][c];
        for (int q = 0; q < 8; ++q) M[c][q] *= ip;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = M[r][c];
            for (int q = 0; q < 8; ++q) M[r][q] -= f * M[c][q];
        }
    }
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) Ei[4 * r + c] = M[r][4 + c];
}

__global__ void depth_guess_kernel(const double* __restrict__ rest, const double* __restrict__ extri, const double* __restrict__ intri,
                                   const float* __restrict__ kps, int B, int J, double* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * J) return;
    const int b = idx / J, j = idx - b * J;
    double E[16], Ei[16];
    for (int i = 0; i < 16; ++i) E[i] = extri[i];
    inv4(E, Ei);
    auto cam = [&](int k, double* c4) {                         // extri . (rest_k ; 1), all four rows (:70-71)
        const double x = rest[3 * k], y = rest[3 * k + 1], z = rest[3 * k + 2];
        for (int r = 0; r < 4; ++r) c4[r] = ((E[4 * r] * x + E[4 * r + 1] * y) + E[4 * r + 2] * z) + E[4 * r + 3];
    };
    // torso: L / R shoulder 5, 6 ; L / R hip 11, 12 (:55-60)
    double s5[4], s6[4], h11[4], h12[4];
    cam(5, s5); cam(6, s6); cam(11, h11); cam(12, h12);
    const double dl = sqrt(((s5[0] - h11[0]) * (s5[0] - h11[0]) + (s5[1] - h11[1]) * (s5[1] - h11[1])) + (s5[2] - h11[2]) * (s5[2] - h11[2]));
    const double dr = sqrt(((s6[0] - h12[0]) * (s6[0] - h12[0]) + (s6[1] - h12[1]) * (s6[1] - h12[1])) + (s6[2] - h12[2]) * (s6[2] - h12[2]));
    const double h3 = (dl + dr) / 2.0;                                                     // :63
    const float* k5 = kps + ((size_t)b * J + 5) * 3;
    const float* k11 = kps + ((size_t)b * J + 11) * 3;
    float h2;
    {
#pragma clang fp contract(off)                             // NumPy's float32 expression tree: no fused multiply-adds
        const float d0 = k5[0] - k11[0], d1 = k5[1] - k11[1], d2 = k5[2] - k11[2];         // :65, float32 rows
        const float q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2;
        const float n2 = sqrtf((q0 + q1) + q2);                                             // (IEEE square root: hipcc's default)
        h2 = (n2 + n2) / 2.0f;                                                              // :66: mean of the same value twice
    }
    const double est_d = intri[0] * (h3 / (double)h2);                                      // :68
    double c4[4];
    cam(j, c4);
    c4[2] += est_d;                                                                         // :72
    for (int r = 0; r < 3; ++r)
        out[(size_t)idx * 3 + r] = ((Ei[4 * r] * c4[0] + Ei[4 * r + 1] * c4[1]) + Ei[4 * r + 2] * c4[2]) + Ei[4 * r + 3] * c4[3];
}

hipError_t launch_depth_guess(const double* rest, const double* extri, const double* intri, const float* kps, int B, int J,
                              double* out, hipStream_t stream) {
    const int n = B * J;
    hipLaunchKernelGGL(depth_guess_kernel, dim3((n + 127) / 128), dim3(128), 0, stream, rest, extri, intri, kps, B, J, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Second stage of the initial guess (code/utils/init_guess.py:95-106): similarity alignment of the rest-pose keypoints
// to the triangulated ones - the reference's umeyama (code/utils/umeyama.py:16-109) with its two local changes kept
// as they are (full-rank branch U diag(d) Vh^T, :73; two candidates with the first two rotation columns negated in
// place and the translation taken from the second one, :84-104) - and cv2.Rodrigues of the chosen rotation.
// One thread per frame, float64 like the reference's NumPy.  The reference's full-rank formula is not invariant under
// the sign freedom of the SVD ((u_k, v_k) -> (-u_k, -v_k)): its value is "what LAPACK returned".  The 3x3 SVD here
// therefore walks LAPACK's own path for a matrix of this size (lapack_svd3.h: dgebd2 + dbdsqr + dormbr) and returns
// np.linalg.svd's pairs - rotation and translation equal the reference's (tests/test_umeyama.py, test_gpu_init_guess.py).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// A (row-major 3x3) = U diag(S) Vh, S descending, with the singular-vector signs of np.linalg.svd: lapack_svd3.h
__device__ __forceinline__ void svd3(const double* A, double* U, double* S, double* Vh) { lapack3::svd3(A, U, S, Vh); }

// cv2.Rodrigues, matrix -> rotation vector
__device__ void rotvec3(const double* R, double* rv) {
    double r[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const double s = sqrt((r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * 0.25);
    const double c = fmin(fmax((R[0] + R[4] + R[8] - 1.0) * 0.5, -1.0), 1.0);
    const double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { rv[0] = rv[1] = rv[2] = 0.0; return; }
        double x = sqrt(fmax((R[0] + 1) * 0.5, 0.0));
        double y = sqrt(fmax((R[4] + 1) * 0.5, 0.0)) * (R[1] < 0 ? -1.0 : 1.0);
        double z = sqrt(fmax((R[8] + 1) * 0.5, 0.0)) * (R[2] < 0 ? -1.0 : 1.0);
        if (fabs(x) < fabs(y) && fabs(x) < fabs(z) && ((R[5] > 0) != (y * z > 0))) z = -z;
        const double k = theta / sqrt(x * x + y * y + z * z);
        rv[0] = x * k; rv[1] = y * k; rv[2] = z * k;
        return;
    }
    const double k = 0.5 / s * theta;
    rv[0] = r[0] * k; rv[1] = r[1] * k; rv[2] = r[2] * k;
}

__global__ void umeyama_kernel(const double* __restrict__ src, const double* __restrict__ dst, int B, int npts, int estimate_scale,
                               double* __restrict__ rot_out, double* __restrict__ rvec_out, double* __restrict__ trans_out,
                               double* __restrict__ scale_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* D = dst + (size_t)b * npts * 3;
    double sm[3] = {0, 0, 0}, dm[3] = {0, 0, 0};
    for (int i = 0; i < npts; ++i) for (int a = 0; a < 3; ++a) { sm[a] += src[3 * i + a]; dm[a] += D[3 * i + a]; }
    for (int a = 0; a < 3; ++a) { sm[a] /= npts; dm[a] /= npts; }
    double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, var = 0.0;                   // A = dst_demean^T src_demean / num  (:45)
    for (int i = 0; i < npts; ++i) {
        double sd[3], dd[3];
        for (int a = 0; a < 3; ++a) { sd[a] = src[3 * i + a] - sm[a]; dd[a] = D[3 * i + a] - dm[a]; var += sd[a] * sd[a]; }
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[3 * r + c] += dd[r] * sd[c];
    }
    for (int i = 0; i < 9; ++i) A[i] /= npts;
    var /= npts;                                                            // src_demean.var(axis=0).sum()
    double d[3] = {1.0, 1.0, det3(A) < 0 ? -1.0 : 1.0};                     // :48-50
    double U[9], S[3], Vh[9];
    svd3(A, U, S, Vh);
    const double tol = S[0] * 3.0 * 2.220446049250313e-16;                  // np.linalg.matrix_rank
    const int rank = (S[0] > tol) + (S[1] > tol) + (S[2] > tol);
    double T[9];
    const double nanv = nan("");
    if (rank == 0 || S[0] == 0.0) {
        for (int i = 0; i < 9; ++i) rot_out[(size_t)b * 9 + i] = nanv;
        for (int a = 0; a < 3; ++a) { rvec_out[(size_t)b * 3 + a] = nanv; trans_out[(size_t)b * 3 + a] = nanv; }
        scale_out[b] = nanv;
        return;
    }
    if (rank == 2) {                                                        // :60-68: U diag(.) Vh
        double e[3] = {1.0, 1.0, 1.0};
        if (!(det3(U) * det3(Vh) > 0)) e[2] = -1.0;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
            T[3 * r + c] = U[3 * r] * e[0] * Vh[c] + U[3 * r + 1] * e[1] * Vh[3 + c] + U[3 * r + 2] * e[2] * Vh[6 + c];
    } else {                                                                // :73: U diag(d) V.T with V = numpy's Vh
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
            T[3 * r + c] = U[3 * r] * d[0] * Vh[3 * c] + U[3 * r + 1] * d[1] * Vh[3 * c + 1] + U[3 * r + 2] * d[2] * Vh[3 * c + 2];
    }
    const double scale = estimate_scale ? (S[0] * d[0] + S[1] * d[1] + S[2] * d[2]) / var : 1.0;      // :78-80
    double loss[2], R1[9];
    for (int i = 0; i < 9; ++i) R1[i] = ((i % 3) < 2 ? -T[i] : T[i]);      // second candidate: columns 0, 1 negated (:87)
    for (int cand = 0; cand < 2; ++cand) {
        const double* R = cand ? R1 : T;
        double t[3], acc = 0.0;
        for (int a = 0; a < 3; ++a) t[a] = dm[a] - scale * (R[3 * a] * sm[0] + R[3 * a + 1] * sm[1] + R[3 * a + 2] * sm[2]);
        for (int i = 0; i < npts; ++i)
            for (int a = 0; a < 3; ++a) {
                const double e = scale * (R[3 * a] * src[3 * i] + R[3 * a + 1] * src[3 * i + 1] + R[3 * a + 2] * src[3 * i + 2]) + t[a] - D[3 * i + a];
                acc += e * e;
            }
        loss[cand] = sqrt(acc);
    }
    const double* Rsel = loss[0] > loss[1] ? R1 : T;                       // :105-108
    for (int i = 0; i < 9; ++i) rot_out[(size_t)b * 9 + i] = Rsel[i];
    for (int a = 0; a < 3; ++a)                                            // :104: T holds the second candidate by now
        trans_out[(size_t)b * 3 + a] = dm[a] - scale * (R1[3 * a] * sm[0] + R1[3 * a + 1] * sm[1] + R1[3 * a + 2] * sm[2]);
    scale_out[b] = scale;
    double rv[3];
    rotvec3(Rsel, rv);                                                      // init_guess.py:96
    for (int a = 0; a < 3; ++a) rvec_out[(size_t)b * 3 + a] = rv[a];
}

hipError_t launch_umeyama(const double* src, const double* dst, int B, int npts, int estimate_scale, double* rot, double* rvec,
                          double* trans, double* scale, hipStream_t stream) {
    hipLaunchKernelGGL(umeyama_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, src, dst, B, npts, estimate_scale, rot, rvec, trans, scale);
    return hipGetLastError();
}

}  // namespace mvfit

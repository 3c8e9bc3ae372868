// One closure evaluation for ONE problem, executed by one workgroup with all intermediates in LDS.
//
// Forward = SMPL.forward (reference code/smplx/body_models_scale.py:327-412, code/smplx/lbs.py:135-222)
// restricted to the vertices the objective reads, + SMPLifyLoss.forward
// (code/utils/fitting.py:290-415, camera code/camera.py:93-117, GMoF code/utils/utils.py:427-438,
// priors code/prior.py:53-231, VPoser decoder code/model/VPoser.py:218-232).
// Backward = the hand-derived adjoint that replaces total_loss.backward() (fitting.py:190-192);
// it is the transcription of oracle/closure_np.py:_backward, which matches the reference's
// autograd to 1e-16 in float64.
#pragma once
#include "mvfit_device.h"

namespace mvfit {

struct ClosureLds {
    float x[DPAD];
    float theta[72];
    float beta[12];
    float tau[4];
    float scale;
    float gscale;
    float loss_terms[6];            // data, pose, shape, angle, (coll), total
    int flags_dropped;              // bit0: pose prior dropped, bit1: angle prior dropped
    int gmm_sel;
    int sh_stage, sh_status;        // optimiser scalars broadcast from wave 0 to the block
    float R[NJ][9];
    float ang[NJ];
    float J[NJ][3];
    float Rm[NJ][9];
    float tm[NJ][3];
    float Gr[NJ][9];
    float Gt[NJ][3];
    float A[NJ][12];
    float coef[KROWS];
    float vposed[NC_MAX];
    float xs[NC_MAX];
    float T[NS_MAX][12];
    float kp[NKP][3];
    float gkp_part[MVFIT_MAX_VIEWS][NKP][3];
    float gkp[NKP][3];
    float gx[NC_MAX];
    float gvp[NC_MAX];
    float gAr[NJ][9];
    float gAt[NJ][3];
    float gGr[NJ][9];
    float gGt[NJ][3];
    float gJ[NJ][3];
    float gRm[NJ][9];
    float gtm[NJ][3];
    float gR[NJ][9];
    float gcoef[KROWS];
    float gtheta[72];
    float gbeta[12];
    float gtau[4];
    float grad[DPAD];
    // VPoser activations (decoder fwd/bwd)
    float vp_pre1[512];
    float vp_pre2[512];
    float vp_h[512];
    float vp_g[512];
    float vp_o[144];
    float vp_go[144];
    float vp_cache[23][32];
    // GMM
    float gmm_t[8][72];
    float gmm_ll[8];
    // scratch for k-split partial sums: max(nks * nc_pad, nks * KROWS)
    float scratch[4096];
    double red[16];
};

__device__ __forceinline__ void mat3_mul(const float* a, const float* b, float* c) {   // c = a b
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            c[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}

// ---------------------------------------------------------------------------------------------
// VPoser decoder forward (VPoser.py:218-232,165-174,263-273,29-156) for one latent, block-wide.
// ---------------------------------------------------------------------------------------------
__device__ void vposer_forward(const DevModel& M, ClosureLds& L, int tid, int nt) {
    // h1 = lrelu(W1 z + b1)
    for (int o = tid; o < 512; o += nt) {
        float s = M.vp_b1[o];
        const float* w = M.vp_w1 + o * 32;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) s = fmaf(w[i], L.x[X_EMB + i], s);
        L.vp_pre1[o] = s;
        L.vp_h[o] = s > 0.f ? s : 0.2f * s;
    }
    __syncthreads();
    // h2 = lrelu(W2 h1 + b2): w2T[i][o] so consecutive threads read consecutive o
    for (int o = tid; o < 512; o += nt) {
        float s = M.vp_b2[o];
#pragma unroll 8
        for (int i = 0; i < 512; ++i) s = fmaf(M.vp_w2T[i * 512 + o], L.vp_h[i], s);
        L.vp_pre2[o] = s;
    }
    __syncthreads();
    for (int o = tid; o < 512; o += nt) { float s = L.vp_pre2[o]; L.vp_h[o] = s > 0.f ? s : 0.2f * s; }
    __syncthreads();
    for (int o = tid; o < 138; o += nt) {
        float s = M.vp_b3[o];
#pragma unroll 8
        for (int i = 0; i < 512; ++i) s = fmaf(M.vp_w3T[i * 144 + o], L.vp_h[i], s);
        L.vp_o[o] = s;
    }
    __syncthreads();
    // per joint: Gram-Schmidt -> R^T rows -> quaternion (4-way branch) -> axis-angle
    for (int j = tid; j < 23; j += nt) {
        float* C = L.vp_cache[j];
        const float* o = L.vp_o + j * 6;                // view(23,3,2): [c][0]=a1, [c][1]=a2
        float a1[3] = {o[0], o[2], o[4]}, a2[3] = {o[1], o[3], o[5]};
        float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
        float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
        float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
        float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
        float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
        float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
        float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
        // m = R^T: rows b1,b2,b3
        const float m00 = b1[0], m01 = b1[1], m02 = b1[2], m10 = b2[0], m11 = b2[1], m12 = b2[2],
                    m20 = b3[0], m21 = b3[1], m22 = b3[2];
        float q[4], t;
        int cs;
        if (m22 < 1e-6f) {
            if (m00 > m11) { cs = 0; t = 1 + m00 - m11 - m22; q[0] = m12 - m21; q[1] = t; q[2] = m01 + m10; q[3] = m20 + m02; }
            else           { cs = 1; t = 1 - m00 + m11 - m22; q[0] = m20 - m02; q[1] = m01 + m10; q[2] = t; q[3] = m12 + m21; }
        } else {
            if (m00 < -m11) { cs = 2; t = 1 - m00 - m11 + m22; q[0] = m01 - m10; q[1] = m20 + m02; q[2] = m12 + m21; q[3] = t; }
            else            { cs = 3; t = 1 + m00 + m11 + m22; q[0] = t; q[1] = m12 - m21; q[2] = m20 - m02; q[3] = m01 - m10; }
        }
        float rs = 0.5f / sqrtf(t);
        float qn[4] = {q[0] * rs, q[1] * rs, q[2] * rs, q[3] * rs};
        float s2 = qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3];
        float s = sqrtf(s2), c = qn[0];
        float tt = 2.0f * (c < 0.f ? atan2f(-s, -c) : atan2f(s, c));
        float kk = s2 > 0.f ? tt / s : 2.0f;
        L.theta[3 + 3 * j + 0] = qn[1] * kk;
        L.theta[3 + 3 * j + 1] = qn[2] * kk;
        L.theta[3 + 3 * j + 2] = qn[3] * kk;
        // cache for the backward
        C[0] = n1; C[1] = b1[0]; C[2] = b1[1]; C[3] = b1[2]; C[4] = d; C[5] = n2;
        C[6] = b2[0]; C[7] = b2[1]; C[8] = b2[2]; C[9] = a2[0]; C[10] = a2[1]; C[11] = a2[2];
        C[12] = (float)cs; C[13] = t; C[14] = q[0]; C[15] = q[1]; C[16] = q[2]; C[17] = q[3];
        C[18] = qn[0]; C[19] = qn[1]; C[20] = qn[2]; C[21] = qn[3]; C[22] = s2; C[23] = tt; C[24] = kk;
    }
    __syncthreads();
}

// g_z += (d body_pose / d z)^T g_body_pose   (transcription of oracle vposer_decode_bwd)
__device__ void vposer_backward(const DevModel& M, ClosureLds& L, int tid, int nt) {
    for (int j = tid; j < 23; j += nt) {
        const float* C = L.vp_cache[j];
        const float n1 = C[0], d = C[4], n2 = C[5];
        const float b1[3] = {C[1], C[2], C[3]}, b2[3] = {C[6], C[7], C[8]}, a2[3] = {C[9], C[10], C[11]};
        const int cs = (int)C[12];
        const float t = C[13];
        const float qraw[4] = {C[14], C[15], C[16], C[17]};
        const float qn[4] = {C[18], C[19], C[20], C[21]};
        const float s2 = C[22], tt = C[23], kk = C[24];
        const float gaa[3] = {L.gtheta[3 + 3 * j], L.gtheta[3 + 3 * j + 1], L.gtheta[3 + 3 * j + 2]};
        float gq[4] = {0.f, gaa[0] * kk, gaa[1] * kk, gaa[2] * kk};
        float gk = gaa[0] * qn[1] + gaa[1] * qn[2] + gaa[2] * qn[3];
        if (s2 > 0.f) {
            float s = sqrtf(s2), c = qn[0];
            float gtt = gk / s;
            float den = s2 + c * c;
            float gs = -gk * tt / s2 + gtt * 2.0f * c / den;
            float gc = gtt * (-2.0f * s / den);
            float f = gs / (2.0f * s) * 2.0f;
            gq[1] += qn[1] * f; gq[2] += qn[2] * f; gq[3] += qn[3] * f;
            gq[0] += gc;
        }
        float rs = 0.5f / sqrtf(t);
        float gqr[4] = {gq[0] * rs, gq[1] * rs, gq[2] * rs, gq[3] * rs};
        float gt = (gq[0] * qraw[0] + gq[1] * qraw[1] + gq[2] * qraw[2] + gq[3] * qraw[3]) * 0.5f * (-0.5f) / (t * sqrtf(t));
        float gm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        if (cs == 0) {
            gt += gqr[1];
            gm[1][2] += gqr[0]; gm[2][1] -= gqr[0];
            gm[0][1] += gqr[2]; gm[1][0] += gqr[2];
            gm[2][0] += gqr[3]; gm[0][2] += gqr[3];
            gm[0][0] += gt; gm[1][1] -= gt; gm[2][2] -= gt;
        } else if (cs == 1) {
            gt += gqr[2];
            gm[2][0] += gqr[0]; gm[0][2] -= gqr[0];
            gm[0][1] += gqr[1]; gm[1][0] += gqr[1];
            gm[1][2] += gqr[3]; gm[2][1] += gqr[3];
            gm[0][0] -= gt; gm[1][1] += gt; gm[2][2] -= gt;
        } else if (cs == 2) {
            gt += gqr[3];
            gm[0][1] += gqr[0]; gm[1][0] -= gqr[0];
            gm[2][0] += gqr[1]; gm[0][2] += gqr[1];
            gm[1][2] += gqr[2]; gm[2][1] += gqr[2];
            gm[0][0] -= gt; gm[1][1] -= gt; gm[2][2] += gt;
        } else {
            gt += gqr[0];
            gm[1][2] += gqr[1]; gm[2][1] -= gqr[1];
            gm[2][0] += gqr[2]; gm[0][2] -= gqr[2];
            gm[0][1] += gqr[3]; gm[1][0] -= gqr[3];
            gm[0][0] += gt; gm[1][1] += gt; gm[2][2] += gt;
        }
        float gb1[3] = {gm[0][0], gm[0][1], gm[0][2]}, gb2[3] = {gm[1][0], gm[1][1], gm[1][2]};
        const float gb3[3] = {gm[2][0], gm[2][1], gm[2][2]};
        // b3 = b1 x b2
        gb1[0] += b2[1] * gb3[2] - b2[2] * gb3[1];
        gb1[1] += b2[2] * gb3[0] - b2[0] * gb3[2];
        gb1[2] += b2[0] * gb3[1] - b2[1] * gb3[0];
        gb2[0] += gb3[1] * b1[2] - gb3[2] * b1[1];
        gb2[1] += gb3[2] * b1[0] - gb3[0] * b1[2];
        gb2[2] += gb3[0] * b1[1] - gb3[1] * b1[0];
        float pb = b2[0] * gb2[0] + b2[1] * gb2[1] + b2[2] * gb2[2];
        float gu[3] = {(gb2[0] - b2[0] * pb) / n2, (gb2[1] - b2[1] * pb) / n2, (gb2[2] - b2[2] * pb) / n2};
        float ga2[3] = {gu[0], gu[1], gu[2]};
        float gd = -(gu[0] * b1[0] + gu[1] * b1[1] + gu[2] * b1[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) { gb1[c] += -d * gu[c] + gd * a2[c]; ga2[c] += gd * b1[c]; }
        float pa = b1[0] * gb1[0] + b1[1] * gb1[1] + b1[2] * gb1[2];
        float* go = L.vp_go + j * 6;
#pragma unroll
        for (int c = 0; c < 3; ++c) { go[2 * c] = (gb1[c] - b1[c] * pa) / n1; go[2 * c + 1] = ga2[c]; }
    }
    __syncthreads();
    // g_h2 = W3^T g_o ; through lrelu
    for (int i = tid; i < 512; i += nt) {
        float s = 0.f;
#pragma unroll 6
        for (int o = 0; o < 138; ++o) s = fmaf(M.vp_w3[o * 512 + i], L.vp_go[o], s);
        L.vp_g[i] = s * (L.vp_pre2[i] > 0.f ? 1.0f : 0.2f);
    }
    __syncthreads();
    // g_h1 = W2^T g_pre2 : w2[o][i], consecutive threads i -> consecutive addresses
    for (int i = tid; i < 512; i += nt) {
        float s = 0.f;
#pragma unroll 8
        for (int o = 0; o < 512; ++o) s = fmaf(M.vp_w2[o * 512 + i], L.vp_g[o], s);
        L.vp_h[i] = s * (L.vp_pre1[i] > 0.f ? 1.0f : 0.2f);
    }
    __syncthreads();
    for (int i = tid; i < 32; i += nt) {
        float s = 0.f;
        for (int o = 0; o < 512; ++o) s = fmaf(M.vp_w1[o * 32 + i], L.vp_h[o], s);
        L.grad[X_EMB + i] += s;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// pose_prep: x -> theta, R, J, kinematic chain, skinning transforms A, blendshape coefficients.
// lbs.py:183-205,269-370 + body_models_scale.py:377
// ---------------------------------------------------------------------------------------------
__device__ void pose_prep(const DevModel& M, ClosureLds& L, uint32_t flags, int tid, int nt) {
    if (flags & MVFIT_F_VPOSER) {
        vposer_forward(M, L, tid, nt);
    } else {
        for (int i = tid; i < 69; i += nt) L.theta[3 + i] = L.x[X_BP + i];
    }
    if (tid < 3) { L.theta[tid] = L.x[X_GO + tid]; L.tau[tid] = L.x[X_TR + tid]; }
    if (tid < 10) L.beta[tid] = L.x[X_BETAS + tid];
    if (tid == 0) L.scale = L.x[X_SC];
    __syncthreads();
    // Rodrigues (lbs.py:269-300): theta=||r+1e-8||, k=r/theta
    if (tid < NJ) {
        const float rx = L.theta[3 * tid], ry = L.theta[3 * tid + 1], rz = L.theta[3 * tid + 2];
        const float ex = rx + 1e-8f, ey = ry + 1e-8f, ez = rz + 1e-8f;
        const float a = sqrtf(ex * ex + ey * ey + ez * ez);
        const float kx = rx / a, ky = ry / a, kz = rz / a;
        float sn, cs;
        sincosf(a, &sn, &cs);
        const float oc = 1.0f - cs;
        // K = [[0,-kz,ky],[kz,0,-kx],[-ky,kx,0]] ; K^2 = k k^T - |k|^2 I
        const float kk = kx * kx + ky * ky + kz * kz;
        float* R = L.R[tid];
        R[0] = 1.f + oc * (kx * kx - kk); R[1] = -sn * kz + oc * kx * ky;   R[2] = sn * ky + oc * kx * kz;
        R[3] = sn * kz + oc * kx * ky;    R[4] = 1.f + oc * (ky * ky - kk); R[5] = -sn * kx + oc * ky * kz;
        R[6] = -sn * ky + oc * kx * kz;   R[7] = sn * kx + oc * ky * kz;    R[8] = 1.f + oc * (kz * kz - kk);
        L.ang[tid] = a;
    }
    // J = J_t + J_S beta   (== J_regressor (v_template + shapedirs beta), lbs.py:179-183)
    for (int i = tid; i < NJ * 3; i += nt) {
        float s = M.J_t[i];
        const float* js = M.J_S + i * 10;
#pragma unroll
        for (int l = 0; l < 10; ++l) s = fmaf(js[l], L.x[X_BETAS + l], s);
        (&L.J[0][0])[i] = s;
    }
    __syncthreads();
    // blendshape coefficients: pose_feature (lbs.py:192), betas, zero pad
    for (int p = tid; p < KROWS; p += nt) {
        float v = 0.f;
        if (p < 207) { int e = p % 9; v = L.R[1 + p / 9][e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f); }
        else if (p < 217) v = L.beta[p - 207];
        L.coef[p] = v;
    }
    // relative transforms (lbs.py:341-348)
    for (int i = tid; i < NJ * 12; i += nt) {
        const int j = i / 12, e = i - j * 12;
        if (e < 9) L.Rm[j][e] = (j == 0 ? L.scale : 1.0f) * L.R[j][e];
        else { const int a = e - 9; const int pa = M.parents[j]; L.tm[j][a] = L.J[j][a] - (j > 0 ? L.J[pa][a] : 0.f); }
    }
    __syncthreads();
    // chain, level by level (lbs.py:349-355)
    for (int i = tid; i < 12; i += nt) { if (i < 9) L.Gr[0][i] = L.Rm[0][i]; else L.Gt[0][i - 9] = L.tm[0][i - 9]; }
    __syncthreads();
    for (int lv = 1; lv < M.nlevels; ++lv) {
        const int n = (M.level_start[lv + 1] - M.level_start[lv]) * 12;
        for (int i = tid; i < n; i += nt) {
            const int j = M.level_joints[M.level_start[lv] + i / 12], e = i % 12;
            const int pa = M.parents[j];
            if (e < 9) {
                const int a = e / 3, b = e % 3;
                L.Gr[j][e] = L.Gr[pa][a * 3] * L.Rm[j][b] + L.Gr[pa][a * 3 + 1] * L.Rm[j][3 + b] + L.Gr[pa][a * 3 + 2] * L.Rm[j][6 + b];
            } else {
                const int a = e - 9;
                L.Gt[j][a] = L.Gr[pa][a * 3] * L.tm[j][0] + L.Gr[pa][a * 3 + 1] * L.tm[j][1] + L.Gr[pa][a * 3 + 2] * L.tm[j][2] + L.Gt[pa][a];
            }
        }
        __syncthreads();
    }
    // A_j = [Gr_j | Gt_j - Gr_j J_j]  (lbs.py:365-368), rows of 4
    for (int i = tid; i < NJ * 12; i += nt) {
        const int j = i / 12, e = i - j * 12, a = e >> 2, c = e & 3;
        float v;
        if (c < 3) v = L.Gr[j][a * 3 + c];
        else v = L.Gt[j][a] - (L.Gr[j][a * 3] * L.J[j][0] + L.Gr[j][a * 3 + 1] * L.J[j][1] + L.Gr[j][a * 3 + 2] * L.J[j][2]);
        L.A[j][e] = v;
    }
    __syncthreads();
}

// write the operands of the vertex pass for problem b
__device__ void publish_pose(const ClosureLds& L, const DevPose& P, int b, int tid, int nt) {
    float* ct = P.coefT + (size_t)(b >> 5) * KROWS * 32 + (b & 31);
    for (int p = tid; p < KROWS; p += nt) ct[p * 32] = L.coef[p];
    for (int i = tid; i < NJ * 12; i += nt) P.Amat[(size_t)b * 288 + i] = (&L.A[0][0])[i];
    if (tid < 3) P.tau[(size_t)b * 4 + tid] = L.tau[tid];
}

// ---------------------------------------------------------------------------------------------
// Objective-relevant vertices: v_posed_s (always needed by the adjoint) and, when the full
// vertex pass did not run (or for the sparse mode), the skinned positions xs.
// ---------------------------------------------------------------------------------------------
__device__ void sparse_forward(const DevModel& M, ClosureLds& L, const float* verts_b, int tid, int nt) {
    const int ncq = M.nc_pad >> 2;                 // float4 column groups
    const int nks = max(1, min(min(nt / ncq, 8), 4096 / M.nc_pad));   // k-slices
    const int rps = (KROWS + nks - 1) / nks;       // rows per slice
    {
        const int cq = tid % ncq, ks = tid / ncq;
        if (ks < nks) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const int r0 = ks * rps, r1 = min(KROWS, r0 + rps);
            const float4* src = reinterpret_cast<const float4*>(M.pd_sub) + cq;
#pragma unroll 8
            for (int p = r0; p < r1; ++p) {
                const float4 v = src[(size_t)p * ncq];
                const float c = L.coef[p];
                acc.x = fmaf(c, v.x, acc.x); acc.y = fmaf(c, v.y, acc.y);
                acc.z = fmaf(c, v.z, acc.z); acc.w = fmaf(c, v.w, acc.w);
            }
            float* dst = L.scratch + ks * M.nc_pad + 4 * cq;
            dst[0] = acc.x; dst[1] = acc.y; dst[2] = acc.z; dst[3] = acc.w;
        }
    }
    // skinning transforms of the selected vertices: T_s = sum_j W[s][j] A_j  (lbs.py:209-213)
    for (int i = tid; i < M.ns * 12; i += nt) {
        const int s = i / 12, e = i - s * 12;
        const float* w = M.w_sub + s * NJ;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc = fmaf(w[j], L.A[j][e], acc);
        L.T[s][e] = acc;
    }
    __syncthreads();
    for (int c = tid; c < M.nc; c += nt) {
        float s = M.vt_sub[c];
        for (int k = 0; k < nks; ++k) s += L.scratch[k * M.nc_pad + c];
        L.vposed[c] = s;
    }
    __syncthreads();
    for (int c = tid; c < M.nc; c += nt) {
        const int s = c / 3, a = c - 3 * s;
        if (verts_b) {
            // full mode: the objective reads the vertex pass output (vertices already hold +transl)
            L.xs[c] = verts_b[(size_t)M.sel_v[s] * 3 + a] - L.tau[a];
        } else {
            const float* T = L.T[s] + 4 * a;
            L.xs[c] = T[0] * L.vposed[3 * s] + T[1] * L.vposed[3 * s + 1] + T[2] * L.vposed[3 * s + 2] + T[3];
        }
    }
    __syncthreads();
    // 17 keypoints = selection rows . xs + transl  (body_models_scale.py:393-403)
    for (int i = tid; i < NKP * 3; i += nt) {
        const int k = i / 3, a = i - 3 * k;
        const float* row = M.ksel_sub + k * NS_MAX;
        float s = 0.f;
        for (int v = 0; v < M.ns; ++v) s = fmaf(row[v], L.xs[3 * v + a], s);
        L.kp[k][a] = s + L.tau[a];
    }
    __syncthreads();
}

// block-wide deterministic sum of one double per thread -> every thread gets the result
__device__ __forceinline__ double block_sum(double v, double* red, int tid, int nt) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
    const int nw = (nt + 63) >> 6;
    for (int w = 0; w < nw; ++w) s += red[w];
    __syncthreads();
    return s;
}

// ---------------------------------------------------------------------------------------------
// SMPLifyLoss.forward (fitting.py:290-415, no SDF term) + gradient w.r.t. keypoints / priors.
// Returns the total loss (same value in every thread).
// ---------------------------------------------------------------------------------------------
__device__ double loss_and_keypoint_grad(const DevModel& M, ClosureLds& L, const DevProblems& Q, int b,
                                         const DevWeights& W, bool want_grad, int tid, int nt) {
    const int V = Q.V;
    const size_t cb = Q.cam_batched ? (size_t)b * V : 0;
    double part = 0.0;
    for (int i = tid; i < V * NKP; i += nt) {
        const int v = i / NKP, k = i - v * NKP;
        const float* Rc = Q.cam_R + (cb + v) * 9;
        const float* tc = Q.cam_t + (cb + v) * 3;
        const float f = Q.cam_f[cb + v];
        const float cx = Q.cam_c[(cb + v) * 2], cy = Q.cam_c[(cb + v) * 2 + 1];
        const float X = L.kp[k][0], Y = L.kp[k][1], Z = L.kp[k][2];
        const float px = Rc[0] * X + Rc[1] * Y + Rc[2] * Z + tc[0];            // camera.py:106-110
        const float py = Rc[3] * X + Rc[4] * Y + Rc[5] * Z + tc[1];
        const float pz = Rc[6] * X + Rc[7] * Y + Rc[8] * Z + tc[2];
        const float u = f * (px / pz) + cx, w_ = f * (py / pz) + cy;            // camera.py:112-116
        const size_t gi = ((size_t)b * V + v) * NKP + k;
        const float rx = Q.gt_xy[gi * 2] - u, ry = Q.gt_xy[gi * 2 + 1] - w_;
        const float wc = Q.w_conf[gi];
        const float w2 = wc * wc;
        const float rx2 = rx * rx, ry2 = ry * ry;
        const float gmx = W.rho2 * (rx2 / (rx2 + W.rho2)), gmy = W.rho2 * (ry2 / (ry2 + W.rho2));   // utils.py:435-438
        part += (double)(w2 * (gmx + gmy));
        if (want_grad) {
            const float dx = rx2 + W.rho2, dy = ry2 + W.rho2;
            const float gu = -w2 * W.data_w2 * (2.f * rx * W.rho2 * W.rho2 / (dx * dx));
            const float gv = -w2 * W.data_w2 * (2.f * ry * W.rho2 * W.rho2 / (dy * dy));
            const float gpx = f * gu / pz, gpy = f * gv / pz;
            const float gpz = -f * (gu * px + gv * py) / (pz * pz);
            L.gkp_part[v][k][0] = Rc[0] * gpx + Rc[3] * gpy + Rc[6] * gpz;
            L.gkp_part[v][k][1] = Rc[1] * gpx + Rc[4] * gpy + Rc[7] * gpz;
            L.gkp_part[v][k][2] = Rc[2] * gpx + Rc[5] * gpy + Rc[8] * gpz;
        }
    }
    const double l_data = block_sum(part, L.red, tid, nt) * (double)W.data_w2;    // fitting.py:311-316
    // ---- priors (few terms: one wave's worth of threads) ----
    const bool use_vp = (W.flags & MVFIT_F_VPOSER) != 0;
    double pp = 0.0;     // body_pose^2 (or z^2) partial
    for (int i = tid; i < (use_vp ? 32 : 69); i += nt) {
        const float v = use_vp ? L.x[X_EMB + i] : L.theta[3 + i];
        pp += (double)v * (double)v;
    }
    const double sq = block_sum(pp, L.red, tid, nt);
    double bb = 0.0;
    for (int i = tid; i < 10; i += nt) bb += (double)L.beta[i] * (double)L.beta[i];
    const double sqb = block_sum(bb, L.red, tid, nt);
    const double wp2 = (double)W.pose_w * (double)W.pose_w;
    double l_pose;
    int dropped = 0;
    if (use_vp) {
        l_pose = sq * wp2;                                                    // fitting.py:327-329
    } else {
        double P;
        if (W.flags & MVFIT_F_PRIOR_GMM) {
            // merged_log_likelihood (prior.py:181-196): min_m 0.5 d^T P_m d - log nll_w_m
            const int Mg = M.gmm_M;
            for (int i = tid; i < Mg * 69; i += nt) {
                const int m = i / 69, r = i - m * 69;
                const float* prow = M.gmm_prec + ((size_t)m * 69 + r) * 69;
                const float* mu = M.gmm_means + m * 69;
                float s = 0.f;
                for (int c = 0; c < 69; ++c) s = fmaf(prow[c], L.theta[3 + c] - mu[c], s);
                L.gmm_t[m][r] = s;
            }
            __syncthreads();
            if (tid < Mg) {
                const float* mu = M.gmm_means + tid * 69;
                float qd = 0.f;
                for (int r = 0; r < 69; ++r) qd = fmaf(L.gmm_t[tid][r], L.theta[3 + r] - mu[r], qd);
                L.gmm_ll[tid] = 0.5f * qd - M.gmm_lognw[tid];
            }
            __syncthreads();
            int sel = 0;
            float best = L.gmm_ll[0];
            for (int m = 1; m < Mg; ++m) if (L.gmm_ll[m] < best) { best = L.gmm_ll[m]; sel = m; }
            if (tid == 0) L.gmm_sel = sel;
            P = (double)best;
        } else {
            P = sq;                                                            // prior.py:92-97
        }
        P *= wp2;
        if ((float)P > 5e4f) { P = 0.0; dropped |= 1; }                        // fitting.py:334-335
        l_pose = P + sq * (16.0 * wp2);                                        // fitting.py:336-337
    }
    double l_shape = 0.0;
    if (!(W.flags & MVFIT_F_FIX_SHAPE)) l_shape = sqb * (double)W.shape_w * (double)W.shape_w;   // :339-342
    // angle prior (prior.py:73-89): exp(pose[idx]*sgn)^2 on full_pose[3:66] idx 52,55,9,12
    double l_angle;
    {
        const float e0 = expf(L.theta[3 + 52]), e1 = expf(-L.theta[3 + 55]);
        const float e2 = expf(-L.theta[3 + 9]), e3 = expf(-L.theta[3 + 12]);
        l_angle = ((double)(e0 * e0) + (double)(e1 * e1) + (double)(e2 * e2) + (double)(e3 * e3)) * (double)W.bend_w;
        if ((float)l_angle > 1e4f && !use_vp) { l_angle = 0.0; dropped |= 2; }   // fitting.py:349-350
    }
    const double total = l_data + l_pose + l_shape + l_angle;
    if (tid == 0) {
        L.loss_terms[0] = (float)l_data; L.loss_terms[1] = (float)l_pose; L.loss_terms[2] = (float)l_shape;
        L.loss_terms[3] = (float)l_angle; L.loss_terms[5] = (float)total;
        L.flags_dropped = dropped;
    }
    __syncthreads();
    if (want_grad) {
        for (int i = tid; i < NKP * 3; i += nt) {
            const int k = i / 3, a = i - 3 * k;
            float s = 0.f;
            for (int v = 0; v < V; ++v) s += L.gkp_part[v][k][a];
            L.gkp[k][a] = s;
        }
        __syncthreads();
    }
    return total;
}

// ---------------------------------------------------------------------------------------------
// Adjoint: g_kp -> grad[118]  (oracle/closure_np.py:_backward)
// ---------------------------------------------------------------------------------------------
__device__ void closure_backward(const DevModel& M, ClosureLds& L, const DevWeights& W, int tid, int nt) {
    const bool use_vp = (W.flags & MVFIT_F_VPOSER) != 0;
    // gx = Ksel^T g_kp ; g_tau = sum_k g_kp
    for (int c = tid; c < M.nc; c += nt) {
        const int s = c / 3, a = c - 3 * s;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < NKP; ++k) acc = fmaf(M.ksel_sub[k * NS_MAX + s], L.gkp[k][a], acc);
        L.gx[c] = acc;
    }
    if (tid < 3) {
        float s = 0.f;
        for (int k = 0; k < NKP; ++k) s += L.gkp[k][tid];
        L.gtau[tid] = s;
    }
    __syncthreads();
    // g_vposed = Tr^T gx ; g_Ar, g_At
    for (int c = tid; c < M.nc_pad; c += nt) {
        float v = 0.f;
        if (c < M.nc) {
            const int s = c / 3, bq = c - 3 * s;
            v = L.T[s][0 + bq] * L.gx[3 * s] + L.T[s][4 + bq] * L.gx[3 * s + 1] + L.T[s][8 + bq] * L.gx[3 * s + 2];
        }
        L.gvp[c] = v;
    }
    for (int i = tid; i < NJ * 12; i += nt) {
        const int j = i / 12, e = i - j * 12;
        float acc = 0.f;
        if (e < 9) {
            const int a = e / 3, bq = e - 3 * a;
            for (int s = 0; s < M.ns; ++s) acc = fmaf(M.w_sub[s * NJ + j] * L.gx[3 * s + a], L.vposed[3 * s + bq], acc);
            L.gAr[j][e] = acc;
        } else {
            const int a = e - 9;
            for (int s = 0; s < M.ns; ++s) acc = fmaf(M.w_sub[s * NJ + j], L.gx[3 * s + a], acc);
            L.gAt[j][a] = acc;
        }
    }
    __syncthreads();
    // A_j = [Gr_j | Gt_j - Gr_j J_j]
    for (int i = tid; i < NJ * 12; i += nt) {
        const int j = i / 12, e = i - j * 12;
        if (e < 9) { const int a = e / 3, bq = e - 3 * a; L.gGr[j][e] = L.gAr[j][e] - L.gAt[j][a] * L.J[j][bq]; }
        else { const int a = e - 9; L.gGt[j][a] = L.gAt[j][a];
               L.gJ[j][a] = -(L.Gr[j][0 + a] * L.gAt[j][0] + L.Gr[j][3 + a] * L.gAt[j][1] + L.Gr[j][6 + a] * L.gAt[j][2]); }
    }
    // g_coef = PD_sub . g_vposed  (transposed contraction, k-split over columns)
    {
        const int npq = KROWS >> 2;                 // 56 float4 row groups
        const int ncs = max(1, min(nt / npq, 16));   // column slices
        const int cps = (M.nc_pad + ncs - 1) / ncs;
        const int pq = tid % npq, cs = tid / npq;
        if (cs < ncs) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const int c0 = cs * cps, c1 = min(M.nc_pad, c0 + cps);
            const float4* src = reinterpret_cast<const float4*>(M.pd_subT) + pq;
#pragma unroll 8
            for (int c = c0; c < c1; ++c) {
                const float4 v = src[(size_t)c * npq];
                const float g = L.gvp[c];
                acc.x = fmaf(g, v.x, acc.x); acc.y = fmaf(g, v.y, acc.y);
                acc.z = fmaf(g, v.z, acc.z); acc.w = fmaf(g, v.w, acc.w);
            }
            float* dst = L.scratch + cs * KROWS + 4 * pq;
            dst[0] = acc.x; dst[1] = acc.y; dst[2] = acc.z; dst[3] = acc.w;
        }
        __syncthreads();
        for (int p = tid; p < KROWS; p += nt) {
            float s = 0.f;
            for (int k = 0; k < ncs; ++k) s += L.scratch[k * KROWS + p];
            L.gcoef[p] = s;
        }
    }
    __syncthreads();
    // kinematic chain, deepest parents first: children lists give a fixed summation order
    for (int lv = M.nlevels - 2; lv >= 0; --lv) {
        const int n = (M.level_start[lv + 1] - M.level_start[lv]) * 12;
        for (int i = tid; i < n; i += nt) {
            const int p = M.level_joints[M.level_start[lv] + i / 12], e = i % 12;
            float acc = 0.f;
            for (int ci = M.child_start[p]; ci < M.child_start[p + 1]; ++ci) {
                const int c = M.child_list[ci];
                if (e < 9) {
                    const int a = e / 3, bq = e - 3 * a;
                    acc += L.gGr[c][a * 3] * L.Rm[c][bq * 3] + L.gGr[c][a * 3 + 1] * L.Rm[c][bq * 3 + 1] +
                           L.gGr[c][a * 3 + 2] * L.Rm[c][bq * 3 + 2] + L.gGt[c][a] * L.tm[c][bq];
                } else {
                    acc += L.gGt[c][e - 9];
                }
            }
            if (e < 9) L.gGr[p][e] += acc; else L.gGt[p][e - 9] += acc;
        }
        __syncthreads();
    }
    for (int i = tid; i < NJ * 12; i += nt) {
        const int j = i / 12, e = i - j * 12;
        if (j == 0) {
            if (e < 9) L.gRm[0][e] = L.gGr[0][e]; else L.gtm[0][e - 9] = L.gGt[0][e - 9];
        } else {
            const int pa = M.parents[j];
            if (e < 9) { const int a = e / 3, bq = e - 3 * a;
                L.gRm[j][e] = L.Gr[pa][a] * L.gGr[j][bq] + L.Gr[pa][3 + a] * L.gGr[j][3 + bq] + L.Gr[pa][6 + a] * L.gGr[j][6 + bq]; }
            else { const int a = e - 9;
                L.gtm[j][a] = L.Gr[pa][a] * L.gGt[j][0] + L.Gr[pa][3 + a] * L.gGt[j][1] + L.Gr[pa][6 + a] * L.gGt[j][2]; }
        }
    }
    __syncthreads();
    // g_J, g_R, g_scale
    for (int i = tid; i < NJ * 3; i += nt) {
        const int j = i / 3, a = i - 3 * j;
        float s = L.gJ[j][a] + L.gtm[j][a];
        for (int ci = M.child_start[j]; ci < M.child_start[j + 1]; ++ci) s -= L.gtm[M.child_list[ci]][a];
        L.gJ[j][a] = s;
    }
    for (int i = tid; i < NJ * 9; i += nt) {
        const int j = i / 9, e = i - 9 * j;
        L.gR[j][e] = (j == 0) ? L.scale * L.gRm[0][e] : (L.gRm[j][e] + L.gcoef[9 * (j - 1) + e]);
    }
    if (tid == 0) {
        float s = 0.f;
        for (int e = 0; e < 9; ++e) s += L.gRm[0][e] * L.R[0][e];
        L.gscale = s;
    }
    __syncthreads();
    // Rodrigues adjoint per joint
    if (tid < NJ) {
        const float rx = L.theta[3 * tid], ry = L.theta[3 * tid + 1], rz = L.theta[3 * tid + 2];
        const float ex = rx + 1e-8f, ey = ry + 1e-8f, ez = rz + 1e-8f;
        const float a = L.ang[tid];
        const float kx = rx / a, ky = ry / a, kz = rz / a;
        float sn, cs;
        sincosf(a, &sn, &cs);
        const float oc = 1.f - cs;
        const float K[9] = {0.f, -kz, ky, kz, 0.f, -kx, -ky, kx, 0.f};
        float KK[9];
        mat3_mul(K, K, KK);
        const float* g = L.gR[tid];
        float gK_dot = 0.f, gKK_dot = 0.f;
#pragma unroll
        for (int e = 0; e < 9; ++e) { gK_dot += g[e] * K[e]; gKK_dot += g[e] * KK[e]; }
        float ga = cs * gK_dot + sn * gKK_dot;
        // gK = sn g + oc (g K^T + K^T g)
        float gKt[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int jq = 0; jq < 3; ++jq) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int m = 0; m < 3; ++m) { s1 += g[i * 3 + m] * K[jq * 3 + m]; s2 += K[m * 3 + i] * g[m * 3 + jq]; }
                gKt[i * 3 + jq] = sn * g[i * 3 + jq] + oc * (s1 + s2);
            }
        const float gkx = gKt[7] - gKt[5], gky = gKt[2] - gKt[6], gkz = gKt[3] - gKt[1];
        ga -= (gkx * rx + gky * ry + gkz * rz) / (a * a);
        L.gtheta[3 * tid] = gkx / a + ga * ex / a;
        L.gtheta[3 * tid + 1] = gky / a + ga * ey / a;
        L.gtheta[3 * tid + 2] = gkz / a + ga * ez / a;
    }
    // g_beta = shapedirs part of g_coef + J_S^T g_J  (+ shape prior)
    if (tid >= 64 && tid < 74) {
        const int l = tid - 64;
        float s = L.gcoef[207 + l];
        for (int i = 0; i < NJ * 3; ++i) s = fmaf(M.J_S[i * 10 + l], (&L.gJ[0][0])[i], s);
        if (!(W.flags & MVFIT_F_FIX_SHAPE)) s += 2.f * L.beta[l] * W.shape_w * W.shape_w;
        L.gbeta[l] = s;
    }
    __syncthreads();
    // priors on the pose
    const float wp2 = W.pose_w * W.pose_w;
    if (!use_vp) {
        if ((W.flags & MVFIT_F_PRIOR_GMM) && !(L.flags_dropped & 1)) {
            const int m = L.gmm_sel;
            for (int i = tid; i < 69; i += nt) {
                // 0.5 (P d + P^T d): P d is gmm_t[m]; P^T d by columns
                const float* mu = M.gmm_means + m * 69;
                const float* P = M.gmm_prec + (size_t)m * 69 * 69;
                float s = 0.f;
                for (int r = 0; r < 69; ++r) s = fmaf(P[r * 69 + i], L.theta[3 + r] - mu[r], s);
                L.gtheta[3 + i] += 0.5f * (L.gmm_t[m][i] + s) * wp2;
            }
        }
        __syncthreads();
        for (int i = tid; i < 69; i += nt) {
            float g = L.gtheta[3 + i];
            const float bp = L.theta[3 + i];
            if (!(W.flags & MVFIT_F_PRIOR_GMM) && !(L.flags_dropped & 1)) g += 2.f * bp * wp2;
            g += 2.f * bp * 16.f * wp2;
            L.gtheta[3 + i] = g;
        }
        __syncthreads();
    }
    if (tid < 4 && !(L.flags_dropped & 2)) {
        const int idx[4] = {3 + 52, 3 + 55, 3 + 9, 3 + 12};
        const float sg[4] = {1.f, -1.f, -1.f, -1.f};
        const float th = L.theta[idx[tid]];
        L.gtheta[idx[tid]] += 2.f * expf(2.f * th * sg[tid]) * sg[tid] * W.bend_w;
    }
    __syncthreads();
    // assemble the flat gradient (frozen / unused slots are zero)
    for (int i = tid; i < DPAD; i += nt) {
        float g = 0.f;
        if (i < X_GO) g = (W.flags & MVFIT_F_FIX_SHAPE) ? 0.f : L.gbeta[i];
        else if (i < X_BP) g = L.gtheta[i - X_GO];
        else if (i < X_TR) g = use_vp ? 0.f : L.gtheta[3 + i - X_BP];
        else if (i < X_SC) g = L.gtau[i - X_TR];
        else if (i == X_SC) g = (W.flags & MVFIT_F_FIX_SCALE) ? 0.f : L.gscale;
        else if (i < DV) g = use_vp ? 2.f * L.x[i] * wp2 : 0.f;              // fitting.py:328 (d/dz |z|^2 w^2)
        L.grad[i] = g;
    }
    __syncthreads();
    if (use_vp) vposer_backward(M, L, tid, nt);
}

}  // namespace mvfit

// One closure evaluation for ONE problem, executed by one 512-thread workgroup with every
// intermediate AND every small model constant in LDS; the only global traffic inside a closure
// is the streamed blendshape basis of the objective's vertices (fwd: pd_sub, bwd: pd_subT) and,
// when enabled, the VPoser / GMM weights.
//
// Forward = SMPL.forward (reference code/smplx/body_models_scale.py:327-412, code/smplx/lbs.py:135-222)
// restricted to the vertices the objective reads, + SMPLifyLoss.forward
// (code/utils/fitting.py:290-415, camera code/camera.py:93-117, GMoF code/utils/utils.py:427-438,
// priors code/prior.py:53-231, VPoser decoder code/model/VPoser.py:218-232).
// Backward = the hand-derived adjoint that replaces total_loss.backward() (fitting.py:190-192);
// it is the transcription of oracle/closure_np.py:_backward, which matches the reference's
// autograd to 1e-16 in float64.
//
// Latency structure (the closure is ~0.3 MFLOP: what costs is dependent latency, not work):
//   E1  Rodrigues + J(beta) + blendshape coefficients + relative transforms     (all threads)
//   E2  wave 0: kinematic chain, level by level, 12 lanes per joint, no s_barrier
//       waves 1-7: stream pd_sub (k-split partial sums)                         [overlapped]
//   E3  v_posed, skinning transform rows, skinned positions                     (thread per coordinate)
//   E4  thread per (view, keypoint): keypoint, projection, GMoF, d/d keypoint; priors on wave 4;
//       DPP wave reductions -> LDS
//   E5  g_x, g_vposed                                                            (thread per coordinate)
//   E6  g_A = W^T [g_x v_posed^T | g_x]: 16-lane row per joint, DPP butterfly
//   E7  wave 0: chain adjoint; waves 1-7: stream pd_subT (g_coef partials)      [overlapped]
//   E8  g_R assembly, g_J, g_beta; E9 Rodrigues adjoint + pose priors; flat gradient
#pragma once
#include <cstddef>
#include "mvfit_device.h"

namespace mvfit {

// Blocks of the LDS image that are moved to / from HBM as 16-byte words (one load per thread, all
// issued before a single wait: a kernel prologue costs one memory round trip, not one per array).
struct ObsBlock {                   // per-problem observations, packed by mvfit_set_problems
    float camR[MVFIT_MAX_VIEWS][9];
    float camt[MVFIT_MAX_VIEWS][3];
    float camf[MVFIT_MAX_VIEWS];
    float camc[MVFIT_MAX_VIEWS][2];
    float gt[MVFIT_MAX_VIEWS * NKP * 2];
    float wc[MVFIT_MAX_VIEWS * NKP];
    float gt3d[NKP * 3];            // use_3d targets (zeros / zero confidence when unused)
    float c3d[NKP];
};
struct PoseBlock {                  // everything pose_prep + the chain derive from x (handed from launch to launch)
    float theta[72];
    float R[NJ][9];
    float rod[NJ][3];               // angle, sin, cos
    float J[NJ][3];
    // 3x4 transforms, row-major [a][4] = [rotation | translation]; rows NJ.. of Mj are zero (absent child)
    float Mj[32][12];               // relative transforms [Rm | tm]          (lbs.py:341-348)
    float G[NJ][12];                // chained transforms  [Gr | Gt]          (lbs.py:349-355)
    float A[NJ][12];                // skinning transforms                    (lbs.py:365-368)
};
struct OptBlock {                   // optimiser state that survives between closure rounds
    float x[DPAD];                  // current trial point, flat parameter layout
    float lb_ro[104];               // ro = 1/(y.s) per history slot
    float lb_ys[104];               // y.s per history slot (the diagonal of the compact form, lbfgs_device.h)
    LbState lbS;
    LbVecs<float> lbV[LB_LANES];    // the optimiser's working vectors, lane-major (lbfgs_device.h)
};
struct VpBlock {                    // VPoser decoder state of the current trial point (image of ClosureLds::vp_pre1..vp_cpad)
    float pre1[512];
    float pre2[512];
    float cache[23][25];
    float pad;
};
static_assert(sizeof(ObsBlock) % 16 == 0 && sizeof(PoseBlock) % 16 == 0 && sizeof(OptBlock) % 16 == 0 && sizeof(VpBlock) % 16 == 0,
              "16-byte blocks");

// The SDF term's per-problem result (SdfAdj, sdf_term.hip) is written by ANOTHER kernel - in the chained rounds a launch
// boundary earlier, in the service rounds of the single-launch fit (round 6) while this kernel runs, on CUs of any XCD: read at
// agent scope (sc1: past this CU's L1 and this XCD's L2, which may hold the previous round's line).
__device__ __forceinline__ float sdf_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct ClosureLds {
    ModelLds M;                     // model constants
    __attribute__((aligned(16))) ObsBlock obs;
    __attribute__((aligned(16))) OptBlock opt;
    __attribute__((aligned(16))) PoseBlock pose;
    // state of one evaluation
    float coef[KROWS];
    __attribute__((aligned(16))) float vposed[NC_MAX];
    __attribute__((aligned(16))) float xs[NC_MAX];
    __attribute__((aligned(16))) float T[NS_MAX][12];
    float kp[NKP][3];
    float gkp_part[MVFIT_MAX_VIEWS][NKP][3];
    float gkp[NKP][3];              // summed over views
    float gkp3[NKP][3];             // use_3d part of d/d keypoint
    float gx[NC_MAX];
    __attribute__((aligned(16))) float gvp[NC_MAX];
    __attribute__((aligned(16))) float gG[32][12];     // adjoint of G (rows NJ.. zero)
    __attribute__((aligned(16))) float gM[NJ][12];     // adjoint of Mj = [g_Rm | g_tm]
    float gJ[NJ][3];
    float gR[NJ][9];
    float gtheta[72];
    float gbeta[12];
    float gtau[4];
    float gscale;
    float loss_terms[6];            // data, pose, shape, angle, (coll), total
    int flags_dropped;              // bit0: pose prior dropped, bit1: angle prior dropped
    int gmm_sel;
    int sh_stage, sh_status;        // optimiser scalars broadcast from wave 0 to the block
    unsigned sh_pass_done;          // asynchronous fit: last value read from the ring's pass_done (publishing wave only)
    unsigned vp_seq;                // decoder service (vposer_service.h): number of the last request of this problem
    int vp_remote;                  // 1: the decoder layers run on the launch's helper workgroups
    const SdfAdj* sdf_adj;          // SDF term of this problem (sdf_term.hip), or null
    float sdf_fac;                  // 2 w^2 S: factor on the S-adjoint (0 when the term is off)
    unsigned sh_sdf_ok;             // service rounds: the answer arrived (0: timed out)
    int sh_next;                    // work queue: the problem this workgroup takes next
    int sh_prob;                    // the problem this workgroup is fitting (global index)
    const unsigned* sdf_wait_tag;   // service rounds: the answer tag the loss's combining wave waits for before it reads S (or null)
    unsigned sdf_wait_want;
    unsigned* sdf_wait_stats;       // where a timed-out wait is counted (AsyncRing::stats + 3)
    double total;
    double red_d[STEP_NW + 2];      // per-wave partials of the data term; [5..7] priors; [STEP_NW] 3-D term
    float red_f[STEP_NW][4];        // per-wave partials: g_tau (3)
    __attribute__((aligned(16))) float grad[DPAD];
    // scratch for k-split partial sums: max(8 * nc_pad, 8 * KROWS) - closure phases only; the optimiser's work area
    // (direction computation) shares the bytes: the two never overlap in time
    union {
        __attribute__((aligned(16))) float scratch[8 * NC_MAX];
        __attribute__((aligned(16))) LbWork<float> lbW;
    };
    // GMM
    float gmm_d[72];
    __attribute__((aligned(16))) float gmm_t[8][72];
    float gmm_ll[8];
    // per-stage weights and optimiser options: read from LDS inside the round loop so that they do not
    // pin ~100 SGPRs across it
    DevWeights sw[MVFIT_MAX_STAGES];
    __attribute__((aligned(16))) LbOpts opts;
    // VPoser activations (decoder fwd/bwd) - LAST: a single-launch fit without VPoser lays its L-BFGS history over them
    // (persistent_tail_offset).  What the decoder adjoint needs from the forward is handed from launch to launch as one
    // block (VpBlock = vp_pre1 .. vp_cpad)
    __attribute__((aligned(16))) float vp_pre1[512];
    float vp_pre2[512];
    float vp_cache[23][25];
    float vp_cpad;
    __attribute__((aligned(16))) float vp_h[512];
    float vp_g[512];
    float vp_o[144];
    float vp_go[144];
};
static_assert(sizeof(LbWork<float>) <= 8 * NC_MAX * 4, "the optimiser's work area lives in the closure's scratch bytes");
static_assert(offsetof(ClosureLds, vp_pre1) % 16 == 0, "history tail alignment");

// Kernel prologue: the LDS image blocks as 16-byte-word copies, all loads of a thread in flight before
// the first wait.  Null pointers skip a block; x_g (flat parameters, DV floats) fills L.opt.x.
__device__ __forceinline__ void prologue(ClosureLds& L, const DevModel& M, const ObsBlock* obs_g, const PoseBlock* pose_g,
                                         const OptBlock* opt_g, const float* vposed_g, const float* xs_g,
                                         const float* x_g, int tid, const SdfAdj* sdf_adj = nullptr,
                                         const VpBlock* vp_g = nullptr) {
    constexpr int n16 = sizeof(ModelLds) / 16;
    constexpr int nobs = sizeof(ObsBlock) / 16, npose = sizeof(PoseBlock) / 16, nopt = sizeof(OptBlock) / 16;
    static_assert(n16 <= 4 * STEP_NT && nobs <= STEP_NT && npose <= STEP_NT && nopt <= STEP_NT, "one word per thread");
    const int4* src = reinterpret_cast<const int4*>(M.mlds);
    // (clamped indices instead of predicated loads: the surplus words are simply not stored)
    const int4 m0 = src[min(tid, n16 - 1)];
    const int4 m1 = src[min(tid + STEP_NT, n16 - 1)];
    const int4 m2 = src[min(tid + 2 * STEP_NT, n16 - 1)];
    const int4 m3 = src[min(tid + 3 * STEP_NT, n16 - 1)];
    // (plain conditional loads into zero-initialised values: a select between a loaded word and a zero CONSTANT OBJECT would
    // put that object - 16 bytes of scratch - on the stack)
    float4 vobs = make_float4(0.f, 0.f, 0.f, 0.f), vpose = vobs, vopt = vobs, vvps = vobs, vxss = vobs, vvp = vobs;
    if (obs_g && tid < nobs) vobs = reinterpret_cast<const float4*>(obs_g)[tid];
    if (pose_g && tid < npose) vpose = reinterpret_cast<const float4*>(pose_g)[tid];
    if (opt_g && tid < nopt) vopt = reinterpret_cast<const float4*>(opt_g)[tid];
    if (vposed_g && tid < NC_MAX / 4) vvps = reinterpret_cast<const float4*>(vposed_g)[tid];
    if (xs_g && tid < NC_MAX / 4) vxss = reinterpret_cast<const float4*>(xs_g)[tid];
    const float xv = (x_g && tid < DV) ? x_g[tid] : 0.f;
    constexpr int nvp = sizeof(VpBlock) / 16;
    static_assert(nvp <= STEP_NT, "one word per thread");
    if (vp_g && tid < nvp) vvp = reinterpret_cast<const float4*>(vp_g)[tid];
    int4* dst = reinterpret_cast<int4*>(&L.M);
    if (tid < n16) dst[tid] = m0;
    if (tid + STEP_NT < n16) dst[tid + STEP_NT] = m1;
    if (tid + 2 * STEP_NT < n16) dst[tid + 2 * STEP_NT] = m2;
    if (tid + 3 * STEP_NT < n16) dst[tid + 3 * STEP_NT] = m3;
    if (obs_g && tid < nobs) reinterpret_cast<float4*>(&L.obs)[tid] = vobs;
    if (pose_g) { if (tid < npose) reinterpret_cast<float4*>(&L.pose)[tid] = vpose; }
    else if (tid >= 256 && tid < 256 + 8 * 12) (&L.pose.Mj[NJ][0])[tid - 256] = 0.f;     // "no child" rows
    if (opt_g && tid < nopt) reinterpret_cast<float4*>(&L.opt)[tid] = vopt;
    if (vposed_g && tid < NC_MAX / 4) { reinterpret_cast<float4*>(L.vposed)[tid] = vvps; reinterpret_cast<float4*>(L.xs)[tid] = vxss; }
    if (x_g && tid < DPAD) L.opt.x[tid] = xv;
    if (vp_g && tid < nvp) reinterpret_cast<float4*>(L.vp_pre1)[tid] = vvp;
    if (tid >= 384 && tid < 384 + 8 * 12) (&L.gG[NJ][0])[tid - 384] = 0.f;               // "no child" rows
    if (tid == 511) { L.sdf_adj = sdf_adj; L.sdf_fac = 0.f; L.vp_remote = (M.vps.nsets > 0 && (int)blockIdx.x / max(M.vps.nsets, 1) < VPS_PMAX) ? 1 : 0; L.vp_seq = 0u; }   // (a set has VPS_PMAX slots; the host never launches more problems per set)
}

__device__ __forceinline__ void mat3_mul(const float* a, const float* b, float* c) {   // c = a b
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            c[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}

// Rodrigues (lbs.py:269-300): angle = ||r + 1e-8||, k = r / angle, R = I + sin K + (1 - cos) K^2
__device__ __forceinline__ void rodrigues(const float* r, float* R, float* rod) {
    const float rx = r[0], ry = r[1], rz = r[2];
    const float ex = rx + 1e-8f, ey = ry + 1e-8f, ez = rz + 1e-8f;
    const float a = sqrtf(ex * ex + ey * ey + ez * ez);
    const float ia = 1.0f / a;
    const float kx = rx * ia, ky = ry * ia, kz = rz * ia;
    float sn, cs;
    sincosf(a, &sn, &cs);
    const float oc = 1.0f - cs;
    // K = [[0,-kz,ky],[kz,0,-kx],[-ky,kx,0]] ; K^2 = k k^T - |k|^2 I
    const float kk = kx * kx + ky * ky + kz * kz;
    R[0] = 1.f + oc * (kx * kx - kk); R[1] = -sn * kz + oc * kx * ky;   R[2] = sn * ky + oc * kx * kz;
    R[3] = sn * kz + oc * kx * ky;    R[4] = 1.f + oc * (ky * ky - kk); R[5] = -sn * kx + oc * ky * kz;
    R[6] = -sn * ky + oc * kx * kz;   R[7] = sn * kx + oc * ky * kz;    R[8] = 1.f + oc * (kz * kz - kk);
    rod[0] = a; rod[1] = sn; rod[2] = cs;
}

// Streamed GEMV for the decoder: out[o] = sum_{i < n} W[i * ld + o] * h[i], o < ncols (multiple of 4).
// Thread = (row slice sl, column quad q): 16-byte loads, U rows requested before the first FMA - the weights
// arrive from the Infinity Cache / HBM (the L2 does not keep 2.8 MB of decoder weights next to the basis
// streams), so the stream is bound by bytes in flight per CU.  Partials go to part[sl * ncols + o]; the caller
// sums the slices in ascending order after a barrier (deterministic, independent of the launch geometry).
template <int U>
__device__ __forceinline__ void gemv4_partial(const float* __restrict__ W, int ld, int n, int ncols, int nsl,
                                              const float* h, float* part, int tid) {
    const int nq = ncols >> 2;
    const int sl = tid / nq, q = tid - sl * nq;
    if (sl >= nsl) return;
    const int per = (n + nsl - 1) / nsl;
    const int i0 = sl * per, i1 = min(n, i0 + per);
    const float4* Wq = reinterpret_cast<const float4*>(W) + q;
    const int ld4 = ld >> 2;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = i0;
    for (; i + U <= i1; i += U) {
        float4 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = Wq[(size_t)(i + u) * ld4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float hv = h[i + u];
            acc.x = fmaf(w[u].x, hv, acc.x); acc.y = fmaf(w[u].y, hv, acc.y);
            acc.z = fmaf(w[u].z, hv, acc.z); acc.w = fmaf(w[u].w, hv, acc.w);
        }
    }
    if (i < i1) {                                    // last, partial chunk: clamped row index, zero multiplier
        float4 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = Wq[(size_t)min(i + u, i1 - 1) * ld4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float hv = h[min(i + u, i1 - 1)];          // read first, select second: as "cond ? h[..] : 0" every read becomes a branch around it
            hv = (i + u < i1) ? hv : 0.f;
            acc.x = fmaf(w[u].x, hv, acc.x); acc.y = fmaf(w[u].y, hv, acc.y);
            acc.z = fmaf(w[u].z, hv, acc.z); acc.w = fmaf(w[u].w, hv, acc.w);
        }
    }
    *reinterpret_cast<float4*>(&part[sl * ncols + 4 * q]) = acc;
}

// ---------------------------------------------------------------------------------------------
// VPoser decoder forward (VPoser.py:218-232,165-174,263-273,29-156) for one latent, block-wide.
// Weight rows are streamed with the whole block; every GEMV output is k-split over lanes.
// ---------------------------------------------------------------------------------------------
constexpr int VP_U = 16;          // rows (16-byte loads) in flight per thread in the decoder streams

// the three layers in this workgroup: z -> L.vp_o[138] (+ the pre-activations the adjoint needs)
__device__ __forceinline__ void vposer_layers_local(const DevModel& M, ClosureLds& L, int tid) {
    constexpr int nt = STEP_NT;
    // h1 = lrelu(W1 z + b1): one output per thread
    for (int o = tid; o < 512; o += nt) {
        float s = M.vp_b1[o];
        float w[32];                                   // transposed copy: consecutive threads, consecutive addresses
#pragma unroll
        for (int i = 0; i < 32; ++i) w[i] = M.vp_w1T[i * 512 + o];
#pragma unroll
        for (int i = 0; i < 32; ++i) s = fmaf(w[i], L.opt.x[X_EMB + i], s);
        L.vp_pre1[o] = s;
        L.vp_h[o] = s > 0.f ? s : 0.2f * s;
    }
    __syncthreads();
    PH_T(26);
    // h2 = lrelu(W2 h1 + b2): w2T[i][o], thread o, consecutive threads read consecutive o
    gemv4_partial<VP_U>(M.vp_w2T, 512, 512, 512, 4, L.vp_h, L.scratch, tid);
    __syncthreads();
    for (int o = tid; o < 512; o += nt) {
        const float s = M.vp_b2[o] + (((L.scratch[o] + L.scratch[512 + o]) + L.scratch[1024 + o]) + L.scratch[1536 + o]);
        L.vp_pre2[o] = s;
        L.vp_h[o] = s > 0.f ? s : 0.2f * s;
    }
    __syncthreads();
    PH_T(27);
    // out = W3 h2 + b3: 36 column quads (144 padded outputs) x 14 row slices
    gemv4_partial<VP_U>(M.vp_w3T, 144, 512, 144, 14, L.vp_h, L.scratch, tid);
    __syncthreads();
    for (int o = tid; o < 138; o += nt) {
        float sacc = M.vp_b3[o];
#pragma unroll
        for (int sl = 0; sl < 14; ++sl) sacc += L.scratch[sl * 144 + o];
        L.vp_o[o] = sacc;
    }
    __syncthreads();
    PH_T(28);
}

// Answers of the eight helpers of this problem's set -> dst[h * VPS_GRAN + o], o < n_used (n_used * 8 <= 3 * STEP_NT).
// false when an answer did not arrive within 50 ms.
__device__ __forceinline__ bool vps_collect(const unsigned long long* rp, unsigned tag, int n_used, float* dst, int tid) {
    int idx[3];
    unsigned pending = 0u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int item = tid + STEP_NT * k;
        const int h = item / n_used, o = item - h * n_used;
        idx[k] = h * VPS_GRAN + o;
        if (h < VPS_SLICES) pending |= 1u << k;
    }
    const long long t0 = wall_clock64();
    int it = 0;
    while (pending) {
        unsigned long long g[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) if ((pending >> k) & 1u) g[k] = vps_load(rp + idx[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (((pending >> k) & 1u) && vps_tag(g[k]) == tag) { dst[idx[k]] = vps_val(g[k]); pending &= ~(1u << k); }
        if (pending) {
            if ((++it & 63) == 0 && wall_clock64() - t0 > 5000000) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return pending == 0u;
}

__device__ __forceinline__ unsigned long long* vps_request_slot(const VpService& V) {
    const int pl = (int)blockIdx.x, s = pl % V.nsets, p = pl / V.nsets;
    return V.req + ((size_t)s * VPS_PMAX + p) * VPS_GRAN;
}
__device__ __forceinline__ const unsigned long long* vps_answer_slot(const VpService& V) {
    const int pl = (int)blockIdx.x, s = pl % V.nsets, p = pl / V.nsets;
    return V.resp + ((size_t)s * VPS_PMAX + p) * VPS_SLICES * VPS_GRAN;
}

// the three layers on the helpers (vposer_service.h): false = no answer, the caller decodes locally from now on
__device__ __forceinline__ bool vposer_layers_remote(const DevModel& M, ClosureLds& L, int tid) {
    const VpService& V = M.vps;
    const unsigned seq = L.vp_seq + 1u, tag = seq << 2 | VPS_FWD;
    if (tid < 32) vps_store(vps_request_slot(V) + tid, L.opt.x[X_EMB + tid], tag);
    PH_T(26);
    const bool ok = vps_collect(vps_answer_slot(V), tag, 138, L.scratch, tid);
    const int bad = __syncthreads_or(ok ? 0 : 1);
    PH_T(27);
    if (bad) {
        // local from now on; tell the set's helpers at once (a BYE numbered past the request that timed out: accepted whether
        // or not the helpers still get to that request) - they must not wait 0.2 s for this problem before they can retire
        if (tid == 0) { L.vp_remote = 0; atomicAdd(V.stat, 1u); vps_store(vps_request_slot(V), 0.f, (seq + 1u) << 2 | VPS_BYE); }
        __syncthreads();
        return false;
    }
    if (tid == 0) L.vp_seq = seq;
    if (tid < 138) {
        float acc = M.vp_b3[tid];
#pragma unroll
        for (int h = 0; h < VPS_SLICES; ++h) acc += L.scratch[h * VPS_GRAN + tid];
        L.vp_o[tid] = acc;
    }
    __syncthreads();
    PH_T(28);
    return true;
}

// REMOTE: the single-launch fit kernel (its launches may carry decoder helpers); every other kernel instantiates the
// local decoder only - their per-round launches must stay below the scratch size the runtime keeps resident.
template <bool REMOTE>
__device__ __forceinline__ void vposer_forward(const DevModel& M, ClosureLds& L, int tid) {
    constexpr int nt = STEP_NT;
    bool decoded = false;
    if constexpr (REMOTE) decoded = L.vp_remote && vposer_layers_remote(M, L, tid);
    if (!decoded) vposer_layers_local(M, L, tid);
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
        L.pose.theta[3 + 3 * j + 0] = qn[1] * kk;
        L.pose.theta[3 + 3 * j + 1] = qn[2] * kk;
        L.pose.theta[3 + 3 * j + 2] = qn[3] * kk;
        // cache for the backward
        C[0] = n1; C[1] = b1[0]; C[2] = b1[1]; C[3] = b1[2]; C[4] = d; C[5] = n2;
        C[6] = b2[0]; C[7] = b2[1]; C[8] = b2[2]; C[9] = a2[0]; C[10] = a2[1]; C[11] = a2[2];
        C[12] = (float)cs; C[13] = t; C[14] = q[0]; C[15] = q[1]; C[16] = q[2]; C[17] = q[3];
        C[18] = qn[0]; C[19] = qn[1]; C[20] = qn[2]; C[21] = qn[3]; C[22] = s2; C[23] = tt; C[24] = kk;
    }
    __syncthreads();
}

// g_z += (d body_pose / d z)^T g_body_pose   (transcription of oracle vposer_decode_bwd)
template <bool REMOTE>
__device__ __forceinline__ void vposer_backward(const DevModel& M, ClosureLds& L, int tid) {
    constexpr int nt = STEP_NT;
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
    PH_T(29);
    if (REMOTE && L.vp_remote) {
        const VpService& V = M.vps;
        const unsigned seq = L.vp_seq + 1u, tag = seq << 2 | VPS_BWD;
        if (tid < 138) vps_store(vps_request_slot(V) + tid, L.vp_go[tid], tag);
        const bool ok = vps_collect(vps_answer_slot(V), tag, 32, L.scratch, tid);
        const int bad = __syncthreads_or(ok ? 0 : 1);
        PH_T(30);
        if (!bad) {
            if (tid == 0) L.vp_seq = seq;
            if (tid < 32) {
                float acc = L.scratch[tid];
#pragma unroll
                for (int h = 1; h < VPS_SLICES; ++h) acc += L.scratch[h * VPS_GRAN + tid];
                L.grad[X_EMB + tid] += acc;
            }
            __syncthreads();
            PH_T(31);
            return;
        }
        // no answer: local from now on (BYE as in the forward); the local adjoint needs the pre-activations of the local forward
        if (tid == 0) { L.vp_remote = 0; atomicAdd(V.stat, 1u); vps_store(vps_request_slot(V), 0.f, (seq + 1u) << 2 | VPS_BYE); }
        __syncthreads();
        vposer_layers_local(M, L, tid);
    }
    // g_h2 = W3^T g_o ; through lrelu.  w3[o][i]: thread i, consecutive threads consecutive addresses
    gemv4_partial<VP_U>(M.vp_w3, 512, 138, 512, 4, L.vp_go, L.scratch, tid);
    __syncthreads();
    for (int i = tid; i < 512; i += nt)
        L.vp_g[i] = (((L.scratch[i] + L.scratch[512 + i]) + L.scratch[1024 + i]) + L.scratch[1536 + i]) * (L.vp_pre2[i] > 0.f ? 1.0f : 0.2f);
    __syncthreads();
    PH_T(30);
    // g_h1 = W2^T g_pre2 : w2[o][i], consecutive threads i -> consecutive addresses
    gemv4_partial<VP_U>(M.vp_w2, 512, 512, 512, 4, L.vp_g, L.scratch, tid);
    __syncthreads();
    for (int i = tid; i < 512; i += nt)
        L.vp_h[i] = (((L.scratch[i] + L.scratch[512 + i]) + L.scratch[1024 + i]) + L.scratch[1536 + i]) * (L.vp_pre1[i] > 0.f ? 1.0f : 0.2f);
    __syncthreads();
    PH_T(31);
    // g_z = W1^T g_pre1: 32 outputs x 16 k-slices, DPP row reduction (lane&15 = slice)
    {
        const int i = tid >> 4, ks = tid & 15;        // 512 threads = 32 outputs x 16 slices of 32 consecutive rows
        const float4* wr = reinterpret_cast<const float4*>(M.vp_w1T + (size_t)i * 512 + ks * 32);
        float4 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = wr[u];
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 hv = *reinterpret_cast<const float4*>(&L.vp_h[ks * 32 + 4 * u]);
            s = fmaf(w[u].x, hv.x, s); s = fmaf(w[u].y, hv.y, s); s = fmaf(w[u].z, hv.z, s); s = fmaf(w[u].w, hv.w, s);
        }
        s = row16_sum(s);
        if (ks == 0) L.grad[X_EMB + i] += s;
    }
    __syncthreads();
}

constexpr int FWD_SLICES = 8, FWD_RPS = KROWS / FWD_SLICES;      // 28 basis rows per k-slice of the forward contraction

// ---------------------------------------------------------------------------------------------
// E1: x -> theta, R, J, blendshape coefficients, relative transforms.   lbs.py:183-195,269-348
// Every thread derives what it needs straight from x (Rodrigues is recomputed per output element)
// so the phase has no internal barrier.  Ends with __syncthreads.
// ---------------------------------------------------------------------------------------------
template <bool REMOTE = false>
// (inlining is spelled out for everything the fit kernels use: left to the heuristics it changes with the number of
// kernel instantiations, and the single-launch kernel's register allocation - hence its speed - with it.  The
// single-launch kernel CALLS this phase - pose_prep_decode below -, every other kernel inlines it: a call's frame in a
// per-round kernel would push its scratch size past what the runtime keeps resident between dispatches)
__device__ __forceinline__ void pose_prep_decode_inl(const DevModel& M, ClosureLds& L, uint32_t flags, int tid) {
    if (flags & MVFIT_F_VPOSER) {
        vposer_forward<REMOTE>(M, L, tid);
        if (tid < 3) L.pose.theta[tid] = L.opt.x[X_GO + tid];
        __syncthreads();
    } else {
        // global_orient | body_pose are contiguous in x.  No barrier: pose_prep_elems reads the angles from x itself in this
        // mode (the same values), the copy is for the later phases - which all sit behind pose_prep_elems' barrier
        if (tid < 72) L.pose.theta[tid] = L.opt.x[X_GO + tid];
    }
}

// second half of E1 (force-inlined: the fit kernels keep a thread's prefetched basis rows in registers across it)
__device__ __forceinline__ void pose_prep_elems(const DevModel& M, ClosureLds& L, uint32_t flags, int tid) {
    // the axis-angle vectors: the decoder's output with VPoser (behind its barrier), else x's own slots (no copy to wait for)
    const float* th = (flags & MVFIT_F_VPOSER) ? &L.pose.theta[0] : &L.opt.x[X_GO];
    if (tid < KROWS) {
        // blendshape coefficients: pose_feature (lbs.py:192), betas, zero pad
        const int p = tid;
        float v = 0.f;
        if (p < 207) {
            const int j = 1 + p / 9, e = p % 9;
            float R[9], rod[3];
            rodrigues(&th[3 * j], R, rod);
            float sel = R[0];
#pragma unroll
            for (int q = 1; q < 9; ++q) sel = (e == q) ? R[q] : sel;
            v = sel - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
        } else if (p < 217) {
            v = L.opt.x[X_BETAS + p - 207];
        }
        L.coef[p] = v;
    } else {
        // relative transforms (lbs.py:341-348) + the cached R / J the adjoint needs
        const int i = tid - KROWS;                 // 0..287
        const int j = i / 12, e = i - j * 12;
        if (e < 9) {
            float R[9], rod[3];
            rodrigues(&th[3 * j], R, rod);
            float sel = R[0];
#pragma unroll
            for (int q = 1; q < 9; ++q) sel = (e == q) ? R[q] : sel;
            L.pose.R[j][e] = sel;
            L.pose.Mj[j][4 * (e / 3) + (e % 3)] = (j == 0 ? L.opt.x[X_SC] : 1.0f) * sel;
            if (e < 3) L.pose.rod[j][e] = rod[e];
        } else {
            // J = J_t + J_S beta   (== J_regressor (v_template + shapedirs beta), lbs.py:179-183)
            const int a = e - 9;
            const int pa = L.M.parents[j];
            float s = L.M.J_t[3 * j + a];
#pragma unroll
            for (int l = 0; l < 10; ++l) s = fmaf(L.M.J_S[3 * j + a][l], L.opt.x[X_BETAS + l], s);
            float sp = 0.f;
            if (j > 0) {
                sp = L.M.J_t[3 * pa + a];
#pragma unroll
                for (int l = 0; l < 10; ++l) sp = fmaf(L.M.J_S[3 * pa + a][l], L.opt.x[X_BETAS + l], sp);
            }
            L.pose.J[j][a] = s;
            L.pose.Mj[j][4 * a + 3] = s - sp;
        }
    }
    __syncthreads();
}

template <bool REMOTE = false>
__device__ __attribute__((noinline)) void pose_prep_decode(const DevModel& M, ClosureLds& L, uint32_t flags, int tid) {
    pose_prep_decode_inl<REMOTE>(M, L, flags, tid);
}

template <bool CALL = false>
__device__ __forceinline__ void pose_prep(const DevModel& M, ClosureLds& L, uint32_t flags, int tid) {
    if constexpr (CALL) pose_prep_decode<false>(M, L, flags, tid);
    else pose_prep_decode_inl<false>(M, L, flags, tid);
    pose_prep_elems(M, L, flags, tid);
}

// kinematic chain (lbs.py:349-355) G_j = G_parent M_j in 3x4 form, then A_j = [Gr_j | Gt_j - Gr_j J_j]
// (lbs.py:365-368).  Executed by ONE wave following the host-built schedule: lane = (slot q, row a,
// column c); per pass one table word, one ds_read_b128 of the parent row and three FMAs, handed to
// the next pass through LDS without s_barrier.
__device__ __forceinline__ void chain_forward_wave(ClosureLds& L, int lane) {
    if (lane < 12) L.pose.G[0][lane] = L.pose.Mj[0][lane];
    const int q = lane / 12, e = lane - 12 * q, a = e >> 2, c = e & 3;
    const int npass = L.M.n_fwd;
    constexpr int CH = 8;           // passes per chunk: everything that does not depend on the chain
                                    // (table words, the joints' own M columns) is loaded up front, so the
                                    // dependent part of a pass is ds_read_b128 -> 3 FMA -> ds_write
    wave_lds_fence();
    for (int p0 = 0; p0 < npass; p0 += CH) {
        int w[CH];
        float m0[CH], m1[CH], m2[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) w[u] = (q < 5 && p0 + u < npass) ? L.M.fwd_tab[p0 + u][q] : -1;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int j = w[u] >= 0 ? (w[u] & 0xff) : 0;
            m0[u] = L.pose.Mj[j][c]; m1[u] = L.pose.Mj[j][4 + c]; m2[u] = L.pose.Mj[j][8 + c];
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (p0 + u < npass) {                                  // uniform
                if (w[u] >= 0) {
                    const int j = w[u] & 0xff, pa = w[u] >> 8;
                    const float4 g = *reinterpret_cast<const float4*>(&L.pose.G[pa][4 * a]);
                    float v = g.x * m0[u] + g.y * m1[u] + g.z * m2[u];
                    if (c == 3) v += g.w;
                    L.pose.G[j][e] = v;
                }
                wave_lds_fence();
            }
        }
    }
    for (int i = lane; i < NJ * 3; i += 64) {          // (joint, row a)
        const int j = i / 3, aa = i - 3 * j;
        const float4 g = *reinterpret_cast<const float4*>(&L.pose.G[j][4 * aa]);
        const float t = g.w - (g.x * L.pose.J[j][0] + g.y * L.pose.J[j][1] + g.z * L.pose.J[j][2]);
        *reinterpret_cast<float4*>(&L.pose.A[j][4 * aa]) = make_float4(g.x, g.y, g.z, t);
    }
}

// The same chain by the whole workgroup in pointer-jumping form: P_j <- P_anc(j) . P_j with the ancestor
// distance doubling every step, so the longest path (9 joints in SMPL) is complete after 4 steps of one
// 3x4 product each instead of 9 dependent passes of one wave.  Thread = (joint, element) for tid < 288;
// the steps ping-pong between L.pose.G and L.gM (the adjoint's buffer, dead during the forward).  The
// products are associated differently from the sequential chain (last-bit differences in G); every consumer
// (vertex pass, objective, adjoint) reads the same G / A.  Ends with __syncthreads.
__device__ __forceinline__ void chain_forward_block(ClosureLds& L, int tid) {
    const bool act = tid < NJ * 12;
    const int j = act ? tid / 12 : 0, e = tid - 12 * j, a = e >> 2, c = e & 3;
    const int ns = L.M.n_jump;
    int anc[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) anc[s] = L.M.anc_tab[s][j];
    const float* src = &L.pose.Mj[0][0];
    float* dst = &L.pose.G[0][0];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        if (s < ns) {                                             // uniform
            if (act) {
                float v;
                if (anc[s] >= 0) {
                    const float4 x = *reinterpret_cast<const float4*>(src + anc[s] * 12 + 4 * a);
                    v = x.x * src[j * 12 + c] + x.y * src[j * 12 + 4 + c] + x.z * src[j * 12 + 8 + c];
                    if (c == 3) v += x.w;
                } else {
                    v = src[j * 12 + e];
                }
                dst[j * 12 + e] = v;
            }
            __syncthreads();
            src = dst;
            dst = (dst == &L.pose.G[0][0]) ? &L.gM[0][0] : &L.pose.G[0][0];
        }
    }
    // src holds G; A_j = [Gr_j | Gt_j - Gr_j J_j]  (lbs.py:365-368); G itself lands in L.pose.G
    if (tid < NJ * 3) {
        const int jj = tid / 3, aa = tid - 3 * jj;
        const float4 g = *reinterpret_cast<const float4*>(src + jj * 12 + 4 * aa);
        const float t = g.w - (g.x * L.pose.J[jj][0] + g.y * L.pose.J[jj][1] + g.z * L.pose.J[jj][2]);
        *reinterpret_cast<float4*>(&L.pose.A[jj][4 * aa]) = make_float4(g.x, g.y, g.z, t);
        if (src != &L.pose.G[0][0]) *reinterpret_cast<float4*>(&L.pose.G[jj][4 * aa]) = g;
    }
    __syncthreads();
}

// k-split partial sums of v_posed = v_template + coef . pd_sub  (lbs.py:179,192-203 on the selected
// vertices): 8 slices of 28 basis rows x nc_pad/4 float4 column groups = items; threads [0, nthreads)
// take items round-robin (one each for the SMPL keypoint set).  All 28 loads of an item are issued
// before the first FMA: the stream is then limited by bandwidth, not by one L2 round trip per row pair.
__device__ __forceinline__ int fwd_slices(int, int) { return FWD_SLICES; }

__device__ __forceinline__ void contraction_forward(const DevModel& M, ClosureLds& L, int t, int nthreads) {
    const int nc_pad = L.M.nc_pad;
    const int ncq = nc_pad >> 2;                   // float4 column groups
    for (int item = t; item < ncq * FWD_SLICES; item += nthreads) {
        // workgroups of one XCD run in near lock-step on the same matrix: start each at a different slice
        // so that they do not queue on the same L2 channel
        const int cq = item % ncq, ks = (item / ncq + (int)blockIdx.x) & (FWD_SLICES - 1);
        const float4* src = reinterpret_cast<const float4*>(M.pd_sub) + (size_t)(ks * FWD_RPS) * ncq + cq;
        float4 v[FWD_RPS];
#pragma unroll
        for (int r = 0; r < FWD_RPS; ++r) v[r] = src[(size_t)r * ncq];
        float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
#pragma unroll
        for (int r = 0; r < FWD_RPS; r += 2) {
            const float c0 = L.coef[ks * FWD_RPS + r], c1 = L.coef[ks * FWD_RPS + r + 1];
            acc0.x = fmaf(c0, v[r].x, acc0.x); acc0.y = fmaf(c0, v[r].y, acc0.y);
            acc0.z = fmaf(c0, v[r].z, acc0.z); acc0.w = fmaf(c0, v[r].w, acc0.w);
            acc1.x = fmaf(c1, v[r + 1].x, acc1.x); acc1.y = fmaf(c1, v[r + 1].y, acc1.y);
            acc1.z = fmaf(c1, v[r + 1].z, acc1.z); acc1.w = fmaf(c1, v[r + 1].w, acc1.w);
        }
        float4* dst = reinterpret_cast<float4*>(L.scratch + ks * nc_pad + 4 * cq);
        *dst = make_float4(acc0.x + acc1.x, acc0.y + acc1.y, acc0.z + acc1.z, acc0.w + acc1.w);
    }
}

// keypoint k = selection row . xs + transl  (body_models_scale.py:393-403): thread per (keypoint, coordinate).
// Ends with __syncthreads.
__device__ __forceinline__ void keypoints_from_xs(ClosureLds& L, int tid) {
    if (tid < NKP * 3) {
        const int k = tid / 3, a = tid - 3 * k;
        float acc = 0.f;
        if (L.M.padded) {
            int si[KP_NZ];
            float wi[KP_NZ], xv[KP_NZ];
#pragma unroll
            for (int t = 0; t < KP_NZ; ++t) { si[t] = L.M.kpp_s[k][t]; wi[t] = L.M.kpp_w[k][t]; }
#pragma unroll
            for (int t = 0; t < KP_NZ; ++t) xv[t] = L.xs[3 * si[t] + a];
#pragma unroll
            for (int t = 0; t < KP_NZ; ++t) acc = fmaf(wi[t], xv[t], acc);
        } else {
            for (int t = L.M.kp_start[k]; t < L.M.kp_start[k + 1]; ++t) acc = fmaf(L.M.kp_w[t], L.xs[3 * L.M.kp_s[t] + a], acc);
        }
        L.kp[k][a] = acc + L.opt.x[X_TR + a];
    }
    __syncthreads();
}

// E2 + E3: chain || forward contraction, then v_posed, T rows (lbs.py:209-213) and skinned positions.
// from_pass: L.vposed / L.xs of the selected vertices were written by the kernel prologue from the
// vertex pass's side outputs (full mode); run_chain = false when the pose block (G, A) is already in LDS.
__device__ __forceinline__ void sparse_forward(const DevModel& M, ClosureLds& L, bool from_pass, int tid, bool run_chain = true) {
    const int nc = L.M.nc, nc_pad = L.M.nc_pad;
    const int nks = fwd_slices(nc_pad, STEP_NT - 64);
    if (run_chain || !from_pass) {
        // wave 0: the kinematic chain; waves 1-7: the forward basis stream (k-split partial sums) - overlapped
        const long long t_e2 = PH_CLK();
        if (tid < 64) { if (run_chain) chain_forward_wave(L, tid); PH_T(22); }
        else if (!from_pass) { contraction_forward(M, L, tid - 64, STEP_NT - 64); PH_W(60, 64, t_e2); PH_W(62, 448, t_e2); }
        __syncthreads();
    }
    PH_T(1);
    // thread per coordinate c = 3 s + a: T row a of vertex s, v_posed[c], xs[c]
    if (tid < nc) {
        const int c = tid, s = c / 3, a = c - 3 * s;
        float4 tr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (L.M.sel_sparse) {
            // <= 4 non-zero weights per vertex (the SMPL family): the non-zero products of the dense row, ascending joint
            const float4 w4 = *reinterpret_cast<const float4*>(L.M.selw[s]);
            const unsigned jj = L.M.selj[s];
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
            float4 av[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) av[t] = *reinterpret_cast<const float4*>(&L.pose.A[(jj >> (8 * t)) & 255u][4 * a]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                tr.x = fmaf(wv[t], av[t].x, tr.x); tr.y = fmaf(wv[t], av[t].y, tr.y);
                tr.z = fmaf(wv[t], av[t].z, tr.z); tr.w = fmaf(wv[t], av[t].w, tr.w);
            }
        } else {
#pragma unroll 8
            for (int j = 0; j < NJ; ++j) {
                const float w = L.M.wT[j][s];
                const float4 av = *reinterpret_cast<const float4*>(&L.pose.A[j][4 * a]);
                tr.x = fmaf(w, av.x, tr.x); tr.y = fmaf(w, av.y, tr.y);
                tr.z = fmaf(w, av.z, tr.z); tr.w = fmaf(w, av.w, tr.w);
            }
        }
        *reinterpret_cast<float4*>(&L.T[s][4 * a]) = tr;
        if (!from_pass) {
            float vp[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                float v = L.M.vt_sub[3 * s + b];
                for (int k = 0; k < nks; ++k) v += L.scratch[k * nc_pad + 3 * s + b];
                vp[b] = v;
            }
            L.vposed[c] = vp[a];
            L.xs[c] = tr.x * vp[0] + tr.y * vp[1] + tr.z * vp[2] + tr.w;
        }
    }
    __syncthreads();
    keypoints_from_xs(L, tid);
}

// the scalar terms of the loss out of the reduced sums (fixed order, float64 like the reference's float() of the tensors):
// called by every thread (all get the total) or by one wave; `writer` stores the side results (L.total, L.loss_terms, the
// dropped-prior flags, the SDF factor, the selected mixture)
// SDFW (service launches of the single-launch fit only): S is the answer of the term's kernels to THIS round's request - the
// combining wave waits for the answer tag here, i.e. the keypoint phase (E4) and E5 run under the term's kernels; a wait that
// times out (200 ms) makes the loss NaN and is counted: the host fails the fit
template <bool SDFW = false>
__device__ __forceinline__ double loss_combine(const DevModel& M, ClosureLds& L, int V, const DevWeights& W, bool writer) {
    const bool use_vp = (W.flags & MVFIT_F_VPOSER) != 0;
    const bool use_gmm = !use_vp && (W.flags & MVFIT_F_PRIOR_GMM);
    const int ndw = (V * NKP + 63) >> 6;
    // ---- every thread: combine (fixed order) ----
    double l_data = 0.0;
    for (int w = 0; w < ndw; ++w) l_data += L.red_d[w];
    const bool use_3d = (W.flags & MVFIT_F_USE_3D) != 0;
    if (use_3d) l_data += L.red_d[STEP_NW];
    l_data *= (double)W.data_w2;                                               // fitting.py:311-316, 319-324
    const double sq = L.red_d[STEP_NW - 1], sqb = L.red_d[STEP_NW - 2], san = L.red_d[STEP_NW - 3];
    const double wp2 = (double)W.pose_w * (double)W.pose_w;
    double l_pose;
    int dropped = 0;
    if (use_vp) {
        l_pose = sq * wp2;                                                    // fitting.py:327-329
    } else {
        double P;
        if (use_gmm) {
            int sel = 0;
            float best = L.gmm_ll[0];
            for (int m = 1; m < M.gmm_M; ++m) if (L.gmm_ll[m] < best) { best = L.gmm_ll[m]; sel = m; }
            if (writer) L.gmm_sel = sel;
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
    double l_angle = san * (double)W.bend_w;
    if ((float)l_angle > 1e4f && !use_vp) { l_angle = 0.0; dropped |= 2; }     // fitting.py:349-350
    // interpenetration term (fitting.py:352-393): pen = (w S / valid_people)^2 with S from sdf_term.hip
    double l_coll = 0.0;
    float sdf_fac = 0.f;
    if (L.sdf_adj && W.coll_w > 0.f) {
        bool answered = true;
        if constexpr (SDFW) {
            if (L.sdf_wait_tag) {
                if (!L.sh_sdf_ok) answered = false;                  // (an earlier wait of this problem timed out: no more waiting)
                else {
                    const long long t0 = wall_clock64();
                    while (__hip_atomic_load(L.sdf_wait_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < L.sdf_wait_want) {
                        if (wall_clock64() - t0 > 20000000) { answered = false; break; }
                        __builtin_amdgcn_s_sleep(8);
                    }
                    if (!answered && writer) { atomicAdd(L.sdf_wait_stats, 1u); L.sh_sdf_ok = 0u; }
                }
            }
        }
        const float S = answered ? sdf_ld(&L.sdf_adj->S) : __builtin_nanf("");
        const double ws = (double)W.coll_w * (double)S;
        l_coll = ws * ws;
        sdf_fac = 2.f * W.coll_w * W.coll_w * S;
    }
    const double total = l_data + l_pose + l_shape + l_angle + l_coll;
    if (writer) {
        L.sdf_fac = sdf_fac;
        L.loss_terms[0] = (float)l_data; L.loss_terms[1] = (float)l_pose; L.loss_terms[2] = (float)l_shape;
        L.loss_terms[3] = (float)l_angle; L.loss_terms[4] = (float)l_coll; L.loss_terms[5] = (float)total;
        L.flags_dropped = dropped;
        L.total = total;
    }
    return total;
}

// ---------------------------------------------------------------------------------------------
// E4: SMPLifyLoss.forward (fitting.py:290-415, no SDF term) + gradient w.r.t. keypoints / priors.
// Returns the total loss (same value in every thread).  Ends with __syncthreads.
// ---------------------------------------------------------------------------------------------
template <bool DEFER = false>
__device__ __forceinline__ double loss_and_keypoint_grad(const DevModel& M, ClosureLds& L, int V, const DevWeights& W,
                                         bool want_grad, int tid) {
    const bool use_vp = (W.flags & MVFIT_F_VPOSER) != 0;
    const bool use_gmm = !use_vp && (W.flags & MVFIT_F_PRIOR_GMM);
    const int wave = tid >> 6;
    const int ndata = V * NKP;                 // <= 272: waves 0 .. 4
    const int ndw = (ndata + 63) >> 6;         // waves holding data-term threads
    double part = 0.0;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (tid < ndata) {
        const int v = tid / NKP, k = tid - v * NKP;
        const float X = L.kp[k][0], Y = L.kp[k][1], Z = L.kp[k][2];      // computed once per closure (keypoints_from_xs)
        const float* Rc = L.obs.camR[v];
        const float* tc = L.obs.camt[v];
        const float f = L.obs.camf[v];
        const float cx = L.obs.camc[v][0], cy = L.obs.camc[v][1];
        const float px = Rc[0] * X + Rc[1] * Y + Rc[2] * Z + tc[0];            // camera.py:106-110
        const float py = Rc[3] * X + Rc[4] * Y + Rc[5] * Z + tc[1];
        const float pz = Rc[6] * X + Rc[7] * Y + Rc[8] * Z + tc[2];
        const float ipz = 1.0f / pz;
        const float u = f * (px * ipz) + cx, w_ = f * (py * ipz) + cy;          // camera.py:112-116
        const float rx = L.obs.gt[tid * 2] - u, ry = L.obs.gt[tid * 2 + 1] - w_;
        const float wcf = L.obs.wc[tid];
        const float w2 = wcf * wcf;
        const float rx2 = rx * rx, ry2 = ry * ry;
        const float idx = 1.0f / (rx2 + W.rho2), idy = 1.0f / (ry2 + W.rho2);
        const float gmx = W.rho2 * (rx2 * idx), gmy = W.rho2 * (ry2 * idy);     // utils.py:435-438
        part = (double)(w2 * (gmx + gmy));
        if (want_grad) {
            const float gu = -w2 * W.data_w2 * (2.f * rx * W.rho2 * W.rho2 * (idx * idx));
            const float gv = -w2 * W.data_w2 * (2.f * ry * W.rho2 * W.rho2 * (idy * idy));
            const float gpx = f * gu * ipz, gpy = f * gv * ipz;
            const float gpz = -f * (gu * px + gv * py) * (ipz * ipz);
            g0 = Rc[0] * gpx + Rc[3] * gpy + Rc[6] * gpz;
            g1 = Rc[1] * gpx + Rc[4] * gpy + Rc[7] * gpz;
            g2 = Rc[2] * gpx + Rc[5] * gpy + Rc[8] * gpz;
            L.gkp_part[v][k][0] = g0; L.gkp_part[v][k][1] = g1; L.gkp_part[v][k][2] = g2;
        }
    }
    if (wave < ndw) {
        // whole-wave reductions (wave-uniform branch: every lane takes part, idle lanes add zeros)
        const double pd = wave64_sum(part);
        if ((tid & 63) == 0) L.red_d[wave] = pd;
        if (want_grad) {
            const float s0 = wave64_sum(g0), s1 = wave64_sum(g1), s2 = wave64_sum(g2);
            if ((tid & 63) == 0) { L.red_f[wave][0] = s0; L.red_f[wave][1] = s1; L.red_f[wave][2] = s2; }
        }
    } else if (wave == STEP_NW - 3 && (W.flags & MVFIT_F_USE_3D)) {
        // 3-D joint term (fitting.py:319-324): sum conf3d^2 GMoF(gt3d - joints) data_weight^2, lane = (keypoint, coord)
        const int lane = tid & 63;
        double p3 = 0.0;
        if (lane < NKP * 3) {
            const int k = lane / 3;
            const float r = L.obs.gt3d[lane] - (&L.kp[0][0])[lane];
            const float c2 = L.obs.c3d[k] * L.obs.c3d[k];
            const float id = 1.0f / (r * r + W.rho2);
            p3 = (double)(c2 * (W.rho2 * (r * r * id)));
            if (want_grad) (&L.gkp3[0][0])[lane] = -c2 * W.data_w2 * (2.f * r * W.rho2 * W.rho2 * (id * id));
        }
        const double s3 = wave64_sum(p3);
        if (lane == 0) L.red_d[STEP_NW] = s3;
    } else if (wave == STEP_NW - 1) {
        // priors on the last wave: |body_pose|^2 or |z|^2, |beta|^2, angle prior
        const int lane = tid & 63;
        double pp = 0.0, bb = 0.0, an = 0.0;
        if (use_vp) { if (lane < 32) { const double v = (double)L.opt.x[X_EMB + lane]; pp = v * v; } }
        else {
            { const double v = (double)L.pose.theta[3 + lane]; pp = v * v; }
            if (lane < 5) { const double v = (double)L.pose.theta[3 + 64 + lane]; pp += v * v; }
        }
        if (lane < 10) { const double v = (double)L.opt.x[X_BETAS + lane]; bb = v * v; }
        if (lane < 4) {
            // angle prior (prior.py:73-89): exp(pose[idx]*sgn)^2 on full_pose[3:66] idx 52,55,9,12
            const int idx = lane == 0 ? 52 : lane == 1 ? 55 : lane == 2 ? 9 : 12;
            const float sg = lane == 0 ? 1.f : -1.f;
            const float e0 = expf(sg * L.pose.theta[3 + idx]);
            an = (double)(e0 * e0);
        }
        const double spp = wave64_sum(pp), sbb = wave64_sum(bb), san = wave64_sum(an);
        if (lane == 0) { L.red_d[STEP_NW - 1] = spp; L.red_d[STEP_NW - 2] = sbb; L.red_d[STEP_NW - 3] = san; }
    }
    if (use_gmm) {
        // merged_log_likelihood (prior.py:181-196): min_m 0.5 d^T P_m d - log nll_w_m ; t_m = P_m d_m
        const int Mg = M.gmm_M;
        __syncthreads();
        // d_m = theta - mu_m (zero-padded to the 72-float rows) into the scratch, which is free between the forward and the
        // adjoint; then t_m = P_m d_m with 4 lanes per row (m, r): a lane takes the row's 16-byte words q, q + 4, ... and the
        // words of three passes of 128 rows are requested before the first is used (the 152 KB of the eight precision
        // matrices are a stream: 18 passes of five dependent loads each cost 8 us per closure)
        float* dm = L.scratch;
        for (int i = tid; i < Mg * 72; i += STEP_NT) {
            const int m = i / 72, c = i - 72 * m;
            dm[i] = c < 69 ? L.pose.theta[3 + c] - M.gmm_means[m * 69 + c] : 0.f;
        }
        __syncthreads();
        {
            const int row4 = tid >> 2, q = tid & 3, nrows = Mg * 69;
            constexpr int RPP = STEP_NT / 4;
            for (int p0 = 0; p0 * RPP < nrows; p0 += 3) {
                float4 w[3][5];
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) {
                    const int mr = min(row4 + (p0 + pp) * RPP, nrows - 1);
                    const float4* prow = reinterpret_cast<const float4*>(M.gmm_prec + (size_t)mr * 72);
#pragma unroll
                    for (int j = 0; j < 5; ++j) w[pp][j] = prow[min(q + 4 * j, 17)];
                }
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) {
                    const int mr0 = row4 + (p0 + pp) * RPP, mr = min(mr0, nrows - 1);
                    const int m = mr / 69, r = mr - m * 69;
                    const float4* dv = reinterpret_cast<const float4*>(dm + m * 72);
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        if (j == 4 && q >= 2) continue;                       // a row has 18 words
                        const float4 d4 = dv[q + 4 * j], w4 = w[pp][j];
                        acc = fmaf(w4.x, d4.x, acc); acc = fmaf(w4.y, d4.y, acc);
                        acc = fmaf(w4.z, d4.z, acc); acc = fmaf(w4.w, d4.w, acc);
                    }
                    acc += dpp_mov<DPP_XOR1>(acc);
                    acc += dpp_mov<DPP_XOR2>(acc);
                    if (q == 0 && mr0 < nrows) L.gmm_t[m][r] = acc;
                }
            }
        }
        __syncthreads();
        if (tid < 16 * Mg) {
            const int m = tid >> 4, l16 = tid & 15;
            const float* mu = M.gmm_means + m * 69;
            float qd = 0.f;
            for (int r = l16; r < 69; r += 16) qd = fmaf(L.gmm_t[m][r], L.pose.theta[3 + r] - mu[r], qd);
            qd = row16_sum(qd);
            if (l16 == 0) L.gmm_ll[m] = 0.5f * qd - M.gmm_lognw[m];
        }
    }
    __syncthreads();
    // the scalar terms are combined by every thread here (each then returns the total), or - DEFER - by one otherwise idle wave
    // inside the adjoint's first phase (closure_backward), which leaves the total in L.total
    double total = 0.0;
    if constexpr (!DEFER) total = loss_combine<false>(M, L, V, W, tid == 0);
    const bool use_3d = (W.flags & MVFIT_F_USE_3D) != 0;
    if (want_grad && tid < 3) {
        float s = 0.f;
        for (int w = 0; w < ndw; ++w) s += L.red_f[w][tid];
        if (use_3d) for (int k = 0; k < NKP; ++k) s += L.gkp3[k][tid];
        L.gtau[tid] = s;                                                       // g_tau = sum_k g_kp
    }
    if (want_grad && tid >= 64 && tid < 64 + NKP * 3) {
        // g_kp[k][a] = sum over views, ascending (independent loads; the adjoint's first phase reads it
        // after its own barrier)
        const int i = tid - 64;
        float s = 0.f;
#pragma unroll
        for (int vv = 0; vv < MVFIT_MAX_VIEWS; ++vv) if (vv < V) s += (&L.gkp_part[vv][0][0])[i];
        if (use_3d) s += (&L.gkp3[0][0])[i];
        (&L.gkp[0][0])[i] = s;
    }
    return total;
}

// chain adjoint by ONE wave, deepest parents first, following the host-built schedule:
// g_G_p[a][m] += sum_children sum_col g_G_c[a][col] M_c[m][col]  (m < 3),  g_G_p[a][3] += g_G_c[a][3].
__device__ __forceinline__ void chain_backward_wave(ClosureLds& L, int lane) {
    const int q = lane / 12, e = lane - 12 * q, a = e >> 2, m = e & 3;
    const int mr = m < 3 ? m : 0;
    const int npass = L.M.n_bwd;
    constexpr int CH = 4;           // passes per chunk (operands independent of the chain are preloaded)
    wave_lds_fence();
    for (int p0 = 0; p0 < npass; p0 += CH) {
        int w[CH];
        float4 k0[CH], k1[CH], k2[CH];
        float own[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) w[u] = (q < 5 && p0 + u < npass) ? L.M.bwd_tab[p0 + u][q] : -1;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            // absent entries / children point at the zero rows 24..31
            const int ww = w[u] != -1 ? w[u] : 0x1f1f1f00;
            const int c0 = (ww >> 8) & 0xff, c1 = (ww >> 16) & 0xff, c2 = (ww >> 24) & 0xff;
            k0[u] = *reinterpret_cast<const float4*>(&L.pose.Mj[c0][4 * mr]);
            k1[u] = *reinterpret_cast<const float4*>(&L.pose.Mj[c1][4 * mr]);
            k2[u] = *reinterpret_cast<const float4*>(&L.pose.Mj[c2][4 * mr]);
            own[u] = L.gG[ww & 0x1f][e];              // E6's value of the parent entry (see flag 0x80 below)
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (p0 + u < npass) {                                  // uniform
                if (w[u] != -1) {
                    const int p = w[u] & 0x1f, c0 = (w[u] >> 8) & 0xff, c1 = (w[u] >> 16) & 0xff, c2 = (w[u] >> 24) & 0xff;
                    const float4 g0 = *reinterpret_cast<const float4*>(&L.gG[c0][4 * a]);
                    const float4 g1 = *reinterpret_cast<const float4*>(&L.gG[c1][4 * a]);
                    const float4 g2 = *reinterpret_cast<const float4*>(&L.gG[c2][4 * a]);
                    float acc;
                    if (m < 3) {
                        acc = (g0.x * k0[u].x + g0.y * k0[u].y + g0.z * k0[u].z + g0.w * k0[u].w) +
                              (g1.x * k1[u].x + g1.y * k1[u].y + g1.z * k1[u].z + g1.w * k1[u].w) +
                              (g2.x * k2[u].x + g2.y * k2[u].y + g2.z * k2[u].z + g2.w * k2[u].w);
                    } else {
                        acc = g0.w + g1.w + g2.w;
                    }
                    // 0x80: a further entry of a parent with more than three children - add to the live value
                    const float base = (w[u] & 0x80) ? L.gG[p][e] : own[u];
                    L.gG[p][e] = base + acc;
                }
                wave_lds_fence();
            }
        }
    }
}

// g_coef partials = pd_subT . g_vposed: 8 column slices x 56 float4 row groups = 448 items, one per
// thread of waves 1-7; partials in L.scratch[cs][KROWS].  Loads first, as in the forward stream.
constexpr int BWD_SLICES = 8, BWD_CPS = NC_MAX / BWD_SLICES;     // up to 36 columns per slice
__device__ __forceinline__ int bwd_slices(int) { return BWD_SLICES; }

// CPS = compile-time bound of the columns per slice (the SMPL keypoint set: 28 of the array's 36 - the loop over the array's
// length with a run-time bound held 144 VGPRs for 112 used and the scheduler then issued the stream in pieces)
template <int CPS>
__device__ __forceinline__ void contraction_backward_t(const DevModel& M, ClosureLds& L, int t, int nthreads) {
    constexpr int npq = KROWS >> 2;                 // 56 float4 row groups
    const int nc_pad = L.M.nc_pad;
    const int cps = ((nc_pad >> 2) + BWD_SLICES - 1) / BWD_SLICES * 4;     // columns per slice (multiple of 4, <= CPS)
    for (int item = t; item < npq * BWD_SLICES; item += nthreads) {
        const int pq = item % npq, cs = (item / npq + (int)blockIdx.x) & (BWD_SLICES - 1);
        const int c0 = cs * cps;
        const float4* src = reinterpret_cast<const float4*>(M.pd_subT) + pq;
        float4 v[CPS];
#pragma unroll
        for (int r = 0; r < CPS; ++r) {
            const int c = min(c0 + r, nc_pad - 1);              // clamped: surplus columns get a zero coefficient
            if (CPS == BWD_CPS ? r < cps : true) v[r] = src[(size_t)c * npq];
        }
        float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
#pragma unroll
        for (int r = 0; r < CPS; r += 2) {
            if (r < cps) {
                // (columns past nc_pad: E5 stored zeros there - a conditional LDS read is a branch around the read, 28 of them
                // were 3 k cycles of the phase)
                const float g0 = L.gvp[c0 + r], g1 = L.gvp[c0 + r + 1];
                acc0.x = fmaf(g0, v[r].x, acc0.x); acc0.y = fmaf(g0, v[r].y, acc0.y);
                acc0.z = fmaf(g0, v[r].z, acc0.z); acc0.w = fmaf(g0, v[r].w, acc0.w);
                acc1.x = fmaf(g1, v[r + 1].x, acc1.x); acc1.y = fmaf(g1, v[r + 1].y, acc1.y);
                acc1.z = fmaf(g1, v[r + 1].z, acc1.z); acc1.w = fmaf(g1, v[r + 1].w, acc1.w);
            }
        }
        float4* dst = reinterpret_cast<float4*>(L.scratch + cs * KROWS + 4 * pq);
        *dst = make_float4(acc0.x + acc1.x, acc0.y + acc1.y, acc0.z + acc1.z, acc0.w + acc1.w);
    }
}

__device__ __forceinline__ void contraction_backward(const DevModel& M, ClosureLds& L, int t, int nthreads) {
    constexpr int SMPL_CPS = 28;
    if (((L.M.nc_pad >> 2) + BWD_SLICES - 1) / BWD_SLICES * 4 <= SMPL_CPS) contraction_backward_t<SMPL_CPS>(M, L, t, nthreads);      // uniform
    else contraction_backward_t<BWD_CPS>(M, L, t, nthreads);
}

// ---------------------------------------------------------------------------------------------
// Adjoint: g_kp -> grad[118]  (oracle/closure_np.py:_backward).
// Ends with __syncthreads; L.grad holds the flat gradient.
// ---------------------------------------------------------------------------------------------
// DEFER: the loss's scalar terms were left uncombined by loss_and_keypoint_grad<true>; the last wave - idle in E5 - combines
// them here, under E5 (400 cycles that every thread used to spend between the loss's barrier and this function's first one)
template <bool REMOTE = false, bool DEFER = false, bool SDFW = false>
__device__ __forceinline__ void closure_backward(const DevModel& M, ClosureLds& L, int V, const DevWeights& W, int tid) {
    const bool use_vp = (W.flags & MVFIT_F_VPOSER) != 0;
    const int ns = L.M.ns, nc = L.M.nc, nc_pad = L.M.nc_pad;
    __syncthreads();                  // L.gkp (view sums) and L.gtau are written after E4's last barrier
    // ---- E5: g_x = Ksel^T g_kp ; g_vposed = Tr^T g_x ----
    if constexpr (DEFER) {
        static_assert(NC_MAX <= STEP_NT - 64, "E5's threads leave the last wave free");
        if (tid >= STEP_NT - 64) loss_combine<SDFW>(M, L, V, W, tid == STEP_NT - 64);
    }
    if (tid < nc_pad) {
        float v = 0.f;
        if (tid < nc) {
            const int s = tid / 3, bq = tid - 3 * s;
            float gx0 = 0.f, gx1 = 0.f, gx2 = 0.f;
            // g_kp[k] = sum over views (ascending) of the per-view parts: L.gkp (summed once in E4b)
            if (L.M.padded) {
#pragma unroll
                for (int t = 0; t < VS_NZ; ++t) {
                    const int k = L.M.vsp_k[s][t];
                    const float w = L.M.vsp_w[s][t];
                    gx0 = fmaf(w, L.gkp[k][0], gx0); gx1 = fmaf(w, L.gkp[k][1], gx1); gx2 = fmaf(w, L.gkp[k][2], gx2);
                }
            } else {
                for (int t = L.M.vs_start[s]; t < L.M.vs_start[s + 1]; ++t) {
                    const int k = L.M.vs_k[t];
                    const float w = L.M.vs_w[t];
                    gx0 = fmaf(w, L.gkp[k][0], gx0); gx1 = fmaf(w, L.gkp[k][1], gx1); gx2 = fmaf(w, L.gkp[k][2], gx2);
                }
            }
            v = L.T[s][0 + bq] * gx0 + L.T[s][4 + bq] * gx1 + L.T[s][8 + bq] * gx2;
            L.gx[tid] = bq == 0 ? gx0 : (bq == 1 ? gx1 : gx2);
        }
        L.gvp[tid] = v;
    } else if (tid < NC_MAX) {
        L.gvp[tid] = 0.f;                     // surplus columns of the transposed contraction's slices (E7 reads them unconditionally)
    }
    __syncthreads();
    PH_T(4);
    const float sdf_fac = L.sdf_fac;       // (side results of the loss's combine: complete behind this barrier in either mode)
    // ---- E6: g_A = sum_s W[s][j] [g_x v_posed^T | g_x]: 16-lane row per joint, lanes stride s ----
    if (tid < NJ * 16) {
        const int j = tid >> 4, g = tid & 15;
        float acc[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) acc[e] = 0.f;
        for (int s = g; s < ns; s += 16) {
            const float w = L.M.wT[j][s];
            const float g0 = w * L.gx[3 * s], g1 = w * L.gx[3 * s + 1], g2 = w * L.gx[3 * s + 2];
            const float v0 = L.vposed[3 * s], v1 = L.vposed[3 * s + 1], v2 = L.vposed[3 * s + 2];
            acc[0] = fmaf(g0, v0, acc[0]); acc[1] = fmaf(g0, v1, acc[1]); acc[2] = fmaf(g0, v2, acc[2]);
            acc[3] = fmaf(g1, v0, acc[3]); acc[4] = fmaf(g1, v1, acc[4]); acc[5] = fmaf(g1, v2, acc[5]);
            acc[6] = fmaf(g2, v0, acc[6]); acc[7] = fmaf(g2, v1, acc[7]); acc[8] = fmaf(g2, v2, acc[8]);
            acc[9] += g0; acc[10] += g1; acc[11] += g2;
        }
#pragma unroll
        for (int e = 0; e < 12; ++e) acc[e] = row16_sum(acc[e]);
        if (g == 0) {
            if (sdf_fac != 0.f) {                          // + the SDF term's dense-vertex part of g_A
#pragma unroll
                for (int e = 0; e < 12; ++e) acc[e] = fmaf(sdf_fac, sdf_ld(&L.sdf_adj->gA[j * 12 + e]), acc[e]);
            }
            // A_j = [Gr_j | Gt_j - Gr_j J_j]:  g_Gt = g_At ; g_Gr = g_Ar - g_At J^T ; g_J = -Gr^T g_At
            const float J0 = L.pose.J[j][0], J1 = L.pose.J[j][1], J2 = L.pose.J[j][2];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                *reinterpret_cast<float4*>(&L.gG[j][4 * a]) =
                    make_float4(acc[3 * a + 0] - acc[9 + a] * J0, acc[3 * a + 1] - acc[9 + a] * J1,
                                acc[3 * a + 2] - acc[9 + a] * J2, acc[9 + a]);
                L.gJ[j][a] = -(L.pose.G[j][0 + a] * acc[9] + L.pose.G[j][4 + a] * acc[10] + L.pose.G[j][8 + a] * acc[11]);
            }
        }
    }
    __syncthreads();
    PH_T(5);
    // ---- E7: chain adjoint on wave 0 || transposed contraction on waves 1-7 ----
    // (measured: the two overlap - the one-wave walk up the tree, 5 k cycles, hides the 186 KB stream; the whole-workgroup
    // form of the adjoint, chain_backward_block, is shorter alone but runs behind the stream: +0.7-1.1 us per round)
    const long long t_e7 = PH_CLK();
    if (tid < 64) { chain_backward_wave(L, tid); PH_T(23); }
    else { contraction_backward(M, L, tid - 64, STEP_NT - 64); PH_W(59, 64, t_e7); PH_W(61, 448, t_e7); }
    const int ncs = bwd_slices(STEP_NT - 64);
    __syncthreads();
    PH_T(6);
    // ---- E8a: g_M = G_parent^T g_G (root: identity) ; g_R = g_Rm (+ g_coef) ----
    // g_M element (joint j with parent pa, row mm, column c) = column mm of G_pa . column c of g_G_j
    // g_M element (joint j with parent pa, row mm, column c) = column mm of G_pa . column c of g_G_j.  The value is made
    // opaque before anything is added to it: the sum then contracts into FMAs on its own, i.e. to the same bits wherever it
    // is used (inside a longer expression the surrounding additions would fuse with its products differently).
    auto gm_elem = [&](int pa, int j, int mm, int c) -> float {
        float v = L.pose.G[pa][mm] * L.gG[j][c] + L.pose.G[pa][4 + mm] * L.gG[j][4 + c] + L.pose.G[pa][8 + mm] * L.gG[j][8 + c];
        asm volatile("" : "+v"(v));
        return v;
    };
    constexpr int GJ_T0 = 320;                         // waves 5-6: idle in this phase (the g_M threads are waves 0-4)
    static_assert(GJ_T0 >= NJ * 12 && GJ_T0 % 64 == 0 && GJ_T0 + NJ * 3 <= STEP_NT, "g_J totals on otherwise idle waves");
    if (tid >= GJ_T0 && tid < GJ_T0 + NJ * 3) {
        // total g_J[j][a] = E6's part + the translation column of the joint's own g_M - its children's (lbs.py:341-348: a
        // child's relative translation is J_child - J_parent), with the g_M elements taken straight from g_G (the same
        // products as the threads that store g_M): E9's g_beta is then a plain 72-long product per shape coefficient instead
        // of a walk over child lists behind the next barrier.  Staged in L.gvp (dead since E7).  The first three children
        // are read branch-free (an absent one = a zero row of g_G: subtracting +0 changes nothing), further ones in a loop.
        const int i = tid - GJ_T0, j = i / 3, a = i - 3 * j;
        const int c_lo = L.M.child_start[j], c_hi = L.M.child_start[j + 1];
        int ch[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) ch[k] = c_lo + k < c_hi ? L.M.child_list[min(c_lo + k, NJ - 1)] : NJ;      // (g_G rows 24..31 are zero)
        float gj = L.gJ[j][a] + (j == 0 ? L.gG[0][4 * a + 3] : gm_elem(L.M.parents[j], j, a, 3));
        float sub[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) sub[k] = gm_elem(j, ch[k], a, 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) gj -= sub[k];
        for (int ci = c_lo + 3; ci < c_hi; ++ci) gj -= gm_elem(j, L.M.child_list[ci], a, 3);
        L.gvp[i] = gj;
    }
    if (tid < NJ * 12) {
        const int j = tid / 12, e = tid - 12 * j, mm = e >> 2, c = e & 3;
        float v;
        if (j == 0) v = L.gG[0][e];
        else {
            const int pa = L.M.parents[j];
            v = L.pose.G[pa][mm] * L.gG[j][c] + L.pose.G[pa][4 + mm] * L.gG[j][4 + c] + L.pose.G[pa][8 + mm] * L.gG[j][8 + c];
        }
        L.gM[j][e] = v;
        if (c < 3) {
            float gr;
            if (j == 0) gr = L.opt.x[X_SC] * v;
            else {
                float gc = 0.f;
                for (int k = 0; k < ncs; ++k) gc += L.scratch[k * KROWS + 9 * (j - 1) + 3 * mm + c];
                if (sdf_fac != 0.f) gc = fmaf(sdf_fac, sdf_ld(&L.sdf_adj->gcoef[9 * (j - 1) + 3 * mm + c]), gc);
                gr = v + gc;
            }
            L.gR[j][3 * mm + c] = gr;
        }
    }
    if (!use_vp && (W.flags & MVFIT_F_PRIOR_GMM) && !(L.flags_dropped & 1)) {
        // 0.5 (P d + P^T d) of the selected mixture: P d is gmm_t[m]; P^T d via the transposed copy
        const int m = L.gmm_sel;
        // d of the selected mixture is staged in a row of gmm_t that is dead by now (another mixture's t)
        float* dsel = L.gmm_t[m ^ 1];
        if (tid < 72) dsel[tid] = tid < 69 ? L.pose.theta[3 + tid] - M.gmm_means[m * 69 + tid] : 0.f;
        __syncthreads();
        const int row4 = tid >> 2, q = tid & 3;
        {
            const int i = min(row4, 68);
            const float4* prow = reinterpret_cast<const float4*>(M.gmm_precT + ((size_t)m * 69 + i) * 72);
            float4 w[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) w[j] = prow[min(q + 4 * j, 17)];
            const float4* dv = reinterpret_cast<const float4*>(dsel);
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                if (j == 4 && q >= 2) continue;
                const float4 d4 = dv[q + 4 * j], w4 = w[j];
                acc = fmaf(w4.x, d4.x, acc); acc = fmaf(w4.y, d4.y, acc);
                acc = fmaf(w4.z, d4.z, acc); acc = fmaf(w4.w, d4.w, acc);
            }
            acc += dpp_mov<DPP_XOR1>(acc);
            acc += dpp_mov<DPP_XOR2>(acc);
            if (q == 0 && row4 < 69) L.gmm_d[row4] = 0.5f * (L.gmm_t[m][row4] + acc);
        }
    }
    __syncthreads();
    PH_T(7);
    // ---- E8b || E9: g_scale ; g_J -> g_beta ; Rodrigues adjoint + pose priors ----
    const float wp2 = W.pose_w * W.pose_w;
    const long long t_e9 = PH_CLK();
    if (tid >= 256 && tid < 256 + 160) {
        // g_beta[l] = g_coef[207 + l] + sum_i J_S[i][l] g_J[i]  (+ shape prior): 16 lanes per l
        const int l = (tid - 256) >> 4, g = tid & 15;
        float s = 0.f;
        for (int i = g; i < NJ * 3; i += 16) s = fmaf(L.M.J_S[i][l], L.gvp[i], s);      // (g_J totals: E8a)
        s = row16_sum(s);
        if (g == 0) {
            float gc = 0.f;
            for (int k = 0; k < ncs; ++k) gc += L.scratch[k * KROWS + 207 + l];
            if (sdf_fac != 0.f) gc = fmaf(sdf_fac, sdf_ld(&L.sdf_adj->gcoef[207 + l]), gc);
            s += gc;
            if (!(W.flags & MVFIT_F_FIX_SHAPE)) s += 2.f * L.opt.x[X_BETAS + l] * W.shape_w * W.shape_w;
            L.gbeta[l] = s;
            L.grad[X_BETAS + l] = (W.flags & MVFIT_F_FIX_SHAPE) ? 0.f : s;
        }
        PH_W(56, 256, t_e9);
    } else if (tid == 448) {
        float s = 0.f;
        {
            // products rounded one by one, then added in order (what the compiler's packed multiplies made of this sum when it
            // stood alone; said explicitly, it no longer depends on the code around it)
#pragma clang fp contract(off)
            for (int e = 0; e < 9; ++e) s += L.gM[0][4 * (e / 3) + (e % 3)] * L.pose.R[0][e];
        }
        L.gscale = s;
        L.grad[X_SC] = (W.flags & MVFIT_F_FIX_SCALE) ? 0.f : s;
    } else if (tid > 448 && tid < 449 + (DPAD - X_TR - 1)) {
        // the slots of the flat gradient nobody else writes: translation (+ the SDF term's part), the embedding's own prior
        // (fitting.py:328, d/dz |z|^2 w^2; the decoder's adjoint adds to it behind the barrier), zero padding
        const int i = X_TR + (tid - 449) + (tid - 449 >= 3 ? 1 : 0);           // X_TR .. X_TR + 2, X_EMB .. DPAD - 1
        float g = 0.f;
        if (i < X_SC) { g = L.gtau[i - X_TR]; if (sdf_fac != 0.f) g = fmaf(sdf_fac, sdf_ld(&L.sdf_adj->gtau[i - X_TR]), g); }
        else if (i < DV) g = use_vp ? 2.f * L.opt.x[i] * wp2 : 0.f;
        L.grad[i] = g;
    } else if (tid < NJ) {
        const float rx = L.pose.theta[3 * tid], ry = L.pose.theta[3 * tid + 1], rz = L.pose.theta[3 * tid + 2];
        const float ex = rx + 1e-8f, ey = ry + 1e-8f, ez = rz + 1e-8f;
        const float a = L.pose.rod[tid][0], sn = L.pose.rod[tid][1], cs = L.pose.rod[tid][2];
        const float ia = 1.0f / a;
        const float kx = rx * ia, ky = ry * ia, kz = rz * ia;
        const float oc = 1.f - cs;
        const float K[9] = {0.f, -kz, ky, kz, 0.f, -kx, -ky, kx, 0.f};
        float KK[9];
        mat3_mul(K, K, KK);
        float g[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) g[e] = L.gR[tid][e];
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
        ga -= (gkx * rx + gky * ry + gkz * rz) * (ia * ia);
        float gth[3] = {(gkx + ga * ex) * ia, (gky + ga * ey) * ia, (gkz + ga * ez) * ia};
        if (tid > 0 && !use_vp) {
            // priors on body_pose (fitting.py:330-337)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int i = 3 * (tid - 1) + q;               // body_pose index
                const float bp = L.pose.theta[3 + i];
                float gq = gth[q];
                if (W.flags & MVFIT_F_PRIOR_GMM) { if (!(L.flags_dropped & 1)) gq += L.gmm_d[i] * wp2; }
                else if (!(L.flags_dropped & 1)) gq += 2.f * bp * wp2;
                gq += 2.f * bp * 16.f * wp2;
                gth[q] = gq;
            }
        }
        if (tid > 0 && !(L.flags_dropped & 2)) {
            // angle prior gradient on full_pose[3:66] idx 52,55,9,12
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int i = 3 * (tid - 1) + q;
                if (i == 52 || i == 55 || i == 9 || i == 12) {
                    const float sg = (i == 52) ? 1.f : -1.f;
                    gth[q] += 2.f * expf(2.f * L.pose.theta[3 + i] * sg) * sg * W.bend_w;
                }
            }
        }
        L.gtheta[3 * tid] = gth[0]; L.gtheta[3 * tid + 1] = gth[1]; L.gtheta[3 * tid + 2] = gth[2];
        // the flat gradient is written where its parts are made (global_orient | body_pose are contiguous; with VPoser the
        // body pose is not a parameter: zeros)
#pragma unroll
        for (int q = 0; q < 3; ++q) L.grad[X_GO + 3 * tid + q] = (tid > 0 && use_vp) ? 0.f : gth[q];
        PH_W(57, 0, t_e9);
    }
    __syncthreads();
    PH_W(58, 0, t_e9);
    if (use_vp) vposer_backward<REMOTE>(M, L, tid);
}

// write the operands of the vertex pass for problem b
__device__ __forceinline__ void publish_pose(const ClosureLds& L, const DevPose& P, int b, int tid) {
    float* ct = P.coefT + (size_t)(b >> 5) * KROWS * 32 + (b & 31);
    _Float16* ch = reinterpret_cast<_Float16*>(P.coefH) + (size_t)(b >> 5) * (KROWS / 16) * 2 * 64 * 8;
    for (int p = tid; p < KROWS; p += STEP_NT) {
        const float cv = L.coef[p];
        ct[p * 32] = cv;
        // split-fp16 A operand of the vertex pass: block p / 16, lane = 32 * ((p % 16) / 8) + problem, element p % 8
        const _Float16 hi = (_Float16)cv;
        const _Float16 lo = (_Float16)(cv - (float)hi);
        const size_t at = ((size_t)(p >> 4) * 2 * 64 + (size_t)(((p >> 3) & 1) * 32 + (b & 31))) * 8 + (p & 7);
        ch[at] = hi;
        ch[at + 64 * 8] = lo;
    }
    for (int i = tid; i < NJ * 12; i += STEP_NT) P.Amat[(size_t)b * 288 + i] = (&L.pose.A[0][0])[i];
    if (tid < 3) P.tau[(size_t)b * 4 + tid] = L.opt.x[X_TR + tid];
}

// ---------------------------------------------------------------------------------------------------------
// Asynchronous fit: hand the vertex-pass operands of THIS trial point to the pass kernel that is already queued on
// the other CUs (another XCD's L2 is not coherent with ours): 16-byte write-through (sc1) stores - 56 words of
// split-fp16 coefficients in MFMA A-operand order, 72 words of skinning transforms, 1 word of translation - and
// later, once every storing wave has drained them (publish_tag), one relaxed agent-scope store of the problem's tag.
// ---------------------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store16_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, const float4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (int)byte_off, 0, /*aux = sc1*/ 16);
}

constexpr int PUBLISH_WAVE = 3;      // idle in the loss phase that follows (data term: waves 0-1 at <= 7 views; 3-D term and priors: waves 5-7)

// b = the workgroup's ring row (sub-batch-relative); L.sh_prob = the global index of the problem it is fitting (the passes write the
// round's vertices there: rows take new problems when theirs has finished; read from LDS by the one lane that needs it - as a
// register it was live through the whole round loop: 8 more spilled registers in the kernels without a queue)
__device__ __forceinline__ void publish_pose_async(ClosureLds& L, const AsyncRing& R, int slot, unsigned round, int b, int tid) {
    // All 129 words go out from ONE wave, which also stores the tag later: the hand-off needs no workgroup barrier.
    const int l = tid - 64 * PUBLISH_WAVE;
    if (l < 0 || l >= 64) return;
    // Back-pressure: slot r % nslots still holds the operands of round r - nslots until that round's pass has run.  The
    // gate kernels publish how many rounds' passes are complete; the value is cached in LDS and only re-read when it does
    // not cover this round (it grows by ~nslots between two reads when the passes keep up: a few polls per fit).  Only this
    // wave waits - it idles through the loss phase anyway; when the passes are slower than the optimiser the whole
    // workgroup ends up waiting for it at the next barrier, i.e. the optimiser runs at the passes' rate.  Bounded by the
    // wall clock (20 ms): a stuck pass stream cannot hang the fit (the give-up is counted in stats[3]; a resident workgroup that
    // finds its slot overwritten skips the round and counts the lost operand sets in stats[2]).
    if (round >= (unsigned)R.nslots) {
        const unsigned need = round - (unsigned)R.nslots + 1u;
        unsigned have = L.sh_pass_done;
        if (have < need) {
            const long long t0 = wall_clock64();
            for (;;) {
                // the minimum over the consumers' words: one (the gate kernels of the per-round launches) or one per
                // workgroup of the resident pass
                have = 0xffffffffu;
                for (int i = l; i < R.npass; i += 64)
                    have = min(have, __hip_atomic_load(R.pass_done + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                for (int o = 32; o; o >>= 1) have = min(have, (unsigned)__shfl_xor((int)have, o));
                if (have >= need) break;
                // timed out: stop waiting for the passes for the rest of this fit (the overwritten slots are counted as
                // missed by the pass and fit() reports them) instead of stalling 20 ms in every further round
                if (wall_clock64() - t0 > 2000000) {
                    have = 0xffffffffu;
                    if (l == 0) atomicAdd(R.stats + 3, 1u);       // never silent: this problem gave up on the back-pressure (mvfit_fit_stats)
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
            if (l == 0) L.sh_pass_done = have;
        }
    }
    const unsigned Bp = (unsigned)R.Bpad;
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(R.coefH, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(R.Amat, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(R.tau, 0, 0x7fffffff, 0x00020000);
    {   // 72 words of skinning transforms: lanes 0..63, then 0..7
        const unsigned base = ((unsigned)slot * Bp + (unsigned)b) * 72;
        store16_sc1(rs_a, (base + l) * 16, reinterpret_cast<const float4*>(&L.pose.A[0][0])[l]);
        if (l < 8) store16_sc1(rs_a, (base + 64 + l) * 16, reinterpret_cast<const float4*>(&L.pose.A[0][0])[64 + l]);
    }
    if (l < 56) {
        // word w: block G = w / 4, (hi | lo) = (w / 2) & 1, row half h = w & 1: coefficients p = 16 G + 8 h + t
        const int G = l >> 2, hl = (l >> 1) & 1, h = l & 1;
        _Float16 q[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float cv = L.coef[16 * G + 8 * h + t];
            const _Float16 hi = (_Float16)cv;
            q[t] = hl ? (_Float16)(cv - (float)hi) : hi;
        }
        float4 v;
        __builtin_memcpy(&v, q, 16);
        const unsigned chunk = (unsigned)b >> 5;
        const unsigned word = (((unsigned)slot * (Bp >> 5) + chunk) * (KROWS / 16) + G) * 2 + hl;
        store16_sc1(rs_c, (word * 64 + 32 * h + ((unsigned)b & 31)) * 16, v);
    } else if (l == 56) {
        store16_sc1(rs_t, ((unsigned)slot * Bp + (unsigned)b) * 16, make_float4(L.opt.x[X_TR], L.opt.x[X_TR + 1], L.opt.x[X_TR + 2], __builtin_bit_cast(float, L.sh_prob)));
    }
}

// The tag goes out from the same wave once its stores have drained (free by now: they were issued before the loss and
// the whole adjoint, whose own load waits already covered them); ONE lane, relaxed at agent scope.
__device__ __forceinline__ void publish_tag(const AsyncRing& R, int slot, int b, unsigned round, int tid) {
    if ((tid >> 6) == PUBLISH_WAVE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((tid & 63) == 0) __hip_atomic_store(R.tag + (size_t)slot * R.Bpad + b, round + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------------------
// SDF term as a service beside the single-launch fit (round 6; fitting.py:352-393).  In a stage with coll_loss_weight > 0 the
// loss needs S = sum of phi over ALL 6890 vertices of the trial point and its adjoint (SdfAdj).  The optimiser kernel stays
// resident (state and history in LDS, compact direction) and asks for them: in such a round the publishing wave writes, on top
// of the ring operands, the float32 coefficients the pull-back contracts with (the chained rounds' coefT layout) and the
// problem's gate word, drains its stores and publishes the tag AT ONCE; the host has queued gate -> vertex pass -> SDF front
// -> pull-back for the round on the pass stream; the pull-back's reducing workgroup publishes the answer tag behind its
// write-through result; the workgroup goes on with the keypoint phase and waits for the answer where S is first needed
// (loss_combine<true>, the wave that combines the loss's scalar terms under E5).  Per round: one memory hop to the gate kernel, three
// launches, one hop back - no optimiser state reload, no step-kernel launch, the forward of the trial point already done.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void publish_sdf_request(const ClosureLds& L, const DevPose& P, int* gate, int b, int on, int tid) {
    const int l = tid - 64 * PUBLISH_WAVE;
    if (l < 0 || l >= 64) return;
    if (on) {
        float* ct = P.coefT + (size_t)(b >> 5) * KROWS * 32 + (b & 31);
        for (int p = l; p < KROWS; p += 64) __hip_atomic_store(ct + p * 32, L.coef[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (l == 0) __hip_atomic_store(gate + b, on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace mvfit

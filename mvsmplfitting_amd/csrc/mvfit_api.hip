// libmvfit: C ABI (include/mvfit.h) + the per-problem step kernels.
//
// Kernels in this file (one workgroup per problem, see closure_device.h / lbfgs_device.h):
//   prep_kernel        params -> pose operands of the vertex pass
//   closure_kernel     one closure evaluation (loss, grad, keypoints) - the drop-in closure
//   fit_step_kernel    one closure round of the device-resident fit: objective + adjoint from the
//                      vertex-pass output, L-BFGS state-machine advance, pose operands of the next
//                      trial point
//   fit_sparse_kernel  the whole fit of one problem in a single launch (objective restricted to the
//                      vertices it reads; no vertex pass inside the loop)
//   lbfgs_kat_kernel   float64 instantiation of the state machine on analytic objectives
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "closure_device.h"

namespace mvfit {

hipError_t launch_vertex_pass(const DevModel& M, const DevPose& P, int B, float* verts, int ksplit,
                              hipStream_t stream, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
hipError_t vertex_pass_configure();
hipError_t launch_pass_gate(const DevPose& P, int b_lo, int B, hipStream_t stream);
hipError_t launch_vertex_pass_resident(const DevModel& M, const ResidentArgs& RA, int tpw, hipStream_t stream);
hipError_t launch_sdf_term(const DevModel& M, const DevPose& P, const float* verts, int B, const int32_t* faces, int num_faces,
                           int G, const int* gate, SdfBox* box, float4* samp, void* entries, SdfAdj* adj, hipStream_t stream,
                           void* cull, unsigned* answer_tag = nullptr, unsigned answer = 0u, const unsigned long long* box_parts = nullptr);
size_t sdf_cull_bytes(int B, int num_faces);
size_t sdf_op_ws_bytes(int B, int num_faces);
bool sdf_op_uses_lists(int num_faces);
hipError_t launch_sdf_voxelize_culled(const int32_t* faces, int num_faces, const float* vertices, int B, int num_vertices, int G,
                                      float* phi, void* ws, hipStream_t stream);
size_t sdf_cull_zero_offset(int B, int num_faces);
size_t sdf_cull_zero_bytes(int B);
int sdf_cull_min_faces();
size_t sdf_work_bytes(int B, int nv);
size_t sdf_ticket_offset(int B, int nv);
hipError_t launch_triangulate(const float* kps, const double* intris, const double* extris, int B, int V, int J, double* out,
                              hipStream_t stream);
hipError_t launch_depth_guess(const double* rest, const double* extri, const double* intri, const float* kps, int B, int J,
                              double* out, hipStream_t stream);
hipError_t launch_umeyama(const double* src, const double* dst, int B, int npts, int estimate_scale, double* rot, double* rvec,
                          double* trans, double* scale, hipStream_t stream);
hipError_t launch_project_points(const DevProblems& Q, const float* pts, int N, float* uv, hipStream_t stream);
hipError_t launch_sdf_voxelize(const int32_t* faces, int num_faces, const float* vertices, int B, int num_vertices, int G,
                               float* phi, hipStream_t stream);

struct StageWeights { DevWeights w[MVFIT_MAX_STAGES]; };

// per-problem optimiser storage in HBM
struct FitBuffers {
    OptBlock* opt;       // [B] trial point + L-BFGS scalars / working vectors / ro (LDS image block)
    PoseBlock* pose;     // [B] pose state of the current trial point (handed from launch to launch)
    float* dirs;         // [B][100][LB_D]
    float* stps;         // [B][100][LB_D]
    float* grow;         // [B][LB_GSIZE] pre-scaled Gram matrices (lbfgs_device.h:LbHist)
    float* gcol;         // [B][LB_GSIZE]
    float* rinv;         // [B][LB_RPACK] packed R^-1 of the compact direction form: the single-launch fit keeps it in LDS and parks
                         // it here only when a launch ends at its round cap
    double* stage_final; // [B][MVFIT_MAX_STAGES] run_fitting's return value per stage
    int* n_done;         // [3]: problems finished | problems of the current sub-batch that left the asynchronous phase (finished or
                         // paused at a stage boundary) | the same, all sub-batches of the fit
    VpBlock* vp;             // [B] VPoser decoder state of the current trial point (handed from launch to launch)
    const SdfAdj* sdf_adj;   // SDF term per problem (null: term not configured)
    int* sdf_gate;           // [B] 1 while the problem's current stage has coll_loss_weight > 0 and it is not done
    unsigned* sdf_tag;       // [B] service rounds of the single-launch fit: answer tag (round + 1) written behind the SdfAdj
    float* trace;            // [B][trace_cap][DV + 1] (x_trial, loss) of the first closures of a fit (mvfit_fit_trace); may be null
    int trace_cap;
};

// compact optimiser index (reference final_params order, non_linear_solver.py:164-170) -> flat x slot
__device__ __forceinline__ int cmap(int i, bool use_vp) {
    if (!use_vp) return i;                         // betas go body_pose transl scale = x[0:86]
    return i < 13 ? i : (i < 17 ? X_TR + (i - 13) : X_EMB + (i - 17));   // betas go transl scale embedding
}
__device__ __forceinline__ int dact(bool use_vp) { return use_vp ? 49 : 86; }

// pose operands of the vertex pass only: E1 + chain
template <bool CALL = false>
__device__ __forceinline__ void pose_and_chain(const DevModel& M, ClosureLds& L, uint32_t flags, int tid) {
    pose_prep<CALL>(M, L, flags, tid);
    chain_forward_block(L, tid);
}

__device__ __forceinline__ void store_block16(void* dst_g, const void* src_l, int nbytes, int tid) {
    const int n = nbytes / 16;
    for (int i = tid; i < n; i += STEP_NT) reinterpret_cast<float4*>(dst_g)[i] = reinterpret_cast<const float4*>(src_l)[i];
}

// per-problem observations -> ObsBlock image (one launch per mvfit_set_problems)
__global__ void pack_obs_kernel(DevProblems Q, ObsBlock* __restrict__ obs) {
    const int b = blockIdx.x, V = Q.V;
    const size_t cb = Q.cam_batched ? (size_t)b * V : 0;
    ObsBlock& O = obs[b];
    for (int i = threadIdx.x; i < (int)(sizeof(ObsBlock) / 4); i += blockDim.x) reinterpret_cast<float*>(&O)[i] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < V * 9; i += blockDim.x) (&O.camR[0][0])[i] = Q.cam_R[cb * 9 + i];
    for (int i = threadIdx.x; i < V * 3; i += blockDim.x) (&O.camt[0][0])[i] = Q.cam_t[cb * 3 + i];
    for (int i = threadIdx.x; i < V; i += blockDim.x) O.camf[i] = Q.cam_f[cb + i];
    for (int i = threadIdx.x; i < V * 2; i += blockDim.x) (&O.camc[0][0])[i] = Q.cam_c[cb * 2 + i];
    for (int i = threadIdx.x; i < V * NKP * 2; i += blockDim.x) O.gt[i] = Q.gt_xy[(size_t)b * V * NKP * 2 + i];
    for (int i = threadIdx.x; i < V * NKP; i += blockDim.x) O.wc[i] = Q.w_conf[(size_t)b * V * NKP + i];
}

__global__ void pack_joints3d_kernel(const float* __restrict__ gt3d, const float* __restrict__ conf3d,
                                     ObsBlock* __restrict__ obs) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < NKP * 3; i += blockDim.x) obs[b].gt3d[i] = gt3d[(size_t)b * NKP * 3 + i];
    for (int i = threadIdx.x; i < NKP; i += blockDim.x) obs[b].c3d[i] = conf3d[(size_t)b * NKP + i];
}

__global__ __launch_bounds__(STEP_NT) void prep_kernel(DevModel M, const ObsBlock* __restrict__ obs, DevPose P,
                                                       const float* __restrict__ params, uint32_t flags,
                                                       float* __restrict__ full_pose = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    ClosureLds& L = *reinterpret_cast<ClosureLds*>(smem_raw);
    const int b = blockIdx.x, tid = threadIdx.x;
    prologue(L, M, obs + b, nullptr, nullptr, nullptr, nullptr, params + (size_t)b * DV, tid);
    __syncthreads();
    pose_and_chain(M, L, flags, tid);
    publish_pose(L, P, b, tid);
    // ModelOutput.full_pose (body_models_scale.py:392-412): global_orient | body_pose, the latter decoded from the
    // embedding with MVFIT_F_VPOSER (fitting.py:170-173)
    if (full_pose && tid < 72) full_pose[(size_t)b * 72 + tid] = L.pose.theta[tid];
}

// REMOTE (test route, MVFIT_CLOSURE_VP_HELPERS=1): the launch carries VPoser decoder helpers behind the problems'
// workgroups and the closure decodes through them - the decoder arithmetic of the production single-launch fit
// (vposer_service.h) under the closure-level goldens; the pose operands of the trial point are published for the
// vertex pass that follows (like the asynchronous fit: objective from its own vertices, full pass beside it).
template <bool REMOTE>
__global__ __launch_bounds__(STEP_NT) void closure_kernel(DevModel M, const ObsBlock* __restrict__ obs, int nviews,
                                                          DevWeights W, DevPose P, const float* __restrict__ params,
                                                          int from_pass, float* __restrict__ loss,
                                                          float* __restrict__ grad, float* __restrict__ joints,
                                                          const SdfAdj* __restrict__ sdf_adj) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if (REMOTE && (int)blockIdx.x >= M.vps.nprob) {
        vposer_helper(M.vpt, M.vps, smem_raw, (int)blockIdx.x % M.vps.nsets, ((int)blockIdx.x - M.vps.nprob) / M.vps.nsets);
        return;
    }
    ClosureLds& L = *reinterpret_cast<ClosureLds*>(smem_raw);
    const int b = blockIdx.x, tid = threadIdx.x;
    prologue(L, M, obs + b, nullptr, nullptr, from_pass ? P.vposed_sel + (size_t)b * NC_MAX : nullptr,
             from_pass ? P.xs_sel + (size_t)b * NC_MAX : nullptr, params + (size_t)b * DV, tid, sdf_adj ? sdf_adj + b : nullptr);
    __syncthreads();
    if constexpr (REMOTE) {
        pose_prep_decode_inl<true>(M, L, W.flags, tid);
        pose_prep_elems(M, L, W.flags, tid);
    } else {
        pose_prep(M, L, W.flags, tid);
    }
    sparse_forward(M, L, from_pass != 0, tid);
    if constexpr (REMOTE) publish_pose(L, P, b, tid);
    const bool want_grad = grad != nullptr;
    const double total = loss_and_keypoint_grad(M, L, nviews, W, want_grad, tid);
    if (tid == 0 && loss) loss[b] = (float)total;
    if (joints && tid < NKP * 3) joints[(size_t)b * NKP * 3 + tid] = (&L.kp[0][0])[tid];
    if (want_grad) {
        closure_backward<REMOTE>(M, L, nviews, W, tid);
        if (tid < DV) grad[(size_t)b * DV + tid] = L.grad[tid];
    }
    if constexpr (REMOTE) {
        __syncthreads();
        if (tid == 0 && L.vp_remote) vps_store(vps_request_slot(M.vps), 0.f, (L.vp_seq + 1u) << 2 | VPS_BYE);
    }
}

// keypoints only (mvfit_vertices): gather from the vertex buffer
__global__ __launch_bounds__(64) void joints_kernel(DevModel M, const float* __restrict__ verts,
                                                    float* __restrict__ joints) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const ModelLds& C = *M.mlds;
    if (tid < NKP * 3) {
        const int k = tid / 3, a = tid - 3 * k;
        float s = 0.f;
        for (int t = C.kp_start[k]; t < C.kp_start[k + 1]; ++t)
            s = fmaf(C.kp_w[t], verts[((size_t)b * M.nv + C.sel_v[C.kp_s[t]]) * 3 + a], s);
        joints[(size_t)b * NKP * 3 + tid] = s;      // rows of the selection sum to 1 (+transl already in verts)
    }
}

__device__ __forceinline__ void opts_in(ClosureLds& L, const StageWeights& SW, const LbOpts& O, int tid) {
    constexpr int nsw = sizeof(StageWeights) / 4, nop = sizeof(LbOpts) / 4;
    if (tid < nsw) reinterpret_cast<int*>(&L.sw[0])[tid] = reinterpret_cast<const int*>(&SW)[tid];
    if (tid >= 128 && tid < 128 + nop) reinterpret_cast<int*>(&L.opts)[tid - 128] = reinterpret_cast<const int*>(&O)[tid - 128];
}

// initialise the optimiser state of every problem: x = params, first trial point = x
__global__ __launch_bounds__(STEP_NT) void fit_init_kernel(DevModel M, const ObsBlock* __restrict__ obs, DevPose P,
                                                           FitBuffers F, const float* __restrict__ params,
                                                           uint32_t flags, int publish) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    ClosureLds& L = *reinterpret_cast<ClosureLds*>(smem_raw);
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool use_vp = (flags & MVFIT_F_VPOSER) != 0;
    const float xv = (tid < DV) ? params[(size_t)b * DV + tid] : 0.f;
    const float xc = (tid < dact(use_vp)) ? params[(size_t)b * DV + cmap(tid, use_vp)] : 0.f;
    prologue(L, M, obs + b, nullptr, nullptr, nullptr, nullptr, nullptr, tid);
    for (int i = tid; i < (int)(sizeof(OptBlock) / 4); i += STEP_NT) reinterpret_cast<float*>(&L.opt)[i] = 0.f;
    __syncthreads();
    if (tid < DPAD) L.opt.x[tid] = xv;
    if (tid < LB_D) L.opt.lbV[tid / LB_EPL].x[tid % LB_EPL] = xc;
    if (tid == 0) { L.opt.lbS.phase = PH_STEP_START; L.opt.lbS.H = 1.0; }
    if (tid < MVFIT_MAX_STAGES) F.stage_final[(size_t)b * MVFIT_MAX_STAGES + tid] = (double)NAN;
    __syncthreads();
    store_block16(F.opt + b, &L.opt, sizeof(OptBlock), tid);
    if (publish) {
        pose_and_chain(M, L, flags, tid);
        publish_pose(L, P, b, tid);
        store_block16(F.pose + b, &L.pose, sizeof(PoseBlock), tid);
        if (flags & MVFIT_F_VPOSER) store_block16(F.vp + b, L.vp_pre1, sizeof(VpBlock), tid);
    }
}

// shared by the two fit kernels: evaluate the closure at L.opt.x, advance the optimiser, leave the
// next trial point in L.opt.x.  Returns true when the problem is finished.
// REMOTE: the launch may carry VPoser decoder helpers (fit_persistent_kernel only); REUSE: MVFIT_F_REUSE_OUTER_VALUE;
// LEAN: the stage flags carry none of VPoser / GMM / 3-D term (the host checks) - said to the compiler as a fact about
// the flag word, which lets it drop those branches from the round: 13 KB less code to stream through the instruction
// cache every round (86 -> 73 KB), 1.2-1.6 % per fit (speed only: the result does not depend on it)
// SDFS: the launch serves stages with the SDF term by asking for it (closure_device.h: publish_sdf_request, loss_combine<true>);
// sv = {pass operands of the chained layout (coefT), gate words, answer tags, global problem index, round offset of the launch}
struct SdfService { const DevPose* P; int* gate; const unsigned* tag; int b; int round0; };
template <bool REMOTE = false, bool REUSE = false, bool LEAN = false, bool COMPACT = false, bool SDFS = false, bool ROFF = SDFS>
__device__ __forceinline__ bool fit_round(const DevModel& M, ClosureLds& L, int nviews, const LbHist<float>& H,
                          bool from_pass, bool have_pose, double* stage_final, int tid,
                          LbGramLds GL = LbGramLds{nullptr, 0, 0}, float* trace = nullptr, int trace_cap = 0,
                          const AsyncRing& ring = AsyncRing{}, bool use_ring = false, int pb = 0,
                          const SdfService& sv = SdfService{nullptr, nullptr, nullptr, 0, 0}) {
    DevWeights W = L.sw[L.sh_stage];
    W.flags = __builtin_amdgcn_readfirstlane(W.flags);
    if constexpr (LEAN) {
        W.flags &= ~(uint32_t)(MVFIT_F_VPOSER | MVFIT_F_PRIOR_GMM | MVFIT_F_USE_3D);
        __builtin_assume((W.flags & (MVFIT_F_VPOSER | MVFIT_F_PRIOR_GMM | MVFIT_F_USE_3D)) == 0);
    }
    const LbOpts& O = L.opts;
    const bool use_vp = (W.flags & MVFIT_F_VPOSER) != 0;
    PH_T0();
    // have_pose: the previous launch left the pose block of this x (and, with VPoser, the decoder state the
    // adjoint needs - the VpBlock)
    if (!have_pose) {
        pose_prep_decode_inl<REMOTE>(M, L, W.flags, tid);
        pose_prep_elems(M, L, W.flags, tid);
    }
    PH_T(0);
    sparse_forward(M, L, from_pass, tid, !have_pose);
    PH_T(2);
    // asynchronous fit: the 6890-vertex pass of THIS trial point is already queued on the other CUs and waits for the
    // operands (coefficients, skinning transforms, translation: all complete here) in the ring slot of this round
    // closures consumed so far by this ring row = this round (sv.round0: the problem's closures before this launch, minus the
    // rounds the row spent on earlier problems of the launch - refill)
    const unsigned a_round = use_ring ? (unsigned)(L.opt.lbS.n_closure - (ROFF ? sv.round0 : 0)) : 0u;
    const int a_slot = use_ring ? (int)(a_round % (unsigned)ring.nslots) : 0;
    if (use_ring) publish_pose_async(L, ring, a_slot, a_round, pb, tid);
    bool sdf_round = false;
    if constexpr (SDFS) {
        // a stage that carries the interpenetration term: ask for S and its adjoint at this trial point (the tag goes out at
        // once: the round's passes and the term's kernels are queued behind it) and wait for the answer
        sdf_round = use_ring && L.sdf_adj != nullptr && W.coll_w > 0.f;           // block-uniform
        if (use_ring) publish_sdf_request(L, *sv.P, sv.gate, sv.b, sdf_round ? 1 : 0, tid);
        if (sdf_round) publish_tag(ring, a_slot, pb, a_round, tid);
        // the answer is waited for where S is first needed: by the wave that combines the loss's scalar terms, under E5
        // (closure_device.h: loss_combine<true>) - the keypoint phase overlaps the term's kernels.  Never a silently missing
        // term: a wait that times out makes the loss NaN and is counted (stats[3]: the host fails the fit)
        if (tid == 0) {
            L.sdf_wait_tag = sdf_round ? sv.tag + sv.b : nullptr;
            L.sdf_wait_want = a_round + 1u;
            L.sdf_wait_stats = ring.stats + 3;
        }
    }
    loss_and_keypoint_grad<true>(M, L, nviews, W, true, tid);          // (scalar terms combined under the adjoint's first phase)
    PH_T(3);
    closure_backward<REMOTE, true, SDFS>(M, L, nviews, W, tid);
    const double total = L.total;
    if (trace) {                                           // (x_trial, loss) of this closure call (mvfit_fit_trace)
        const int k = L.opt.lbS.n_closure;                 // closures consumed so far = index of this one
        if (k < trace_cap) {
            if (tid < DV) trace[(size_t)k * (DV + 1) + tid] = L.opt.x[tid];
            if (tid == 0) trace[(size_t)k * (DV + 1) + DV] = (float)total;
        }
    }
    if (use_ring && !sdf_round) publish_tag(ring, a_slot, pb, a_round, tid);             // the stores have long drained by now
    PH_T(8);
    float gnew[LB_EPL], xt[LB_EPL];
    const int D = dact(use_vp);
    if (tid < 64) {
        PH_T(9);
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) {
            const int i = LB_EPL * tid + e;
            gnew[e] = i < D ? L.grad[cmap(i, use_vp)] : 0.f;
        }
    }
    // the optimiser state stays in LDS (L.opt.lbS, L.opt.lbV): lbfgs_round works on it in place
    // the reference reads the loss as a float32 tensor (float(closure()), lbfgs_ls.py:251,281)
    lbfgs_round<float, STEP_NT, REUSE>(&L.opt.lbS, &L.opt.lbV[0], H, L.lbW, O, (double)(float)total, gnew, xt, tid, stage_final, [&]() {
        PH_T(10);
        // the single-launch fit takes the direction in compact form (history and R^-1 in LDS, every phase on all waves);
        // the chained step kernel keeps the two-loop form over its Gram matrices in global memory
        if constexpr (COMPACT) lb_direction_compact<float, STEP_NT>(H, L.lbW, tid);
        else lb_direction_block<float, STEP_NT>(H, L.lbW, tid, GL);
        PH_T(11); PH_ADD(15, 1);
    });
    if (tid < 64) {
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) {
            const int i = LB_EPL * tid + e;
            if (i < D) L.opt.x[cmap(i, use_vp)] = xt[e];
        }
        if (tid == 0) { L.sh_stage = min(L.opt.lbS.stage, O.num_stages - 1); L.sh_status = L.opt.lbS.status; }
        PH_ADD(13, 1); PH_ADD(14, L.opt.lbS.hist_len);
    }
    __syncthreads();
    PH_T(12);
    return L.sh_status != 0;
}

// LDS layout of the single-launch fit behind the closure workspace: [s ring | y ring | packed R^-1].  Without VPoser the
// tail starts over the decoder's arrays (the last members of ClosureLds) and a row holds the 86 active parameters; with
// VPoser the active dimension is 49.
constexpr int kHistLdFull = 88, kHistLdVp = 52;
__host__ __device__ constexpr int persistent_hist_ld(bool vp) { return vp ? kHistLdVp : kHistLdFull; }
__host__ __device__ constexpr size_t persistent_tail_offset(bool vp) {
    return vp ? ((sizeof(ClosureLds) + 15) & ~(size_t)15) : offsetof(ClosureLds, vp_pre1);
}
__host__ __device__ constexpr size_t persistent_lds_bytes(bool vp) {
    return persistent_tail_offset(vp) + ((size_t)2 * LB_HIST * persistent_hist_ld(vp) + LB_RPACK) * sizeof(float);
}
static_assert(persistent_lds_bytes(false) <= 160 * 1024 && persistent_lds_bytes(true) <= 160 * 1024, "one workgroup per CU: 160 KB of LDS");

__device__ __forceinline__ size_t step_lds_dev() { return (sizeof(ClosureLds) + 15) & ~(size_t)15; }

// one closure round per launch (full mode): the objective reads the vertex pass's output for its
// vertices; afterwards the pose operands of the NEXT trial point are published for the next pass.
template <bool REUSE>
__global__ __launch_bounds__(STEP_NT) void fit_step_kernel(DevModel M, const ObsBlock* __restrict__ obs, int nviews,
                                                           StageWeights SW, LbOpts O, DevPose P, FitBuffers F) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    ClosureLds& L = *reinterpret_cast<ClosureLds*>(smem_raw);
    const int b = blockIdx.x, tid = threadIdx.x;
    PH_T0();
    prologue(L, M, obs + b, F.pose + b, F.opt + b, P.vposed_sel + (size_t)b * NC_MAX, P.xs_sel + (size_t)b * NC_MAX, nullptr, tid,
             F.sdf_adj ? F.sdf_adj + b : nullptr, (SW.w[0].flags & MVFIT_F_VPOSER) ? F.vp + b : nullptr);
    opts_in(L, SW, O, tid);
    __syncthreads();
    if (L.opt.lbS.status != 0) return;                    // uniform per block
    if (tid == 0) { L.sh_stage = L.opt.lbS.stage; L.sh_status = 0; }
    LbHist<float> H{F.dirs + (size_t)b * LB_HIST * LB_D, F.stps + (size_t)b * LB_HIST * LB_D, L.opt.lb_ro,
                    F.grow + (size_t)b * LB_GSIZE, F.gcol + (size_t)b * LB_GSIZE};
    __syncthreads();
    // Gram rows of the first recurrence -> LDS while the closure runs (the window covers the current head / length and
    // the one after an insertion); lb_direction_block waits for it
    // Touch every 128-byte line of the live history rows (s and y) once, now: after a launch boundary they are
    // ~2.5 k cycles away, and the direction's row dots and mat-vecs - 18 k cycles from here - would each start with
    // that round trip; afterwards they hit L2.  One load per thread, value never used (kept alive to the end so that
    // the register is not recycled under the load).
    float warm = 0.f;
    {
        const int n0 = L.opt.lbS.hist_len, head0 = L.opt.lbS.hist_head;
        constexpr int LPR = LB_D * 4 / 128;                      // 3 lines per row
        if (tid < 2 * LPR * n0) {
            const int which = tid / (LPR * n0), r = tid - which * LPR * n0, age = r / LPR, ln = r - age * LPR;
            int slot = head0 + age;
            slot = slot >= LB_HIST ? slot - LB_HIST : slot;
            warm = (which ? H.stps : H.dirs)[slot * LB_D + ln * 32];
        }
    }
    LbGramLds GL{reinterpret_cast<float*>(smem_raw + step_lds_dev()), L.opt.lbS.hist_head,
                 min(L.opt.lbS.hist_len + 1, LB_HIST) + 3 + 4 * LB_PD};
    lb_gram_dma<STEP_NT>(H.gcol, GL.row0, GL.buf, GL.nrows, tid);
    PH_T(24);
    const bool done = fit_round<false, REUSE>(M, L, nviews, H, true, true, F.stage_final + (size_t)b * MVFIT_MAX_STAGES, tid, GL,
                                F.trace ? F.trace + (size_t)b * F.trace_cap * (DV + 1) : nullptr, F.trace_cap);
    PH_T0();
    store_block16(F.opt + b, &L.opt, sizeof(OptBlock), tid);
    if (tid == 0 && done) atomicAdd(F.n_done, 1);
    if (tid == 0 && F.sdf_adj) F.sdf_gate[b] = (!done && L.sw[L.sh_stage].coll_w > 0.f) ? 1 : 0;
    // pose operands of the next trial point (also after the last round: final vertices)
    pose_and_chain(M, L, __builtin_amdgcn_readfirstlane(L.sw[L.sh_stage].flags), tid);
    publish_pose(L, P, b, tid);
    store_block16(F.pose + b, &L.pose, sizeof(PoseBlock), tid);
    if (SW.w[0].flags & MVFIT_F_VPOSER) store_block16(F.vp + b, L.vp_pre1, sizeof(VpBlock), tid);
    if (__builtin_expect(warm == 1.7014118e38f, 0)) atomicAdd(F.n_done, 0);       // sink of the warm-up loads
    PH_T(25);
}

// the whole fit of one problem in a single launch (objective-vertices-only closure): the L-BFGS
// history ring lives in LDS behind the closure workspace.
// REMOTE: the launch carries VPoser decoder helpers behind the problems' workgroups (vposer_service.h); launches without
// them run the instantiation that has no trace of the service.
// QUEUE: the launch has a work queue (more problems than ring rows): its own instantiations - the loop over a row's problems around
// the round loop costs the round loop registers (22 instead of 7 spilled, +12 % instructions), which launches without a queue do
// not pay
template <bool REMOTE, bool REUSE, bool LEAN, bool SDFS = false, bool QUEUE = false>
__global__ __launch_bounds__(STEP_NT) void fit_persistent_kernel(DevModel M, const ObsBlock* __restrict__ obs, int nviews,
                                                                 StageWeights SW, LbOpts O, DevPose P, FitBuffers F,
                                                                 int max_rounds, AsyncRing ring, int b_lo, int done_target,
                                                                 int pause_stage, int* queue, int b_end) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if (REMOTE && (int)blockIdx.x >= M.vps.nprob) {
        // decoder helper of this launch (vposer_service.h): workgroups behind the problems' ones; set = blockIdx % nsets
        // like the problems it serves (dispatch is round-robin over the XCDs: same L2 when nsets == 8 - speed only)
        vposer_helper(M.vpt, M.vps, smem_raw, (int)blockIdx.x % M.vps.nsets, ((int)blockIdx.x - M.vps.nprob) / M.vps.nsets);
        return;
    }
    ClosureLds& L = *reinterpret_cast<ClosureLds*>(smem_raw);
    // behind (or, without VPoser, over the decoder's arrays at the end of) the closure workspace: the (s, y) ring, row
    // stride = the active dimension rounded up, and the packed R^-1 of the compact direction form
    const bool vp_mode = (SW.w[0].flags & MVFIT_F_VPOSER) != 0;                        // (flags are the same in all stages)
    const int ldh = LEAN ? kHistLdFull : persistent_hist_ld(vp_mode);
    float* hist = reinterpret_cast<float*>(smem_raw + (LEAN ? persistent_tail_offset(false) : persistent_tail_offset(vp_mode)));   // [2][100][ldh]
    float* rinv = hist + 2 * LB_HIST * ldh;                                              // [LB_RPACK]
    const int tid_k = threadIdx.x;
    const int row = b_lo + (int)blockIdx.x;                        // this workgroup's ring row / done_round word
    int b = row;                                                   // problems [b_lo, b_lo + nprob): one sub-batch of mvfit_fit ...
    // ... and, with a work queue (round 6: `queue` counts the problems handed out, b_end = one past the last), whatever problem
    // the workgroup takes when its own has finished: more problems than optimiser workgroups overlap in ONE launch instead of
    // running as sub-batches one after the other, and a workgroup whose problem converged early does not idle through the
    // tail of the slowest.  The ring row keeps counting closure rounds across its problems (rounds_before); the passes write a
    // round's vertices to the problem the row held in that round (its index travels in the translation word's spare lane).
    int rounds_before = 0, slot_rounds = 0;
  for (;;) {
    // (opaque per problem: nothing derived from the thread index is invariant across this loop - hoisted into its preheader, the
    // prologue's and epilogue's addresses would be live through every round loop: 179 spilled registers, 1.55 -> 1.42 M closures/s)
    int tid = tid_k;
    if constexpr (QUEUE) asm volatile("" : "+v"(tid));
    prologue(L, M, obs + b, nullptr, F.opt + b, nullptr, nullptr, nullptr, tid, SDFS && F.sdf_adj ? F.sdf_adj + b : nullptr);
    opts_in(L, SW, O, tid);
    __syncthreads();
    // closure rounds of THIS launch count from 0 (ring slots, tags, done_round): a service launch continues fits whose problems
    // have spent different numbers of closures in the stages before it
    const int round0 = (SDFS || QUEUE) ? L.opt.lbS.n_closure - rounds_before : 0;
    if (L.opt.lbS.status != 0) {
        if (REMOTE && tid == 0 && L.vp_remote) vps_store(vps_request_slot(M.vps), 0.f, 1u << 2 | VPS_BYE);
        if (tid == 0 && ring.tag) {         // finished in an earlier launch: no pass waits for this problem
            __hip_atomic_store(ring.done_round + row, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (SDFS) {                     // (the host counts the problems that left this launch)
                __hip_atomic_store(F.sdf_gate + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int left = atomicAdd(F.n_done + 1, 1) + 1;
                atomicAdd(F.n_done + 2, 1);
                if (left == done_target) __hip_atomic_store(ring.host_done, left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
    if (tid == 0) { L.sh_stage = L.opt.lbS.stage; L.sh_status = 0; L.sh_sdf_ok = 1u; L.sh_prob = b; }
    if (tid == 64 * PUBLISH_WAVE) L.sh_pass_done = 0u;
    float* gd = F.dirs + (size_t)b * LB_HIST * LB_D;
    float* gs = F.stps + (size_t)b * LB_HIST * LB_D;
    float* gr = F.rinv + (size_t)b * LB_RPACK;
    const bool resume = L.opt.lbS.n_closure > 0;          // relaunch after a round cap: restore the ring
    if (resume) {
        for (int i = tid; i < LB_HIST * ldh; i += STEP_NT) {
            const int r = i / ldh, e = i - r * ldh;
            hist[i] = gd[r * LB_D + e]; hist[LB_HIST * ldh + i] = gs[r * LB_D + e];
        }
        for (int i = tid; i < LB_RPACK; i += STEP_NT) rinv[i] = gr[i];
    } else {
        // dead history rows / R^-1 entries are read with zero coefficients (branch-free phases): they must hold finite values
        for (int i = tid; i < 2 * LB_HIST * ldh + LB_RPACK; i += STEP_NT) hist[i] = 0.f;
    }
    LbHist<float> H{hist, hist + LB_HIST * ldh, L.opt.lb_ro, nullptr, nullptr};
    H.ys = L.opt.lb_ys; H.rinv = rinv; H.ld = ldh;
    __syncthreads();
    bool done = false, paused = false;
    int stage_prev = L.sh_stage;
    for (; max_rounds <= 0 || slot_rounds < max_rounds; ++slot_rounds) {
        // opaque copy of the thread index: keeps the compiler from hoisting every tid-derived address
        // of the closure out of the round loop (which costs >256 live VGPRs and spills)
        int t = tid;
        asm volatile("" : "+v"(t));
        done = fit_round<REMOTE, REUSE, LEAN, true, SDFS, SDFS || QUEUE>(M, L, nviews, H, false, false, F.stage_final + (size_t)b * MVFIT_MAX_STAGES, t, LbGramLds{nullptr, 0, 0},
                         F.trace ? F.trace + (size_t)b * F.trace_cap * (DV + 1) : nullptr, F.trace_cap,
                         ring, ring.tag != nullptr, (int)blockIdx.x,        // ring slots: sub-batch-relative problem index
                         SdfService{SDFS ? &P : nullptr, SDFS ? F.sdf_gate : nullptr, SDFS ? F.sdf_tag : nullptr, b, round0});
        if (done) break;                                  // block-uniform
        if (L.sh_stage != stage_prev) {
            // a new stage starts with a fresh optimiser (non_linear_solver.py:172): its history is empty, and the branch-free
            // phases of the compact direction read dead rows with zero coefficients - a leftover inf / NaN row of a stage that
            // ran off would turn 0 * inf into NaN there.  Dead rows are zeros, as at the launch's start.
            for (int i = tid; i < 2 * LB_HIST * ldh + LB_RPACK; i += STEP_NT) hist[i] = 0.f;
            stage_prev = L.sh_stage;
            __syncthreads();
        }
        // two-phase fit (stages without the SDF term run here, the rest in chained rounds): leave at the stage boundary -
        // the trial point in L.opt.x is the first one of the next stage, the optimiser is fresh (non_linear_solver.py:172)
        if (L.sh_stage >= pause_stage) { paused = true; break; }
    }
    store_block16(F.opt + b, &L.opt, sizeof(OptBlock), tid);
    // the next problem of the batch, if the launch has a queue and this one is finished (a paused problem or the round cap ends
    // the workgroup): decided here, before the row says "nothing more comes"
    int b_next = -1;
    if (QUEUE && queue && done) {                                       // uniform
        if (tid == 0) L.sh_next = atomicAdd(queue, 1);
        __syncthreads();
        if (L.sh_next < b_end) b_next = L.sh_next;
    }
    // passes of later rounds have nothing to wait for from this row - whatever ended the launch for it (finished, paused at a
    // stage boundary, or the round cap: the resident pass ends when every row has said so)
    if (tid == 0 && ring.tag) {
        if (b_next < 0) __hip_atomic_store(ring.done_round + row, (unsigned)(L.opt.lbS.n_closure - round0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (SDFS) __hip_atomic_store(F.sdf_gate + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0 && (done || paused)) {
        if (done) atomicAdd(F.n_done, 1);
        const int left = atomicAdd(F.n_done + 1, 1) + 1;
        atomicAdd(F.n_done + 2, 1);
        // the last problem tells the host (per-round pass launches: it stops queueing them)
        if (ring.tag && left == done_target) __hip_atomic_store(ring.host_done, left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (!done) {
        for (int i = tid; i < LB_HIST * ldh; i += STEP_NT) {
            const int r = i / ldh, e = i - r * ldh;
            gd[r * LB_D + e] = hist[i]; gs[r * LB_D + e] = hist[LB_HIST * ldh + i];
        }
        for (int i = tid; i < LB_RPACK; i += STEP_NT) gr[i] = rinv[i];
    }
    if (REMOTE && L.vp_remote) {
        // goodbye to the helpers; the pose of the final point is decoded here (with the pre-activations the chained
        // rounds of a two-phase fit expect from their predecessor)
        __syncthreads();
        if (tid == 0) { vps_store(vps_request_slot(M.vps), 0.f, (L.vp_seq + 1u) << 2 | VPS_BYE); L.vp_remote = 0; }
        __syncthreads();
    }
    pose_and_chain<false>(M, L, __builtin_amdgcn_readfirstlane(L.sw[L.sh_stage].flags), tid);
    publish_pose(L, P, b, tid);
    if (paused) {
        // what the chained rounds' step kernel expects from its predecessor: the pose block of the trial point (+ the
        // decoder state with VPoser) and the SDF gate of the stage that starts
        store_block16(F.pose + b, &L.pose, sizeof(PoseBlock), tid);
        if (SW.w[0].flags & MVFIT_F_VPOSER) store_block16(F.vp + b, L.vp_pre1, sizeof(VpBlock), tid);
        if (tid == 0 && F.sdf_adj) F.sdf_gate[b] = L.sw[L.sh_stage].coll_w > 0.f ? 1 : 0;
    }
    if (!QUEUE || b_next < 0) break;
    rounds_before = L.opt.lbS.n_closure - round0;                      // the row's rounds so far
    b = b_next;
    __syncthreads();                                                   // (every thread is done with the finished problem's LDS image)
  }
}

__global__ void fit_finish_kernel(FitBuffers F, float* __restrict__ params, float* __restrict__ final_loss,
                                  int32_t* __restrict__ n_closure, int32_t* __restrict__ n_iter, int B,
                                  int num_stages) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < DV; i += blockDim.x) params[(size_t)b * DV + i] = F.opt[b].x[i];
    if (threadIdx.x == 0) {
        const LbState& s = F.opt[b].lbS;
        if (final_loss) final_loss[b] = (float)F.stage_final[(size_t)b * MVFIT_MAX_STAGES + num_stages - 1];
        if (n_closure) n_closure[b] = s.n_closure;
        if (n_iter) n_iter[b] = s.n_lbfgs;
    }
}

// ------------------------------------------------------------------ float64 known-answer test
__device__ double kat_eval(int kind, int D, const double* x, double* g) {
    // mirrors oracle/lbfgs_np.py:kat_objective (serial: one lane)
    double f = 0.0;
    if (kind == 0) {
        for (int i = 0; i < D; ++i) {
            double c = 1.0 + 99.0 * i / (D - 1), r = x[i] - sin((double)i);
            f += c * r * r; g[i] = c * r;
        }
        f *= 0.5;
    } else if (kind == 1) {
        for (int i = 0; i < D; ++i) g[i] = 0.0;
        for (int i = 0; i < D - 1; ++i) {
            double a = x[i + 1] - x[i] * x[i], bb = 1.0 - x[i];
            f += 100.0 * a * a + bb * bb;
            g[i] += -400.0 * a * x[i] - 2.0 * bb;
            g[i + 1] += 200.0 * a;
        }
    } else {
        const double rho2 = 1e4;
        for (int i = 0; i < D; ++i) g[i] = x[i];
        double q = 0.0;
        for (int i = 0; i < D; ++i) {
            int n = (i + 1) % D;
            double r = 50.0 * (x[i] - sin((double)i)) + 20.0 * sin(3.0 * x[n]);
            double r2 = r * r;
            f += rho2 * r2 / (r2 + rho2);
            q += x[i] * x[i];
            double dr = 2.0 * r * rho2 * rho2 / ((r2 + rho2) * (r2 + rho2));
            g[i] += 50.0 * dr;
            g[n] += dr * 60.0 * cos(3.0 * x[n]);
        }
        f += 0.5 * q;
    }
    return f;
}

__global__ __launch_bounds__(64) void lbfgs_kat_kernel(int kind, int D, LbOpts O, double* x_io, double* trace,
                                                       int max_trace, int* n_closure, double* final_loss,
                                                       double* dirs, double* stps, double* ro, double* grow,
                                                       double* gcol, double* cmat) {
    __shared__ double xs[LB_D], gs[LB_D];
    __shared__ double fsh;
    __shared__ LbWork<double> W;
    const bool compact = (kind & 0x100) != 0;              // direction in compact form (lb_direction_compact)
    kind &= 0xff;
    const int lane = threadIdx.x;
    // the state in memory, like the fit kernels keep it (lbfgs_round works on it in place)
    __shared__ LbState S;
    __shared__ LbVecs<double> Vm[LB_LANES];
    if (lane == 0) {
        memset(&S, 0, sizeof(S));
        S.phase = PH_STEP_START; S.H = 1.0;
    }
    LbHist<double> H{dirs, stps, ro, grow, gcol};
    H.rinv = cmat; H.ys = cmat + LB_RPACK;                // compact form: packed R^-1 and the diagonal y.s
    double xt[LB_EPL];
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) {
        const int i = LB_EPL * lane + e;
        xt[e] = i < D ? x_io[i] : 0.0;
    }
    {
        LbVecs<double> z;
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) { z.x[e] = xt[e]; z.d[e] = z.g[e] = z.pg[e] = z.gprev[e] = z.bg0[e] = z.bg1[e] = 0.0; }
        Vm[lane] = z;
    }
    __syncthreads();
    int ncl = 0;
    for (int round = 0; round < 100000; ++round) {
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) if (LB_EPL * lane + e < LB_D) xs[LB_EPL * lane + e] = xt[e];
        __syncthreads();
        if (lane == 0) fsh = kat_eval(kind, D, xs, gs);
        __syncthreads();
        const double f = fsh;
        if (ncl < max_trace && lane == 0) {
            for (int i = 0; i < D; ++i) trace[(size_t)ncl * (D + 1) + i] = xs[i];
            trace[(size_t)ncl * (D + 1) + D] = f;
        }
        ncl += 1;
        double gnew[LB_EPL];
#pragma unroll
        for (int e = 0; e < LB_EPL; ++e) gnew[e] = (LB_EPL * lane + e < D) ? gs[LB_EPL * lane + e] : 0.0;
        __syncthreads();
        lbfgs_round<double, 64, false>(&S, &Vm[0], H, W, O, f, gnew, xt, lane, final_loss, [&]() {   // the production round
            if (compact) lb_direction_compact<double, 64>(H, W, lane);
            else lb_direction_block<double, 64>(H, W, lane);
        });
        __syncthreads();
        if (S.status) break;
    }
#pragma unroll
    for (int e = 0; e < LB_EPL; ++e) if (LB_EPL * lane + e < D) x_io[LB_EPL * lane + e] = Vm[lane].x[e];
    if (lane == 0) *n_closure = ncl;
}

}  // namespace mvfit

// ==================================================================================== host side
using namespace mvfit;

struct mvfit_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    DevModel M{};
    std::vector<void*> allocs;
    bool upload_failed = false;
    int nv = 0;
    // problems
    DevProblems Q{};
    int B = 0, Bpad = 0, V = 0;
    float *d_camR = nullptr, *d_camt = nullptr, *d_camf = nullptr, *d_camc = nullptr, *d_gt = nullptr, *d_wc = nullptr;
    ObsBlock* d_obs = nullptr;         // [B] packed observations (LDS image block)
    // per-problem work buffers
    DevPose P{};
    float* d_verts = nullptr;          // [B][nv][3] internal vertex buffer
    FitBuffers F{};
    int* h_done = nullptr;             // pinned, 2 slots
    hipEvent_t ev_done[2] = {nullptr, nullptr};
    bool has_vposer = false;
    bool has_joints3d = false;
    float *d_gt3d = nullptr, *d_c3d = nullptr;   // staging of mvfit_set_joints3d ([B][17][3], [B][17])
    // asynchronous full-mode fit: ring of pose operands + the side stream the vertex passes are queued on
    AsyncRing ring{};
    hipStream_t pass_stream = nullptr;
    hipEvent_t ev_batch[4] = {nullptr, nullptr, nullptr, nullptr}, ev_init = nullptr;
    int* h_async_done = nullptr;       // pinned host word the last finishing problem writes
    unsigned async_stats[4] = {0, 0, 0, 0};
    mvfit_options opt{};               // precision / path selectors (include/mvfit.h); the library reads no environment variable
    int n_cu = 0;                      // compute units of the device (residency of the resident vertex pass)
    int resident_tpw = 0;              // tiles per workgroup of the resident pass in the last asynchronous fit (0: per-round launches)
    int* d_queue = nullptr;            // work queue of a single-launch fit with more problems than rows: next problem to hand out
    int h_queue0 = 0;
    bool resident_auto_off = false;    // automatic resident_pass: a fit on this ctx timed out waiting - later fits use per-round launches
    unsigned long long* d_vp_log = nullptr;     // mvfit_profile: per-round stamps of the resident pass [kVpLogRounds][grid][2]
    size_t vp_log_words = 0;
    double res_span_ms = 0.0, res_busy_ms = 0.0, res_slowest_ms = 0.0;   // per round: service span / mean workgroup busy time / slowest workgroup (last profiled fit)
    int res_rounds = 0;
    // decoder helpers of the single-launch fit (vposer_service.h): granule memory [requests | answers | 2 counters]
    unsigned long long* vps_mem = nullptr;
    size_t vps_words = 0;
    unsigned vps_stats[3] = {0, 0, 0};     // launches with helpers in the last fit, answers timed out, helpers that gave up
    float* capture_verts = nullptr;    // mvfit_debug_capture_pass: the pass of closure round capture_round writes here
    int capture_round = -1;
    float* trace = nullptr;            // caller's device buffer (mvfit_fit_trace), not owned
    int trace_cap = 0;
    int gmm_M = 0;
    // full-mode round loop captured as a graph: key = everything baked into the kernel nodes
    hipGraphExec_t round_graph = nullptr;
    std::vector<unsigned char> graph_key;
    int graph_rounds = 0;
    // SDF interpenetration term (mvfit_set_sdf): faces as the reference's caller hands them to the op
    int32_t* d_sdf_faces = nullptr;
    int sdf_num_faces = 0, sdf_grid = 0;
    SdfBox* d_sdf_box = nullptr;       // [B]
    float4* d_sdf_samp = nullptr;      // [B][nv]
    void* d_sdf_entries = nullptr;     // [B][nv] entry list
    SdfAdj* d_sdf_adj = nullptr;       // [B]
    unsigned long long* d_sdf_boxpart = nullptr;   // [B][ntiles][6] the vertex pass's own per-tile keys of the term's bounding box (single-chunk split kernel)
    void* d_sdf_cull = nullptr;        // face lists of the all-faces term (sdf_term.hip), sized for (B, sdf_num_faces)
    void* d_sdf_op_ws = nullptr;       // face lists of the stand-alone op (mvfit_sdf), kept between calls of one shape
    int sdf_op_B = 0, sdf_op_F = 0;
    // which path served the last mvfit_sdf / the SDF term of the last fit (mvfit_sdf_info): 0 walk over every face (short
    // list or lists switched off), 1 face lists, 2 walk because the lists' workspace did not fit
    int sdf_op_path = 0, sdf_term_path = 0;
    bool sdf_cull_refused = false;      // the term's workspace did not fit for the current (batch, face list)
    // profiling
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_vp, ev_step;
};

static int fail(mvfit_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}
#define HIP_OK(c, call)                                                                          \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) return fail(c, MVFIT_E_HIP, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

template <typename T>
static T* dev_upload(mvfit_ctx* c, const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)) != hipSuccess) d = nullptr;
    c->allocs.push_back(d);                 // a null entry makes mvfit_create fail (checked after all uploads)
    if (d && !h.empty() && hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) c->upload_failed = true;
    return d;
}

static size_t step_lds() { return (sizeof(ClosureLds) + 15) & ~(size_t)15; }
static size_t step_gram_lds() { return step_lds() + LB_GW_BYTES; }       // fit_step_kernel: + the staged Gram window
static size_t persistent_lds(bool vp) { return std::max(persistent_lds_bytes(vp), sizeof(VpHelperLds)); }

static void drop_graph(mvfit_ctx* c) {
    if (c->round_graph) { hipGraphExecDestroy(c->round_graph); c->round_graph = nullptr; }
    c->graph_key.clear();
}

extern "C" const char* mvfit_last_error(const mvfit_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

// Developer hooks (fault injection, experiment switches) exist only in the -DMVFIT_DEBUG_HOOKS variant build the tests that
// need them load (libmvfit_hooks.so); the released library has no trace of them and reads no environment variable.
#ifdef MVFIT_DEBUG_HOOKS
static int debug_hook(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
#else
static constexpr int debug_hook(const char*) { return 0; }
#endif

extern "C" void mvfit_options_default(mvfit_options* o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->struct_size = (uint32_t)sizeof(mvfit_options);
    o->contraction = MVFIT_CONTRACTION_SPLIT_FP16;
    o->resident_pass = -1;
    o->sdf_two_phase = 1;
    o->sdf_face_lists = 1;
    o->vposer_helpers = 1;
    o->sdf_service = 1;
    o->work_queue = 1;
}

// a caller's struct (possibly shorter: an older header) over the defaults; range checks
static int read_options(mvfit_ctx* c, const mvfit_options* in, mvfit_options& o) {
    mvfit_options_default(&o);
    if (in) {
        if (in->struct_size < 8 || in->struct_size > 4096) return fail(c, MVFIT_E_ARG, "mvfit_options: struct_size %u", in->struct_size);
        memcpy(&o, in, std::min<size_t>(in->struct_size, sizeof(o)));
        o.struct_size = (uint32_t)sizeof(o);
    }
    if (o.contraction < 0 || o.contraction > MVFIT_CONTRACTION_HALF_BASIS) return fail(c, MVFIT_E_ARG, "mvfit_options: contraction %d", o.contraction);
    if (o.round_mode < 0 || o.round_mode > 1) return fail(c, MVFIT_E_ARG, "mvfit_options: round_mode %d", o.round_mode);
    if (o.resident_pass < -1 || o.resident_pass > 3) return fail(c, MVFIT_E_ARG, "mvfit_options: resident_pass %d", o.resident_pass);
    if (o.pass_kernel < 0 || o.pass_kernel > 2) return fail(c, MVFIT_E_ARG, "mvfit_options: pass_kernel %d", o.pass_kernel);
    if (o.vposer_sets < 0) return fail(c, MVFIT_E_ARG, "mvfit_options: vposer_sets %d", o.vposer_sets);
    return MVFIT_OK;
}

extern "C" int mvfit_create_ex(mvfit_ctx** out, int device, void* hip_stream, const mvfit_model* m, const mvfit_options* opts);
extern "C" int mvfit_create(mvfit_ctx** out, int device, void* hip_stream, const mvfit_model* m) {
    return mvfit_create_ex(out, device, hip_stream, m, nullptr);
}

extern "C" int mvfit_get_options(const mvfit_ctx* c, mvfit_options* o) {
    if (!c || !o) return MVFIT_E_ARG;
    *o = c->opt;
    return MVFIT_OK;
}

static void drop_graph(mvfit_ctx* c);
extern "C" int mvfit_set_options(mvfit_ctx* c, const mvfit_options* opts) {
    if (!c || !opts) return MVFIT_E_ARG;
    mvfit_options o;
    const int rc = read_options(c, opts, o);
    if (rc) return rc;
    if (o.contraction != c->opt.contraction || o.dense_skinning != c->opt.dense_skinning)
        return fail(c, MVFIT_E_ARG, "mvfit_set_options: contraction / dense_skinning are fixed at mvfit_create_ex");
    if (o.pass_kernel != c->opt.pass_kernel || o.sdf_face_lists != c->opt.sdf_face_lists) {
        HIP_OK(c, hipSetDevice(c->device));
        HIP_OK(c, hipStreamSynchronize(c->stream));
        drop_graph(c);                                   // the captured round graph bakes the kernel choice in
        if (o.sdf_face_lists != c->opt.sdf_face_lists) c->sdf_cull_refused = false;
    }
    c->opt = o;
    c->resident_auto_off = false;                        // (an explicit call re-arms the automatic resident_pass choice)
    return MVFIT_OK;
}

extern "C" int mvfit_sdf_info(const mvfit_ctx* c, int* op_path, int* term_path) {
    if (!c) return MVFIT_E_ARG;
    if (op_path) *op_path = c->sdf_op_path;
    if (term_path) *term_path = c->sdf_term_path;
    return MVFIT_OK;
}

extern "C" int mvfit_create_ex(mvfit_ctx** out, int device, void* hip_stream, const mvfit_model* m, const mvfit_options* opts) {
    if (!out || !m || !m->v_template || !m->shapedirs || !m->posedirs || !m->J_regressor || !m->parents ||
        !m->lbs_weights || !m->kp_regressor || !m->face_vertex_ids || !m->joint_map || m->num_verts <= 0) {
        if (out) *out = nullptr;
        return MVFIT_E_ARG;
    }
    // From here on *out is a ctx even when an error is returned: it holds the message for mvfit_last_error and owns
    // whatever was allocated so far - the caller releases both with mvfit_destroy (include/mvfit.h).
    mvfit_ctx* c = new mvfit_ctx();
    *out = c;
    c->device = device;
    HIP_OK(c, hipSetDevice(device));
    c->stream = (hipStream_t)hip_stream;
    {
        const int rc = read_options(c, opts, c->opt);
        if (rc) return rc;
    }
    const int nv = m->num_verts;
    c->nv = nv;
    DevModel& M = c->M;
    M.nv = nv;
    M.ntiles = (nv + TILE_V - 1) / TILE_V;
    M.nv_pad = M.ntiles * TILE_V;
    if (m->parents[0] >= 0) return fail(c, MVFIT_E_ARG, "parents[0] must be -1");
    if (persistent_lds(false) > 160 * 1024 || persistent_lds(true) > 160 * 1024)
        return fail(c, MVFIT_E_UNSUPPORTED, "LDS budget exceeded (%zu / %zu B)", persistent_lds(false), persistent_lds(true));
    static_assert(sizeof(VpHelperLds) <= sizeof(ClosureLds), "the decoder helpers share the fit kernel's dynamic LDS");

    // ---- blendshape basis, re-tiled in MFMA B-operand order: [tile][coord][group][lane][4] ----
    // element (tile T, coord k, group g, lane l, q): row p = 2*(4g+q) + (l>>5), vertex v = 32T + (l&31)
    // rows 0..206 posedirs, 207..216 shapedirs (beta index), rest zero.
    {
        std::vector<float> bs((size_t)M.ntiles * 3 * KGROUPS * 64 * 4, 0.f);
        for (int T = 0; T < M.ntiles; ++T)
            for (int k = 0; k < 3; ++k)
                for (int g = 0; g < KGROUPS; ++g)
                    for (int l = 0; l < 64; ++l)
                        for (int q = 0; q < 4; ++q) {
                            const int p = 2 * (4 * g + q) + (l >> 5);
                            const int v = TILE_V * T + (l & 31);
                            float val = 0.f;
                            if (v < nv) {
                                if (p < 207) val = m->posedirs[(size_t)p * nv * 3 + 3 * v + k];
                                else if (p < 217) val = m->shapedirs[((size_t)v * 3 + k) * 10 + (p - 207)];
                            }
                            bs[((((size_t)T * 3 + k) * KGROUPS + g) * 64 + l) * 4 + q] = val;
                        }
        M.bs4 = dev_upload(c, bs);
        // the same basis as split-fp16 MFMA B operands (vertex_pass.hip: lbs_vertex_pass_split_kernel):
        // x * scale = hi + lo, scale = the power of two that brings max |x| into [2^13, 2^14)
        M.bs_h2 = nullptr; M.bs_scale = 1.f; M.half_basis = 0;
        // MVFIT_CONTRACTION_HALF_BASIS (BASELINE configs[4]: half-width blendshape operands): the contraction streams only the
        // fp16 hi halves of the basis - 2 bytes per element like bf16, with 11 instead of 8 significant bits
        M.half_basis = c->opt.contraction == MVFIT_CONTRACTION_HALF_BASIS ? 1 : 0;
        if (c->opt.contraction != MVFIT_CONTRACTION_EXACT_FP32) {
            float mx = 0.f;
            for (float v : bs) mx = std::max(mx, std::fabs(v));
            int ex = 0;
            if (mx > 0.f) std::frexp(mx, &ex);                 // mx = f * 2^ex, f in [0.5, 1)
            const float scale = std::ldexp(1.f, 14 - ex);       // max |x| * scale in [2^13, 2^14)
            constexpr int NB = KROWS / 16;
            std::vector<_Float16> h2((size_t)M.ntiles * 3 * NB * 2 * 64 * 8);
            for (int T = 0; T < M.ntiles; ++T)
                for (int k = 0; k < 3; ++k)
                    for (int G16 = 0; G16 < NB; ++G16)
                        for (int l = 0; l < 64; ++l)
                            for (int t = 0; t < 8; ++t) {
                                const int pr = 16 * G16 + 8 * (l >> 5) + t;
                                const int v = TILE_V * T + (l & 31);
                                float val = 0.f;
                                if (v < nv) {
                                    if (pr < 207) val = m->posedirs[(size_t)pr * nv * 3 + 3 * v + k];
                                    else if (pr < 217) val = m->shapedirs[((size_t)v * 3 + k) * 10 + (pr - 207)];
                                }
                                val *= scale;
                                const _Float16 hi = (_Float16)val;
                                const _Float16 lo = (_Float16)(val - (float)hi);
                                const size_t at = ((((size_t)(T * 3 + k) * NB + G16) * 2) * 64 + l) * 8 + t;
                                h2[at] = hi;
                                h2[at + 64 * 8] = lo;
                            }
            M.bs_h2 = reinterpret_cast<const float4*>(dev_upload(c, h2));
            M.bs_scale = scale;
        }
        std::vector<float> vtp((size_t)3 * M.nv_pad, 0.f);
        for (int v = 0; v < nv; ++v)
            for (int k = 0; k < 3; ++k) vtp[(size_t)k * M.nv_pad + v] = m->v_template[3 * v + k];
        M.vt_planes = dev_upload(c, vtp);
        std::vector<float> wt((size_t)M.ntiles * NJ * 32, 0.f);
        for (int v = 0; v < nv; ++v)
            for (int j = 0; j < NJ; ++j)
                wt[((size_t)(v / 32) * NJ + j) * 32 + (v % 32)] = m->lbs_weights[(size_t)v * NJ + j];
        M.wt_tiles = dev_upload(c, wt);
        // sparse skinning table when the model allows it (SMPL-family weights have <= 4 non-zeros per vertex);
        // mvfit_options::dense_skinning keeps the dense blend (tests compare the two bit for bit)
        {
            bool sparse_ok = true;
            for (int v = 0; v < nv && sparse_ok; ++v) {
                int nz = 0;
                for (int j = 0; j < NJ; ++j) nz += m->lbs_weights[(size_t)v * NJ + j] != 0.f;
                sparse_ok = nz <= 4;
            }
            if (c->opt.dense_skinning) sparse_ok = false;
            M.wsp_w = nullptr; M.wsp_j = nullptr;
            if (sparse_ok) {
                std::vector<float> sw((size_t)M.nv_pad * 4, 0.f);
                std::vector<int> sj((size_t)M.nv_pad * 4, 0);
                for (int v = 0; v < nv; ++v) {
                    int t = 0;
                    for (int j = 0; j < NJ; ++j) {
                        const float w = m->lbs_weights[(size_t)v * NJ + j];
                        if (w != 0.f) { sw[(size_t)v * 4 + t] = w; sj[(size_t)v * 4 + t] = j; ++t; }
                    }
                }
                M.wsp_w = reinterpret_cast<const float4*>(dev_upload(c, sw));
                M.wsp_j = reinterpret_cast<const int4*>(dev_upload(c, sj));
            }
        }
        // vertex-major copies (SDF term pull-back): coefficient row order (posedirs 0..206, shapedirs 207..216)
        std::vector<float> bsv((size_t)nv * 3 * KROWS, 0.f);
        for (int v = 0; v < nv; ++v)
            for (int k = 0; k < 3; ++k) {
                float* row = &bsv[((size_t)v * 3 + k) * KROWS];
                for (int pp = 0; pp < 207; ++pp) row[pp] = m->posedirs[(size_t)pp * nv * 3 + 3 * v + k];
                for (int l = 0; l < 10; ++l) row[207 + l] = m->shapedirs[((size_t)v * 3 + k) * 10 + l];
            }
        M.bs_vm = dev_upload(c, bsv);
        std::vector<float> wv((size_t)nv * NJ);
        for (size_t i = 0; i < wv.size(); ++i) wv[i] = m->lbs_weights[i];
        M.w_vm = dev_upload(c, wv);
    }
    // ---- the LDS image of the per-problem kernels ----
    std::vector<ModelLds> imgv(1);
    ModelLds& G = imgv[0];
    memset(&G, 0, sizeof(G));
    // joints as an affine function of beta (float64 accumulation on the host)
    for (int j = 0; j < NJ; ++j)
        for (int a = 0; a < 3; ++a) {
            double s = 0.0, sl[10] = {0};
            for (int v = 0; v < nv; ++v) {
                const double w = m->J_regressor[(size_t)j * nv + v];
                if (w == 0.0) continue;
                s += w * m->v_template[3 * v + a];
                for (int l = 0; l < 10; ++l) sl[l] += w * m->shapedirs[((size_t)v * 3 + a) * 10 + l];
            }
            G.J_t[j * 3 + a] = (float)s;
            for (int l = 0; l < 10; ++l) G.J_S[j * 3 + a][l] = (float)sl[l];
        }
    // the vertices the objective reads: non-zero columns of the mapped 17 x Nv selection
    {
        std::vector<double> ksel((size_t)NKP * nv, 0.0);
        for (int k = 0; k < NKP; ++k) {
            const int src = m->joint_map[k];
            if (src < 0 || src >= 19) return fail(c, MVFIT_E_ARG, "joint_map entry out of range");
            if (src < 14) for (int v = 0; v < nv; ++v) ksel[(size_t)k * nv + v] = m->kp_regressor[(size_t)src * nv + v];
            else {
                const int v = m->face_vertex_ids[src - 14];
                if (v < 0 || v >= nv) return fail(c, MVFIT_E_ARG, "face vertex id out of range");
                ksel[(size_t)k * nv + v] = 1.0;
            }
        }
        std::vector<int> sel;
        for (int v = 0; v < nv; ++v) {
            bool nz = false;
            for (int k = 0; k < NKP; ++k) nz |= ksel[(size_t)k * nv + v] != 0.0;
            if (nz) sel.push_back(v);
        }
        if ((int)sel.size() > NS_MAX) return fail(c, MVFIT_E_UNSUPPORTED, "keypoint regressor touches %d vertices (max %d)", (int)sel.size(), NS_MAX);
        M.ns = (int)sel.size();
        M.nc = 3 * M.ns;
        M.nc_pad = (M.nc + 3) & ~3;
        G.ns = M.ns; G.nc = M.nc; G.nc_pad = M.nc_pad;
        M.sel_v = dev_upload(c, sel);
        std::vector<float> pds((size_t)KROWS * M.nc_pad, 0.f), pdsT((size_t)M.nc_pad * KROWS, 0.f);
        int sel_sparse = 1;
        for (int s = 0; s < M.ns; ++s) {
            const int v = sel[s];
            G.sel_v[s] = v;
            for (int a = 0; a < 3; ++a) {
                const int cidx = 3 * s + a;
                G.vt_sub[cidx] = m->v_template[3 * v + a];
                for (int p = 0; p < 217; ++p) {
                    const float val = p < 207 ? m->posedirs[(size_t)p * nv * 3 + 3 * v + a]
                                              : m->shapedirs[((size_t)v * 3 + a) * 10 + (p - 207)];
                    pds[(size_t)p * M.nc_pad + cidx] = val;
                    pdsT[(size_t)cidx * KROWS + p] = val;
                }
            }
            for (int j = 0; j < NJ; ++j) G.wT[j][s] = m->lbs_weights[(size_t)v * NJ + j];
            int np = 0;
            G.selj[s] = 0u;
            for (int t = 0; t < 4; ++t) G.selw[s][t] = 0.f;
            for (int j = 0; j < NJ; ++j) {
                const float w = m->lbs_weights[(size_t)v * NJ + j];
                if (w == 0.f) continue;
                if (np < 4) { G.selw[s][np] = w; G.selj[s] |= (unsigned)j << (8 * np); }
                ++np;
            }
            if (np > 4) sel_sparse = 0;
        }
        G.sel_sparse = sel_sparse;
        M.pd_sub = dev_upload(c, pds);
        M.pd_subT = dev_upload(c, pdsT);
        // selection in CSR form, both ways
        int nnz = 0;
        for (int k = 0; k < NKP; ++k) {
            G.kp_start[k] = nnz;
            for (int s = 0; s < M.ns; ++s) {
                const double w = ksel[(size_t)k * nv + sel[s]];
                if (w == 0.0) continue;
                if (nnz >= KNNZ_MAX) return fail(c, MVFIT_E_UNSUPPORTED, "keypoint selection has more than %d non-zeros", KNNZ_MAX);
                G.kp_s[nnz] = s; G.kp_w[nnz] = (float)w; ++nnz;
            }
        }
        G.kp_start[NKP] = nnz;
        nnz = 0;
        for (int s = 0; s < M.ns; ++s) {
            G.vs_start[s] = nnz;
            for (int k = 0; k < NKP; ++k) {
                const double w = ksel[(size_t)k * nv + sel[s]];
                if (w == 0.0) continue;
                G.vs_k[nnz] = k; G.vs_w[nnz] = (float)w; ++nnz;
            }
        }
        for (int s = M.ns; s <= NS_MAX; ++s) G.vs_start[s] = nnz;
        // fixed-length zero-padded copies (entry 0 / weight 0 pads: fmaf(0, x, acc) == acc)
        G.padded = 1;
        for (int k = 0; k < NKP; ++k) {
            const int n0 = G.kp_start[k], cnt = G.kp_start[k + 1] - n0;
            if (cnt > KP_NZ) G.padded = 0;
            for (int t = 0; t < KP_NZ; ++t) { G.kpp_s[k][t] = t < cnt ? G.kp_s[n0 + t] : 0; G.kpp_w[k][t] = t < cnt ? G.kp_w[n0 + t] : 0.f; }
        }
        for (int s = 0; s < NS_MAX; ++s) {
            const int n0 = G.vs_start[s], cnt = G.vs_start[s + 1] - n0;
            if (cnt > VS_NZ) G.padded = 0;
            for (int t = 0; t < VS_NZ; ++t) { G.vsp_k[s][t] = t < cnt ? G.vs_k[n0 + t] : 0; G.vsp_w[s][t] = t < cnt ? G.vs_w[n0 + t] : 0.f; }
        }
        // per-tile lists for the vertex pass side outputs
        std::vector<int> tstart(M.ntiles + 1, 0), tlocal(std::max(M.ns, 1)), tslot(std::max(M.ns, 1));
        int pos = 0;
        for (int T = 0; T < M.ntiles; ++T) {
            tstart[T] = pos;
            for (int s = 0; s < M.ns; ++s)
                if (sel[s] / TILE_V == T) { tlocal[pos] = sel[s] % TILE_V; tslot[pos] = s; ++pos; }
        }
        tstart[M.ntiles] = pos;
        M.tile_sel_start = dev_upload(c, tstart);
        M.tile_sel_local = dev_upload(c, tlocal);
        M.tile_sel_slot = dev_upload(c, tslot);
    }
    // kinematic tree: levels and child lists
    {
        int depth[NJ];
        for (int j = 0; j < NJ; ++j) {
            G.parents[j] = m->parents[j];
            if (j > 0 && (m->parents[j] < 0 || m->parents[j] >= j)) return fail(c, MVFIT_E_ARG, "parents must be topologically ordered");
            depth[j] = j == 0 ? 0 : depth[m->parents[j]] + 1;
        }
        int maxd = 0;
        for (int j = 0; j < NJ; ++j) maxd = std::max(maxd, depth[j]);
        G.nlevels = maxd + 1;
        int pos = 0;
        for (int lv = 0; lv <= maxd; ++lv) {
            G.level_start[lv] = pos;
            for (int j = 0; j < NJ; ++j) if (depth[j] == lv) G.level_joints[pos++] = j;
        }
        for (int lv = maxd + 1; lv <= NJ; ++lv) G.level_start[lv] = pos;
        pos = 0;
        for (int p = 0; p < NJ; ++p) {
            G.child_start[p] = pos;
            for (int j = 1; j < NJ; ++j) if (m->parents[j] == p) G.child_list[pos++] = j;
        }
        G.child_start[NJ] = pos;
        // forward schedule: each level in groups of 5 joints (one wave = 5 x 12 lanes)
        memset(G.fwd_tab, 0xff, sizeof(G.fwd_tab));
        memset(G.bwd_tab, 0xff, sizeof(G.bwd_tab));
        int np = 0;
        for (int lv = 1; lv <= maxd; ++lv)
            for (int base = G.level_start[lv]; base < G.level_start[lv + 1]; base += 5, ++np) {
                if (np >= NJ) return fail(c, MVFIT_E_UNSUPPORTED, "kinematic tree needs more than %d chain passes", NJ);
                for (int q = 0; q < 5 && base + q < G.level_start[lv + 1]; ++q) {
                    const int j = G.level_joints[base + q];
                    G.fwd_tab[np][q] = j | (m->parents[j] << 8);
                }
            }
        G.n_fwd = np;
        // pointer-jumping tables (chain_forward_block)
        {
            for (int j = 0; j < NJ; ++j) G.anc_tab[0][j] = m->parents[j];
            for (int st = 1; st < 5; ++st)
                for (int j = 0; j < NJ; ++j) {
                    const int a = G.anc_tab[st - 1][j];
                    G.anc_tab[st][j] = a < 0 ? -1 : G.anc_tab[st - 1][a];
                }
            int nj = 0;
            while ((1 << nj) < maxd + 1) ++nj;
            if (nj > 5) return fail(c, MVFIT_E_UNSUPPORTED, "kinematic tree deeper than 32 joints");
            G.n_jump = nj;
        }
        // backward schedule: parents with children, deepest level first; <= 3 children per entry
        // (a parent with more children appears in consecutive passes), <= 5 entries per pass.
        // Two entries of the same parent never share a pass (they would race on its row).
        np = 0;
        for (int lv = maxd - 1; lv >= 0; --lv) {
            std::vector<int> entries;     // packed words of this level
            for (int i = G.level_start[lv]; i < G.level_start[lv + 1]; ++i) {
                const int p = G.level_joints[i];
                const int nc = G.child_start[p + 1] - G.child_start[p];
                for (int k = 0; k < nc; k += 3) {
                    int ch[3] = {31, 31, 31};
                    for (int t = 0; t < 3 && k + t < nc; ++t) ch[t] = G.child_list[G.child_start[p] + k + t];
                    entries.push_back(p | (k > 0 ? 0x80 : 0) | (ch[0] << 8) | (ch[1] << 16) | (ch[2] << 24));
                }
            }
            // greedy packing into passes: at most 5 entries, no repeated parent inside a pass
            std::vector<bool> used(entries.size(), false);
            size_t left = entries.size();
            while (left > 0) {
                if (np >= NJ) return fail(c, MVFIT_E_UNSUPPORTED, "kinematic tree needs more than %d adjoint passes", NJ);
                int q = 0;
                std::vector<int> parents_in_pass;
                for (size_t i = 0; i < entries.size() && q < 5; ++i) {
                    if (used[i]) continue;
                    const int p = entries[i] & 0x1f;
                    bool clash = false;
                    for (int pp : parents_in_pass) clash |= pp == p;
                    if (clash) continue;
                    G.bwd_tab[np][q++] = entries[i];
                    parents_in_pass.push_back(p);
                    used[i] = true;
                    --left;
                }
                ++np;
            }
        }
        G.n_bwd = np;
    }
    M.mlds = dev_upload(c, imgv);
    // ---- VPoser decoder ----
    if (m->vp_fc1_w) {
        if (!m->vp_fc1_b || !m->vp_fc2_w || !m->vp_fc2_b || !m->vp_out_w || !m->vp_out_b) return fail(c, MVFIT_E_ARG, "incomplete vposer weights");
        std::vector<float> w1(m->vp_fc1_w, m->vp_fc1_w + 512 * 32), b1(m->vp_fc1_b, m->vp_fc1_b + 512),
            w2(m->vp_fc2_w, m->vp_fc2_w + 512 * 512), b2(m->vp_fc2_b, m->vp_fc2_b + 512),
            w3(m->vp_out_w, m->vp_out_w + 138 * 512), b3(m->vp_out_b, m->vp_out_b + 138);
        std::vector<float> w1T(32 * 512), w2T(512 * 512), w3T(512 * 144, 0.f);
        for (int o = 0; o < 512; ++o) for (int i = 0; i < 32; ++i) w1T[i * 512 + o] = w1[o * 32 + i];
        for (int o = 0; o < 512; ++o) for (int i = 0; i < 512; ++i) w2T[i * 512 + o] = w2[o * 512 + i];
        for (int o = 0; o < 138; ++o) for (int i = 0; i < 512; ++i) w3T[i * 144 + o] = w3[o * 512 + i];
        M.vp_w1 = dev_upload(c, w1); M.vp_b1 = dev_upload(c, b1);
        M.vp_w2 = dev_upload(c, w2); M.vp_b2 = dev_upload(c, b2);
        M.vp_w3 = dev_upload(c, w3); M.vp_b3 = dev_upload(c, b3);
        M.vp_w1T = dev_upload(c, w1T); M.vp_w2T = dev_upload(c, w2T); M.vp_w3T = dev_upload(c, w3T);
        {   // register tiles of the decoder helpers (vposer_service.h: VpTiles), 16-byte words, thread-minor
            std::vector<float> tw2((size_t)VPS_SLICES * 16 * 512 * 4), tw3((size_t)VPS_SLICES * 6 * 512 * 4, 0.f);
            for (int h = 0; h < VPS_SLICES; ++h)
                for (int tid = 0; tid < 512; ++tid) {
                    const int w = tid >> 6, l = tid & 63;
                    for (int j = 0; j < 16; ++j)
                        for (int q = 0; q < 4; ++q)
                            tw2[(((size_t)h * 16 + j) * 512 + tid) * 4 + q] = w2[(size_t)(64 * h + 8 * w + (j >> 1)) * 512 + 8 * l + 4 * (j & 1) + q];
                    for (int j = 0; j < 6; ++j) {
                        const int o = l + 64 * (j >> 1);
                        if (o < 138)
                            for (int q = 0; q < 4; ++q)
                                tw3[(((size_t)h * 6 + j) * 512 + tid) * 4 + q] = w3[(size_t)o * 512 + 64 * h + 8 * w + 4 * (j & 1) + q];
                    }
                }
            M.vpt.tw2 = reinterpret_cast<const float4*>(dev_upload(c, tw2));
            M.vpt.tw3 = reinterpret_cast<const float4*>(dev_upload(c, tw3));
            M.vpt.w1T = M.vp_w1T; M.vpt.b1 = M.vp_b1; M.vpt.b2 = M.vp_b2;
            // request / answer granules of one launch (re-initialised before every launch that has helpers)
            c->vps_words = (size_t)VPS_MAX_SETS * VPS_PMAX * VPS_GRAN * (1 + VPS_SLICES);
            std::vector<unsigned long long> zero(c->vps_words + 1, 0ull);
            c->vps_mem = dev_upload(c, zero);
        }
        c->has_vposer = true;
    }
    // ---- GMM ----
    if (m->gmm_M > 0) {
        if (m->gmm_M > 8 || !m->gmm_means || !m->gmm_precisions || !m->gmm_nll_weights) return fail(c, MVFIT_E_ARG, "gmm: M <= 8 and all arrays required");
        M.gmm_M = m->gmm_M;
        std::vector<float> mu(m->gmm_means, m->gmm_means + m->gmm_M * 69), lw(m->gmm_M),
            pr((size_t)m->gmm_M * 69 * 72, 0.f), prT((size_t)m->gmm_M * 69 * 72, 0.f);
        for (int g = 0; g < m->gmm_M; ++g)
            for (int r = 0; r < 69; ++r)
                for (int q = 0; q < 69; ++q) {
                    const float v = m->gmm_precisions[((size_t)g * 69 + r) * 69 + q];
                    pr[((size_t)g * 69 + r) * 72 + q] = v;
                    prT[((size_t)g * 69 + q) * 72 + r] = v;
                }
        for (int i = 0; i < m->gmm_M; ++i) lw[i] = logf(m->gmm_nll_weights[i]);
        M.gmm_means = dev_upload(c, mu); M.gmm_prec = dev_upload(c, pr); M.gmm_precT = dev_upload(c, prT);
        M.gmm_lognw = dev_upload(c, lw);
        c->gmm_M = m->gmm_M;
    }
    for (void* p : c->allocs) if (!p) return fail(c, MVFIT_E_HIP, "device allocation failed");
    if (c->upload_failed) return fail(c, MVFIT_E_HIP, "copying the model constants to the device failed");
    HIP_OK(c, vertex_pass_configure());
    HIP_OK(c, hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device));
    HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(prep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds()));
    HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(closure_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds()));
    HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(closure_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds()));
    HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(fit_init_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds()));
    HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(fit_step_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_gram_lds()));
    HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(fit_step_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_gram_lds()));
    for (const void* k : {reinterpret_cast<const void*>(fit_persistent_kernel<false, false, true>),
                          reinterpret_cast<const void*>(fit_persistent_kernel<false, false, false>),
                          reinterpret_cast<const void*>(fit_persistent_kernel<true, false, false>),
                          reinterpret_cast<const void*>(fit_persistent_kernel<false, true, false>),
                          reinterpret_cast<const void*>(fit_persistent_kernel<false, true, true>),
                          reinterpret_cast<const void*>(fit_persistent_kernel<true, true, false>)})
        HIP_OK(c, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(persistent_lds(false), persistent_lds(true))));
    HIP_OK(c, hipHostMalloc(&c->h_done, 8));
    HIP_OK(c, hipDeviceSynchronize());
    return MVFIT_OK;
}

static void free_problem_buffers(mvfit_ctx* c) {
    drop_graph(c);
    void* ps[] = {c->d_camR, c->d_camt, c->d_camf, c->d_camc, c->d_gt, c->d_wc, c->P.coefH, c->P.coefT, c->P.Amat, c->P.tau,
                  c->P.vposed_sel, c->P.xs_sel, c->d_verts, c->d_obs, c->F.opt, c->F.pose, c->F.dirs, c->F.stps,
                  c->F.grow, c->F.gcol, c->F.rinv, c->F.stage_final, c->F.n_done, c->d_sdf_box, c->d_sdf_samp, c->d_sdf_entries,
                  c->d_sdf_adj, c->d_sdf_cull, c->d_sdf_boxpart, c->F.sdf_gate, c->F.sdf_tag, c->F.vp, c->d_gt3d, c->d_c3d};
    for (void* p : ps) if (p) hipFree(p);
    {
        void* rp[] = {c->ring.coefH, c->ring.Amat, c->ring.tau, c->ring.tag, c->ring.done_round, c->ring.stats, c->ring.pass_done};
        for (void* q : rp) if (q) hipFree(q);
        c->ring = AsyncRing{};
    }
    c->B = c->V = c->Bpad = 0;          // nothing is allocated: a failed re-allocation cannot leave a stale shape behind
    c->d_gt3d = c->d_c3d = nullptr;
    c->d_camR = c->d_camt = c->d_camf = c->d_camc = c->d_gt = c->d_wc = nullptr;
    c->d_obs = nullptr;
    c->P = DevPose{};
    c->d_verts = nullptr;
    c->F = FitBuffers{};
    c->d_sdf_box = nullptr; c->d_sdf_samp = nullptr; c->d_sdf_entries = nullptr; c->d_sdf_adj = nullptr; c->d_sdf_cull = nullptr;
    c->d_sdf_boxpart = nullptr;
    c->sdf_cull_refused = false;
}

extern "C" void mvfit_destroy(mvfit_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    free_problem_buffers(c);
    if (c->d_sdf_faces) hipFree(c->d_sdf_faces);
    if (c->d_sdf_op_ws) hipFree(c->d_sdf_op_ws);
    if (c->d_vp_log) hipFree(c->d_vp_log);
    for (void* p : c->allocs) if (p) hipFree(p);
    if (c->h_done) hipHostFree(c->h_done);
    for (hipEvent_t e : c->ev_done) if (e) hipEventDestroy(e);
    for (hipEvent_t e : c->ev_batch) if (e) hipEventDestroy(e);
    if (c->ev_init) hipEventDestroy(c->ev_init);
    if (c->pass_stream) hipStreamDestroy(c->pass_stream);
    if (c->h_async_done) hipHostFree(c->h_async_done);
    if (c->d_queue) hipFree(c->d_queue);
    for (auto& e : c->ev_vp) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto& e : c->ev_step) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    delete c;
}

extern "C" int mvfit_sync(mvfit_ctx* c) {
    if (!c) return MVFIT_E_ARG;
    HIP_OK(c, hipStreamSynchronize(c->stream));
    return MVFIT_OK;
}

extern "C" int mvfit_set_problems(mvfit_ctx* c, int B, int V, int cam_batched, const float* cam_R, const float* cam_t,
                                  const float* cam_f, const float* cam_c, const float* gt_xy, const float* w_conf) {
    if (!c) return MVFIT_E_ARG;
    if (B <= 0 || V <= 0 || V > MVFIT_MAX_VIEWS || !cam_R || !cam_t || !cam_f || !cam_c || !gt_xy || !w_conf)
        return fail(c, MVFIT_E_ARG, "set_problems: bad argument (B=%d V=%d, V <= %d)", B, V, MVFIT_MAX_VIEWS);
    HIP_OK(c, hipSetDevice(c->device));
    HIP_OK(c, hipStreamSynchronize(c->stream));
    if (B != c->B || V != c->V || (cam_batched != 0) != (c->Q.cam_batched != 0)) {
        free_problem_buffers(c);
        // caller-owned hook buffers were sized for the old batch (include/mvfit.h): switch both hooks off
        c->trace = nullptr; c->trace_cap = 0;
        c->capture_verts = nullptr; c->capture_round = -1;
        const size_t nc = cam_batched ? (size_t)B * V : (size_t)V;
        const int Bpad = (B + 31) / 32 * 32;
        HIP_OK(c, hipMalloc(&c->d_camR, nc * 9 * 4)); HIP_OK(c, hipMalloc(&c->d_camt, nc * 3 * 4));
        HIP_OK(c, hipMalloc(&c->d_camf, nc * 4)); HIP_OK(c, hipMalloc(&c->d_camc, nc * 2 * 4));
        HIP_OK(c, hipMalloc(&c->d_gt, (size_t)B * V * NKP * 2 * 4)); HIP_OK(c, hipMalloc(&c->d_wc, (size_t)B * V * NKP * 4));
        HIP_OK(c, hipMalloc(&c->P.coefT, (size_t)Bpad * KROWS * 4)); HIP_OK(c, hipMalloc(&c->P.Amat, (size_t)Bpad * 288 * 4));
        HIP_OK(c, hipMalloc(&c->P.tau, (size_t)Bpad * 4 * 4));
        HIP_OK(c, hipMalloc(&c->P.vposed_sel, (size_t)Bpad * NC_MAX * 4));
        HIP_OK(c, hipMalloc(&c->P.xs_sel, (size_t)Bpad * NC_MAX * 4));
        HIP_OK(c, hipMemset(c->P.coefT, 0, (size_t)Bpad * KROWS * 4));
        HIP_OK(c, hipMalloc(&c->P.coefH, (size_t)Bpad * KROWS * 4));
        HIP_OK(c, hipMemset(c->P.coefH, 0, (size_t)Bpad * KROWS * 4));
        HIP_OK(c, hipMalloc(&c->d_verts, (size_t)B * c->nv * 3 * 4));
        HIP_OK(c, hipMalloc(&c->d_obs, (size_t)B * sizeof(ObsBlock)));
        HIP_OK(c, hipMalloc(&c->F.opt, (size_t)B * sizeof(OptBlock)));
        HIP_OK(c, hipMalloc(&c->F.pose, (size_t)B * sizeof(PoseBlock)));
        HIP_OK(c, hipMalloc(&c->F.dirs, (size_t)B * LB_HIST * LB_D * 4));
        HIP_OK(c, hipMalloc(&c->F.stps, (size_t)B * LB_HIST * LB_D * 4));
        HIP_OK(c, hipMalloc(&c->F.rinv, (size_t)B * LB_RPACK * 4));
        HIP_OK(c, hipMalloc(&c->F.grow, (size_t)B * LB_GSIZE * 4));
        HIP_OK(c, hipMalloc(&c->F.gcol, (size_t)B * LB_GSIZE * 4));
        HIP_OK(c, hipMemset(c->F.grow, 0, (size_t)B * LB_GSIZE * 4));
        HIP_OK(c, hipMemset(c->F.gcol, 0, (size_t)B * LB_GSIZE * 4));
        HIP_OK(c, hipMalloc(&c->F.stage_final, (size_t)B * MVFIT_MAX_STAGES * 8));
        HIP_OK(c, hipMalloc(&c->F.n_done, 12));
        HIP_OK(c, hipMalloc(&c->F.sdf_gate, (size_t)B * 4));
        HIP_OK(c, hipMalloc(&c->F.sdf_tag, (size_t)Bpad * 4));
        HIP_OK(c, hipMalloc(&c->F.vp, (size_t)B * sizeof(VpBlock)));
        HIP_OK(c, hipMalloc(&c->d_gt3d, (size_t)B * NKP * 3 * 4));
        HIP_OK(c, hipMalloc(&c->d_c3d, (size_t)B * NKP * 4));
        c->B = B; c->V = V; c->Bpad = Bpad;      // only now: every buffer of this shape exists
    }
    const size_t nc = cam_batched ? (size_t)B * V : (size_t)V;
    HIP_OK(c, hipMemcpyAsync(c->d_camR, cam_R, nc * 9 * 4, hipMemcpyDefault, c->stream));
    HIP_OK(c, hipMemcpyAsync(c->d_camt, cam_t, nc * 3 * 4, hipMemcpyDefault, c->stream));
    HIP_OK(c, hipMemcpyAsync(c->d_camf, cam_f, nc * 4, hipMemcpyDefault, c->stream));
    HIP_OK(c, hipMemcpyAsync(c->d_camc, cam_c, nc * 2 * 4, hipMemcpyDefault, c->stream));
    HIP_OK(c, hipMemcpyAsync(c->d_gt, gt_xy, (size_t)B * V * NKP * 2 * 4, hipMemcpyDefault, c->stream));
    HIP_OK(c, hipMemcpyAsync(c->d_wc, w_conf, (size_t)B * V * NKP * 4, hipMemcpyDefault, c->stream));
    c->Q = DevProblems{B, V, cam_batched ? 1 : 0, c->d_camR, c->d_camt, c->d_camf, c->d_camc, c->d_gt, c->d_wc};
    hipLaunchKernelGGL(pack_obs_kernel, dim3(B), dim3(256), 0, c->stream, c->Q, c->d_obs);
    c->has_joints3d = false;
    HIP_OK(c, hipGetLastError());
    HIP_OK(c, hipStreamSynchronize(c->stream));
    return MVFIT_OK;
}

extern "C" int mvfit_set_joints3d(mvfit_ctx* c, const float* gt3d, const float* conf3d) {
    if (!c || !gt3d || !conf3d) return MVFIT_E_ARG;
    if (c->B == 0) return fail(c, MVFIT_E_STATE, "call mvfit_set_problems first");
    HIP_OK(c, hipSetDevice(c->device));
    HIP_OK(c, hipMemcpyAsync(c->d_gt3d, gt3d, (size_t)c->B * NKP * 3 * 4, hipMemcpyDefault, c->stream));
    HIP_OK(c, hipMemcpyAsync(c->d_c3d, conf3d, (size_t)c->B * NKP * 4, hipMemcpyDefault, c->stream));
    hipLaunchKernelGGL(pack_joints3d_kernel, dim3(c->B), dim3(64), 0, c->stream, (const float*)c->d_gt3d, (const float*)c->d_c3d, c->d_obs);
    HIP_OK(c, hipGetLastError());
    c->has_joints3d = true;
    return MVFIT_OK;
}

extern "C" int mvfit_set_sdf(mvfit_ctx* c, const int32_t* faces, int num_faces, int grid_size) {
    if (!c) return MVFIT_E_ARG;
    HIP_OK(c, hipSetDevice(c->device));
    HIP_OK(c, hipStreamSynchronize(c->stream));
    drop_graph(c);
    if (c->d_sdf_faces) { hipFree(c->d_sdf_faces); c->d_sdf_faces = nullptr; }
    if (c->d_sdf_cull) { hipFree(c->d_sdf_cull); c->d_sdf_cull = nullptr; }          // sized by the face count
    c->sdf_cull_refused = false;
    c->sdf_num_faces = 0; c->sdf_grid = 0;
    if (!faces || num_faces == 0) return MVFIT_OK;                 // term switched off
    if (num_faces < 0 || grid_size < 2 || grid_size > 1024)
        return fail(c, MVFIT_E_ARG, "mvfit_set_sdf: bad argument (num_faces=%d grid_size=%d)", num_faces, grid_size);
    if (c->nv > 8192) return fail(c, MVFIT_E_UNSUPPORTED, "the SDF term supports up to 8192 vertices (model has %d)", c->nv);
    HIP_OK(c, hipMalloc(&c->d_sdf_faces, (size_t)num_faces * 3 * 4));
    HIP_OK(c, hipMemcpy(c->d_sdf_faces, faces, (size_t)num_faces * 3 * 4, hipMemcpyDefault));
    std::vector<int32_t> h((size_t)num_faces * 3);
    HIP_OK(c, hipMemcpy(h.data(), c->d_sdf_faces, h.size() * 4, hipMemcpyDeviceToHost));
    for (int32_t vi : h)
        if (vi < 0 || vi >= c->nv) {
            hipFree(c->d_sdf_faces); c->d_sdf_faces = nullptr;
            return fail(c, MVFIT_E_ARG, "mvfit_set_sdf: face vertex index %d outside [0, %d)", (int)vi, c->nv);
        }
    c->sdf_num_faces = num_faces; c->sdf_grid = grid_size;
    return MVFIT_OK;
}

extern "C" int mvfit_sdf_term_read(mvfit_ctx* c, float* samples, float* sums) {
    if (!c) return MVFIT_E_ARG;
    if (!c->d_sdf_adj) return fail(c, MVFIT_E_STATE, "no interpenetration term has been evaluated yet");
    HIP_OK(c, hipSetDevice(c->device));
    if (samples) HIP_OK(c, hipMemcpyAsync(samples, c->d_sdf_samp, (size_t)c->B * c->nv * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    if (sums) HIP_OK(c, hipMemcpy2DAsync(sums, sizeof(float), c->d_sdf_adj, sizeof(SdfAdj), sizeof(float), c->B, hipMemcpyDeviceToDevice, c->stream));
    return MVFIT_OK;
}

// work buffers of the SDF term for the current batch
static int ensure_sdf_buffers(mvfit_ctx* c) {
    // all faces (or any list too long for the staged walk): the per-round face lists of sdf_term.hip.
    // mvfit_options::sdf_face_lists = 0 keeps the brute-force kernel (the check of the culled one).
    if (c->d_sdf_cull && !c->opt.sdf_face_lists) {          // switched off since the workspace was made
        HIP_OK(c, hipStreamSynchronize(c->stream));
        hipFree(c->d_sdf_cull);
        c->d_sdf_cull = nullptr;
    }
    c->sdf_term_path = c->d_sdf_cull ? 1 : 0;
    if (!c->d_sdf_cull && c->opt.sdf_face_lists && c->sdf_num_faces >= sdf_cull_min_faces() && !c->sdf_cull_refused) {
        // (11.6 MB of lists, records and bins per problem at 13,776 faces: a batch whose workspace would not fit keeps the
        // walk - decided ONCE per (batch, face list): the refusal is remembered (and reported by mvfit_sdf_info) instead of
        // querying the free memory on every fit)
        size_t free_b = 0, total_b = 0;
        HIP_OK(c, hipMemGetInfo(&free_b, &total_b));
        if (sdf_cull_bytes(c->B, c->sdf_num_faces) < free_b / 2) {
            HIP_OK(c, hipMalloc(&c->d_sdf_cull, sdf_cull_bytes(c->B, c->sdf_num_faces)));
            HIP_OK(c, hipMemset(reinterpret_cast<unsigned char*>(c->d_sdf_cull) + sdf_cull_zero_offset(c->B, c->sdf_num_faces), 0,
                                sdf_cull_zero_bytes(c->B)));
            c->sdf_term_path = 1;
        } else {
            c->sdf_cull_refused = true;
            c->sdf_term_path = 2;
        }
    } else if (!c->d_sdf_cull && c->sdf_cull_refused) c->sdf_term_path = 2;
    if (c->d_sdf_adj) return MVFIT_OK;
    HIP_OK(c, hipMalloc(&c->d_sdf_box, (size_t)c->B * sizeof(SdfBox)));
    HIP_OK(c, hipMalloc(&c->d_sdf_samp, (size_t)c->B * c->nv * sizeof(float4)));
    HIP_OK(c, hipMalloc(&c->d_sdf_entries, sdf_work_bytes(c->B, c->nv)));      // entry lists + slice partials + heads + tickets
    HIP_OK(c, hipMemset(reinterpret_cast<unsigned char*>(c->d_sdf_entries) + sdf_ticket_offset(c->B, c->nv), 0, (size_t)c->B * sizeof(int)));
    HIP_OK(c, hipMalloc(&c->d_sdf_adj, (size_t)c->B * sizeof(SdfAdj)));
    HIP_OK(c, hipMalloc(&c->d_sdf_boxpart, (size_t)c->Bpad * c->M.ntiles * 6 * 8));
    return MVFIT_OK;
}

// the per-round pass over problems [b_lo, b_hi) runs as lbs_vertex_pass_split_kernel (which writes the tile keys of the term's box when
// DevPose::box_part is set) when the split-fp16 basis exists and the launch is one 32-problem chunk per workgroup
static bool pass_writes_box_parts(const mvfit_ctx* c, int b_lo, int b_hi) {
    const int chunks = (b_hi + 31) / 32 - b_lo / 32;
    return c->M.bs_h2 != nullptr && c->d_sdf_boxpart != nullptr && c->sdf_num_faces <= 128 && (chunks == 1 || c->opt.pass_kernel == 1);
}

static int run_sdf_term(mvfit_ctx* c, const float* verts, const int* gate, hipStream_t st) {
    hipError_t e = launch_sdf_term(c->M, c->P, verts, c->B, c->d_sdf_faces, c->sdf_num_faces, c->sdf_grid, gate, c->d_sdf_box,
                                   c->d_sdf_samp, c->d_sdf_entries, c->d_sdf_adj, st, c->d_sdf_cull);
    if (e != hipSuccess) return fail(c, MVFIT_E_HIP, "sdf term launch: %s", hipGetErrorString(e));
    return MVFIT_OK;
}

static int check_flags(mvfit_ctx* c, uint32_t flags) {
    if ((flags & MVFIT_F_USE_3D) && !c->has_joints3d) return fail(c, MVFIT_E_STATE, "MVFIT_F_USE_3D set but mvfit_set_joints3d was not called");
    if ((flags & MVFIT_F_VPOSER) && !c->has_vposer) return fail(c, MVFIT_E_STATE, "MVFIT_F_VPOSER set but the model has no VPoser decoder");
    if ((flags & MVFIT_F_PRIOR_GMM) && c->gmm_M == 0) return fail(c, MVFIT_E_STATE, "MVFIT_F_PRIOR_GMM set but the model has no GMM");
    return MVFIT_OK;
}

static DevWeights to_dev(const mvfit_weights& w) {
    DevWeights d;
    d.data_w2 = w.data_weight * w.data_weight;
    d.pose_w = w.body_pose_weight; d.shape_w = w.shape_weight; d.bend_w = w.bending_prior_weight;
    d.coll_w = w.coll_loss_weight; d.rho2 = w.rho * w.rho; d.flags = w.flags; d.pad = 0;
    return d;
}

static void prof_begin(mvfit_ctx* c, std::vector<std::pair<hipEvent_t, hipEvent_t>>& evs) {
    if (!c->profile || evs.size() >= 4096) return;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, c->stream);
    evs.emplace_back(a, b);
}
static void prof_end(mvfit_ctx* c, std::vector<std::pair<hipEvent_t, hipEvent_t>>& evs) {
    if (!c->profile || evs.empty()) return;
    hipEventRecord(evs.back().second, c->stream);
}

static int run_vertex_pass(mvfit_ctx* c, float* verts) {
    prof_begin(c, c->ev_vp);
    hipError_t e = launch_vertex_pass(c->M, c->P, c->B, verts, c->opt.pass_kernel, c->stream);
    prof_end(c, c->ev_vp);
    if (e != hipSuccess) return fail(c, MVFIT_E_HIP, "vertex pass launch: %s", hipGetErrorString(e));
    return MVFIT_OK;
}

extern "C" int mvfit_vertices(mvfit_ctx* c, const float* params, uint32_t flags, float* verts, float* joints) {
    if (!c || !params || !verts) return MVFIT_E_ARG;
    if (c->B == 0) return fail(c, MVFIT_E_STATE, "call mvfit_set_problems first");
    int rc = check_flags(c, flags);
    if (rc) return rc;
    HIP_OK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(prep_kernel, dim3(c->B), dim3(STEP_NT), step_lds(), c->stream, c->M, (const ObsBlock*)c->d_obs, c->P, params, flags,
                       (float*)nullptr);
    HIP_OK(c, hipGetLastError());
    rc = run_vertex_pass(c, verts);
    if (rc) return rc;
    if (joints) {
        hipLaunchKernelGGL(joints_kernel, dim3(c->B), dim3(64), 0, c->stream, c->M, (const float*)verts, joints);
        HIP_OK(c, hipGetLastError());
    }
    return MVFIT_OK;
}

extern "C" int mvfit_full_pose(mvfit_ctx* c, const float* params, uint32_t flags, float* full_pose) {
    if (!c || !params || !full_pose) return MVFIT_E_ARG;
    if (c->B == 0) return fail(c, MVFIT_E_STATE, "call mvfit_set_problems first");
    int rc = check_flags(c, flags);
    if (rc) return rc;
    HIP_OK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(prep_kernel, dim3(c->B), dim3(STEP_NT), step_lds(), c->stream, c->M, (const ObsBlock*)c->d_obs, c->P, params, flags,
                       full_pose);
    HIP_OK(c, hipGetLastError());
    return MVFIT_OK;
}

// decoder helpers (vposer_service.h): launch geometry, see launch_persistent
constexpr int kVpsSets = 8;                                             // sets of a launch with more than 32 problems
constexpr int kVpsHelpers = kVpsSets * VPS_SLICES;                      // 64 CUs
constexpr int kVpsMaxSparse = 160, kVpsMaxAsync = 96;
static_assert(kVpsMaxSparse <= kVpsSets * VPS_PMAX && kVpsMaxSparse + kVpsHelpers <= 256 && 32 + VPS_MAX_SETS * VPS_SLICES <= 160, "all workgroups resident");

static int closure_via_helpers(mvfit_ctx* c, const mvfit_weights* w, const float* params, float* loss, float* grad,
                               float* verts, float* joints);

extern "C" int mvfit_closure(mvfit_ctx* c, const mvfit_weights* w, const float* params, float* loss, float* grad,
                             float* verts, float* joints) {
    if (!c || !w || !params || !loss) return MVFIT_E_ARG;
    if (c->B == 0) return fail(c, MVFIT_E_STATE, "call mvfit_set_problems first");
    int rc = check_flags(c, w->flags);
    if (rc) return rc;
    const bool sdf = w->coll_loss_weight > 0.f;
    if (sdf && !c->sdf_num_faces)
        return fail(c, MVFIT_E_STATE, "coll_loss_weight > 0 needs the SDF term's faces: call mvfit_set_sdf first");
    HIP_OK(c, hipSetDevice(c->device));
    if (c->opt.closure_vposer_helpers && (w->flags & MVFIT_F_VPOSER) && c->vps_mem && !sdf)
        return closure_via_helpers(c, w, params, loss, grad, verts, joints);
    float* vbuf = verts ? verts : c->d_verts;
    // the interpenetration term reads every vertex: it forces the vertex pass
    const bool sparse = (w->flags & MVFIT_F_SPARSE_VERTS) != 0 && !sdf;
    if (!sparse || verts) {
        hipLaunchKernelGGL(prep_kernel, dim3(c->B), dim3(STEP_NT), step_lds(), c->stream, c->M, (const ObsBlock*)c->d_obs, c->P, params, w->flags,
                           (float*)nullptr);
        HIP_OK(c, hipGetLastError());
        rc = run_vertex_pass(c, vbuf);
        if (rc) return rc;
    }
    if (sdf) {
        rc = ensure_sdf_buffers(c);
        if (!rc) rc = run_sdf_term(c, vbuf, nullptr, c->stream);
        if (rc) return rc;
    }
    prof_begin(c, c->ev_step);
    hipLaunchKernelGGL(closure_kernel<false>, dim3(c->B), dim3(STEP_NT), step_lds(), c->stream, c->M, (const ObsBlock*)c->d_obs, c->V,
                       to_dev(*w), c->P, params,
                       sparse ? 0 : 1, loss, grad, joints, sdf ? (const SdfAdj*)c->d_sdf_adj : (const SdfAdj*)nullptr);
    prof_end(c, c->ev_step);
    HIP_OK(c, hipGetLastError());
    return MVFIT_OK;
}

// Test route of mvfit_closure (MVFIT_CLOSURE_VP_HELPERS=1, read per call; VPoser flag, no SDF term): the closure decodes
// the body pose on helper workgroups of its own launch - the decoder of the production single-launch fits
// (vposer_service.h), whose summation order differs from the in-workgroup decoder - and, like those fits, evaluates the
// objective from the vertices it computes itself while the full vertex pass runs on the operands it published.
static int closure_via_helpers(mvfit_ctx* c, const mvfit_weights* w, const float* params, float* loss, float* grad,
                               float* verts, float* joints) {
    const int n = c->B;
    if (n > kVpsMaxSparse) return fail(c, MVFIT_E_ARG, "MVFIT_CLOSURE_VP_HELPERS: at most %d problems (all workgroups resident)", kVpsMaxSparse);
    DevModel M = c->M;
    const int cap = n <= 32 ? VPS_MAX_SETS : kVpsSets;
    const int nsets = std::max((n + VPS_PMAX - 1) / VPS_PMAX, std::min(cap, n));
    HIP_OK(c, hipMemsetAsync(c->vps_mem, 0, c->vps_words * 8 + 8, c->stream));
    M.vps.req = c->vps_mem;
    M.vps.resp = c->vps_mem + (size_t)VPS_MAX_SETS * VPS_PMAX * VPS_GRAN;
    M.vps.stat = reinterpret_cast<unsigned*>(c->vps_mem + c->vps_words);
    M.vps.nsets = nsets;
    M.vps.nprob = n;
    M.vps.fault = 0;
    c->vps_stats[0] = 1;
    hipLaunchKernelGGL(closure_kernel<true>, dim3(n + nsets * VPS_SLICES), dim3(STEP_NT), step_lds(), c->stream, M,
                       (const ObsBlock*)c->d_obs, c->V, to_dev(*w), c->P, params, 0, loss, grad, joints, (const SdfAdj*)nullptr);
    HIP_OK(c, hipGetLastError());
    if (verts) return run_vertex_pass(c, verts);
    return MVFIT_OK;
}

static int make_opts(mvfit_ctx* c, const mvfit_lbfgs_opts* o, uint32_t flags, LbOpts& O) {
    if (o->max_iter <= 0 || o->history <= 0 || o->history > MVFIT_HISTORY || o->maxiters <= 0 || o->num_stages <= 0 ||
        o->num_stages > MVFIT_MAX_STAGES)
        return fail(c, MVFIT_E_ARG, "bad lbfgs options");
    memset(&O, 0, sizeof(O));
    O.lr = o->lr; O.tol_grad = o->tolerance_grad; O.tol_change = o->tolerance_change; O.ftol = o->ftol; O.gtol = o->gtol;
    O.max_iter = o->max_iter; O.max_eval = o->max_iter * 5 / 4; O.history = o->history; O.maxiters = o->maxiters;
    O.num_stages = o->num_stages;
    O.reuse_outer = (flags & MVFIT_F_REUSE_OUTER_VALUE) ? 1 : 0;
    // parameter tensors that take part in the gtol test (fitting.py:115-116): requires_grad ones,
    // as index ranges of the compact optimiser vector (reference final_params order)
    int n = 0;
    auto add = [&](int lo, int hi) { O.seg_lo[n] = lo; O.seg_hi[n] = hi; ++n; };
    if (flags & MVFIT_F_VPOSER) {
        if (!(flags & MVFIT_F_FIX_SHAPE)) add(0, 10);
        add(10, 13); add(13, 16);
        if (!(flags & MVFIT_F_FIX_SCALE)) add(16, 17);
        add(17, 49);
    } else {
        if (!(flags & MVFIT_F_FIX_SHAPE)) add(0, 10);
        add(10, 13); add(13, 82); add(82, 85);
        if (!(flags & MVFIT_F_FIX_SCALE)) add(85, 86);
    }
    O.nseg = n;
    return MVFIT_OK;
}

// rounds of (vertex pass, step kernel) between two looks at the done counter, replayed as one graph
static const int kGraphRounds = 24;

static int ensure_round_graph(mvfit_ctx* c, const StageWeights& SW, const LbOpts& O) {
    std::vector<unsigned char> key(sizeof(SW) + sizeof(O) + sizeof(DevPose) + sizeof(FitBuffers) + sizeof(DevProblems) + sizeof(int));
    unsigned char* k = key.data();
    memcpy(k, &SW, sizeof(SW)); k += sizeof(SW);
    memcpy(k, &O, sizeof(O)); k += sizeof(O);
    memcpy(k, &c->P, sizeof(DevPose)); k += sizeof(DevPose);
    memcpy(k, &c->F, sizeof(FitBuffers)); k += sizeof(FitBuffers);
    memcpy(k, &c->Q, sizeof(DevProblems)); k += sizeof(DevProblems);
    memcpy(k, &c->opt.pass_kernel, sizeof(int));
    if (c->round_graph && key == c->graph_key) return MVFIT_OK;
    drop_graph(c);
    hipStream_t cs;
    HIP_OK(c, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
        // rounds with the SDF term: the pass writes its tiles' keys of the term's bounding box (single-chunk split kernel), the
        // front kernel reduces the box from them
        DevPose Pg = c->P;
        if (c->F.sdf_adj && pass_writes_box_parts(c, 0, c->B)) Pg.box_part = c->d_sdf_boxpart;
        for (int r = 0; r < kGraphRounds && e == hipSuccess; ++r) {
            e = launch_vertex_pass(c->M, Pg, c->B, c->d_verts, c->opt.pass_kernel, cs);
            if (e == hipSuccess && c->F.sdf_adj)
                e = launch_sdf_term(c->M, c->P, c->d_verts, c->B, c->d_sdf_faces, c->sdf_num_faces, c->sdf_grid, c->F.sdf_gate,
                                    c->d_sdf_box, c->d_sdf_samp, c->d_sdf_entries, c->d_sdf_adj, cs, c->d_sdf_cull, nullptr, 0u, Pg.box_part);
            hipLaunchKernelGGL(O.reuse_outer ? fit_step_kernel<true> : fit_step_kernel<false>, dim3(c->B), dim3(STEP_NT), step_gram_lds(), cs, c->M, (const ObsBlock*)c->d_obs, c->V, SW, O,
                               c->P, c->F);
        }
        hipError_t e2 = hipStreamEndCapture(cs, &g);
        if (e == hipSuccess) e = e2;
    }
    if (e == hipSuccess) e = hipGraphInstantiate(&c->round_graph, g, nullptr, nullptr, 0);
    if (g) hipGraphDestroy(g);
    hipStreamDestroy(cs);
    if (e != hipSuccess) { c->round_graph = nullptr; return fail(c, MVFIT_E_HIP, "round graph: %s", hipGetErrorString(e)); }
    c->graph_key = key;
    c->graph_rounds = kGraphRounds;
    return MVFIT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Asynchronous full-mode fit (default when no SDF term is active and the batch leaves CUs for the passes).
//
// The objective reads 69 of the 6890 vertices and the optimiser kernel evaluates those itself (sparse_forward: the
// same arithmetic as the pass on the selected vertices), so the 6890-vertex LBS pass of a trial point is not on the
// optimiser's critical path - but every closure still gets its full pass, like the reference's return_verts=True:
//   ctx stream   ONE fit_persistent_kernel launch (one workgroup per problem, L-BFGS history in LDS) runs the whole
//                staged fit; in closure round r it publishes the pose operands of the trial point into ring slot
//                r % kRingSlots (write-through stores + a per-problem tag);
//   pass stream  one lbs_vertex_pass launch per closure round, queued ahead by the host in batches of kPassBatch;
//                the pass of round r waits (bounded spin on the tags of its 32 problems) until the optimiser has
//                published round r, then computes all 6890 vertices of those trial points on the CUs the optimiser
//                does not occupy - concurrently with the optimiser's own work on closure r.
// Nothing the optimiser does waits on a pass (one-directional hand-off: no deadlock; a pass that times out just runs
// on whatever the slot holds).  Passes whose 32 problems have all finished return at once.  stats: passes run /
// skipped / operands overwritten before their pass could read them (ring too short for the drift between problems;
// expected 0) / timed out (expected 0).
// Measured alternatives on configs[1]: chaining pass -> step per round costs pass + step (37 us per round, 766 k
// closures/s); forking the two inside one hipGraph round overlaps them but the cross-queue join costs ~12 us per round
// (632 k); windows of 24 rounds of the persistent kernel followed by their 24 passes lose the lock-step at every
// window end (947 k).
// ---------------------------------------------------------------------------------------------------------
static const int kRingSlots = 128;
static const int kPassBatch = 24;
static const int kAsyncMaxB = 160;        // one CU per problem for the optimiser: leave >= 96 CUs to the passes
static const int kResidentMaxB = 128;     // with the resident pass: its workgroups (ntiles / 2 at this size) hold a CU each for the whole fit
static const int kPassWords = 512;        // back-pressure words of the ring (one per resident-pass workgroup; the gate kernels use word 0)
static const int kVpLogRounds = 1024;     // mvfit_profile: rounds of the resident pass that are stamped

// The ring is sized by the SUB-BATCH (rb problems, a multiple of 32), not by the batch: only one sub-batch uses it at a
// time (128 slots x 2.06 KB per problem: 34 MB at 128 problems whatever the batch size).  Everything indexed by ring slot
// takes sub-batch-relative problem indices; done_round stays indexed by the global problem index.
static int ensure_async(mvfit_ctx* c, int rb) {
    if (!c->pass_stream) {
        HIP_OK(c, hipStreamCreateWithFlags(&c->pass_stream, hipStreamNonBlocking));
        for (hipEvent_t& e : c->ev_batch) HIP_OK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_OK(c, hipEventCreateWithFlags(&c->ev_init, hipEventDisableTiming));
        HIP_OK(c, hipHostMalloc(&c->h_async_done, 64));
        HIP_OK(c, hipMalloc(&c->d_queue, 64));
    }
    AsyncRing& R = c->ring;
    if (R.tag && R.Bpad >= rb) return MVFIT_OK;
    if (R.tag) {
        HIP_OK(c, hipStreamSynchronize(c->stream));
        HIP_OK(c, hipStreamSynchronize(c->pass_stream));
        void* rp[] = {R.coefH, R.Amat, R.tau, R.tag, R.done_round, R.stats, R.pass_done};
        for (void* q : rp) if (q) hipFree(q);
        R = AsyncRing{};
    }
    const size_t Bp = (size_t)rb;
    R.nslots = kRingSlots; R.Bpad = rb;
    HIP_OK(c, hipMalloc(&R.coefH, kRingSlots * Bp * KROWS * 4));
    HIP_OK(c, hipMalloc(&R.Amat, kRingSlots * Bp * 288 * 4));
    HIP_OK(c, hipMalloc(&R.tau, kRingSlots * Bp * 4 * 4));
    HIP_OK(c, hipMalloc(&R.tag, kRingSlots * Bp * 4));
    HIP_OK(c, hipMalloc(&R.done_round, (size_t)c->Bpad * 4));
    HIP_OK(c, hipMalloc(&R.stats, 4 * 4));
    HIP_OK(c, hipMalloc(&R.pass_done, 4 * kPassWords));
    HIP_OK(c, hipMemset(R.coefH, 0, kRingSlots * Bp * KROWS * 4));
    HIP_OK(c, hipMemset(R.Amat, 0, kRingSlots * Bp * 288 * 4));
    HIP_OK(c, hipMemset(R.tau, 0, kRingSlots * Bp * 4 * 4));
    void* dp = nullptr;
    HIP_OK(c, hipHostGetDevicePointer(&dp, c->h_async_done, 0));
    R.host_done = reinterpret_cast<int*>(dp);
    return MVFIT_OK;
}

// Decoder helpers (vposer_service.h) ride on the single-launch fits with the VPoser prior: min(8, n) sets of 8 helper
// workgroups behind the n problems' ones, every set serving the problems b with b % nsets == s.  All workgroups of the
// launch must be resident at once (the problems wait for their helpers' answers): such fits run in sub-batches of at most
// kVpsMaxSparse problems (objective vertices only) / kVpsMaxAsync (asynchronous: the passes keep >= 96 CUs) - every
// problem's arithmetic is the same whatever the slicing.  mvfit_options::vposer_helpers = 0 keeps the decoder in the
// problems' own workgroups (another summation order: results differ in the last bits).
static bool vps_enabled(const mvfit_ctx* c, const StageWeights& SW) {
    return (SW.w[0].flags & MVFIT_F_VPOSER) && c->vps_mem && c->opt.vposer_helpers != 0;
}

static int resident_tiles_per_wg(const mvfit_ctx* c, int opt_grid);

// decoder-helper sets a single-launch fit of n problems carries (0: none).  with_passes: the launch is an asynchronous fit
// (vertex passes beside it) - the automatic choice then keeps the RESIDENT pass: 16 sets next to <= 32 problems are 160
// optimiser-kernel workgroups, which leave no room for the pass's 108; 8 sets (96 workgroups) do, and measure the same closure
// rate (the mode is bound by the decoder hand-offs, profiles/r5_progress.md) - so the shipped yaml's default mode no longer
// runs its passes as per-round launches (round 6).  Results do not depend on the number of sets (fixed summation order).
static int persistent_nsets(const mvfit_ctx* c, const StageWeights& SW, int n, bool with_passes) {
    if (!(vps_enabled(c, SW) && n <= kVpsMaxSparse)) return 0;
    // few problems: 16 sets (two problems per helper at 32: less queueing behind another problem's request)
    const int cap = n <= 32 ? VPS_MAX_SETS : kVpsSets;
    // at least ceil(n / VPS_PMAX) sets: a set has VPS_PMAX request / answer slots (the knob cannot push problems past them)
    const int need = (n + VPS_PMAX - 1) / VPS_PMAX;
    int want = c->opt.vposer_sets > 0 ? std::min(c->opt.vposer_sets, cap) : cap;
    if (c->opt.vposer_sets <= 0 && with_passes && want > kVpsSets) {
        const int full = std::max(need, std::min(want, n)), half = std::max(need, std::min(kVpsSets, n));
        if (!resident_tiles_per_wg(c, n + full * VPS_SLICES) && resident_tiles_per_wg(c, n + half * VPS_SLICES)) want = kVpsSets;
    }
    return std::max(need, std::min(want, n));
}
// workgroups of that launch: one per problem + the helpers behind them, one CU each (LDS)
static int persistent_grid(const mvfit_ctx* c, const StageWeights& SW, int n, bool with_passes) {
    return n + persistent_nsets(c, SW, n, with_passes) * VPS_SLICES;
}

static int launch_persistent(mvfit_ctx* c, const StageWeights& SW, const LbOpts& O, int cap, const AsyncRing& R, int b_lo,
                             int b_hi, int done_target, int pause_stage, bool sdfs = false, int* queue = nullptr, int b_end = 0) {
    const int n = b_hi - b_lo;
    DevModel M = c->M;
    int grid = n;
    if (const int nsets = persistent_nsets(c, SW, n, R.tag != nullptr)) {
        HIP_OK(c, hipMemsetAsync(c->vps_mem, 0, c->vps_words * 8, c->stream));
        M.vps.req = c->vps_mem;
        M.vps.resp = c->vps_mem + (size_t)VPS_MAX_SETS * VPS_PMAX * VPS_GRAN;
        M.vps.stat = reinterpret_cast<unsigned*>(c->vps_mem + c->vps_words);
        M.vps.nsets = nsets;
        M.vps.nprob = n;
        M.vps.fault = debug_hook("MVFIT_VP_FAULT") != 0;                                     // test hook (hooks build only): helpers that never answer
        grid = n + nsets * VPS_SLICES;
        c->vps_stats[0] += 1;
    }
    const bool lean = !(SW.w[0].flags & (MVFIT_F_VPOSER | MVFIT_F_PRIOR_GMM | MVFIT_F_USE_3D));    // (flags are the same in all stages)
    // (service launches - the stages with the SDF term, mvfit_options::sdf_service - have their own instantiations: the other
    // kernels carry no trace of the service; MVFIT_F_REUSE_OUTER_VALUE fits keep the chained rounds, see mvfit_fit)
    auto kern = sdfs ? (M.vps.nsets ? fit_persistent_kernel<true, false, false, true> : fit_persistent_kernel<false, false, false, true>)
                : queue ? (lean ? fit_persistent_kernel<false, false, true, false, true> : fit_persistent_kernel<false, false, false, false, true>)
                : M.vps.nsets ? (O.reuse_outer ? fit_persistent_kernel<true, true, false> : fit_persistent_kernel<true, false, false>)
                : O.reuse_outer ? (lean ? fit_persistent_kernel<false, true, true> : fit_persistent_kernel<false, true, false>)
                : lean ? fit_persistent_kernel<false, false, true> : fit_persistent_kernel<false, false, false>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(STEP_NT), persistent_lds((SW.w[0].flags & MVFIT_F_VPOSER) != 0), c->stream, M,
                       (const ObsBlock*)c->d_obs, c->V, SW, O, c->P, c->F, cap, R, b_lo, done_target, pause_stage, queue, b_end);
    HIP_OK(c, hipGetLastError());
    return MVFIT_OK;
}

// How the passes of an asynchronous (sub-batch) fit run: tiles per workgroup of the RESIDENT pass (one launch per fit, basis
// stationary in registers; its ceil(ntiles / tpw) workgroups must all be resident next to the optimiser's `opt_grid` ones,
// every one of them a CU), or 0 = a gate + a pass launch per closure round (dense skinning rows, exact-fp32 contraction,
// launches that do not leave the CUs - e.g. 160 optimiser + helper workgroups with the VPoser prior).
// mvfit_options::resident_pass = 0 / 1 / 2 forces the choice (a forced value that does not fit can stall the fit: the
// optimiser then waits 20 ms for the ring once and stops waiting; fit() reports the lost passes).
// workgroups of the resident pass: form 1 / 2 = tiles per workgroup; 3 = two tiles per workgroup split into contraction and
// worker waves (lbs_vertex_pass_resident_roles_kernel)
static int resident_grid(const mvfit_ctx* c, int form) {
    if (!form) return 0;
    const int tiles = form >= 2 ? 2 : 1;
    return (c->M.ntiles + tiles - 1) / tiles;
}

static int resident_tiles_per_wg(const mvfit_ctx* c, int opt_grid) {
    if (!c->M.bs_h2 || !c->M.wsp_w || (c->M.nv & 1)) return 0;      // (the resident pass stores vertex pairs: even vertex count)
    if (c->opt.resident_pass >= 0) return c->opt.resident_pass == 2 ? 3 : c->opt.resident_pass;      // (form 2 was dropped: it maps to 3)
    // automatic mode: a fit on this ctx whose resident workgroups (or whose optimiser) gave up waiting has shown that the launch
    // does not get the CUs the choice assumes (a shared device, a CU mask): later fits use the per-round launches
    if (c->resident_auto_off) return 0;
    const int room = c->n_cu - 4 - opt_grid;           // (4 CUs of slack: nothing in HIP promises that every CU takes a workgroup)
    if (c->M.ntiles <= room) return 1;
    if ((c->M.ntiles + 1) / 2 <= room) return 3;       // two tiles per workgroup, contraction / worker roles (15-20 % faster than form 2)
    return 0;
}

// sdf_service: the launch continues fits that are paused in front of their first stage with the SDF term; every closure round
// of such a stage asks for the term (closure_device.h: publish_sdf_request): per round the pass stream carries gate -> vertex
// pass -> the term's kernels (launch_sdf_term, whose pull-back publishes the answer tag) - per-round launches by construction
// (the term's kernels need the round's vertices complete: a launch boundary).
static int fit_async(mvfit_ctx* c, const StageWeights& SW, const LbOpts& O, int cap, int* seen_out, int pause_stage = MVFIT_MAX_STAGES + 1,
                     bool sdf_service = false) {
    const int B = c->B;
    // More problems than the optimiser gets CUs (one workgroup per CU, >= 96 CUs left to the passes): time-sliced in
    // sub-batches of whole 32-problem chunks, one after the other - every sub-batch is the same asynchronous fit (problems
    // are independent: the results do not depend on the slicing, tests/test_gpu_large_batch.py).
    // (the sub-batch size follows from the form the passes REALLY take for the optimiser grid it gives: 128 problems beside the
    // resident pass, 160 when the passes run as per-round launches anyway - dense skinning rows, a forced resident_pass = 0, ...)
    const bool dbg_nopass = debug_hook("MVFIT_DEBUG_NOPASS") != 0;          // (hooks build only)
    auto sub_batch = [&](int maxb) { const int nsub = (B + maxb - 1) / maxb; return ((B + nsub - 1) / nsub + 31) / 32 * 32; };
    int per = sub_batch(vps_enabled(c, SW) ? kVpsMaxAsync : kResidentMaxB);
    int tpw = (dbg_nopass || sdf_service) ? 0 : resident_tiles_per_wg(c, persistent_grid(c, SW, std::min(B, per), true));
    if (!tpw && !vps_enabled(c, SW)) per = sub_batch(kAsyncMaxB);
    // Round 6: more problems than optimiser workgroups run as ONE launch with a work queue when the resident pass serves it (its
    // workgroups follow a ring row through the problems it takes; the per-round launch kernels address problems by row): `per`
    // rows, a row takes the next unfitted problem when its own has finished - no serial sub-batches, no idle tail per sub-batch.
    // (Not with decoder helpers: their request slots belong to problems; not in service launches.)
    const bool refill = tpw != 0 && !vps_enabled(c, SW) && !sdf_service && B > per && c->opt.work_queue != 0 && !O.reuse_outer;
    if (refill) per = kResidentMaxB;
    const int launch_cap = refill ? (int)std::min<long long>((long long)cap * ((B + per - 1) / per + 1), 1 << 30) : cap;
    int rc = ensure_async(c, per);
    if (rc) return rc;
    AsyncRing R = c->ring;
    const size_t rb = (size_t)R.Bpad;                       // ring stride in problems (>= per)
    volatile int* h_done = c->h_async_done;
    // polled words: re-initialised every call
    HIP_OK(c, hipMemsetAsync(R.done_round, 0xff, (size_t)c->Bpad * 4, c->stream));
    HIP_OK(c, hipMemsetAsync(R.stats, 0, 16, c->stream));
    const int res_grid = resident_grid(c, tpw);
    if (res_grid > kPassWords) return fail(c, MVFIT_E_ARG, "resident vertex pass: %d workgroups > %d back-pressure words", res_grid, kPassWords);
    if (tpw && per > kResidentMaxB) return fail(c, MVFIT_E_ARG, "resident vertex pass: %d ring rows > %d", per, kResidentMaxB);
    c->resident_tpw = tpw;
    R.npass = tpw ? res_grid : 1;
    c->res_rounds = 0; c->res_span_ms = c->res_busy_ms = c->res_slowest_ms = 0.0;
    const bool log_on = tpw && c->profile;
    if (log_on) {
        const size_t words = (size_t)kVpLogRounds * res_grid * 2;
        if (c->vp_log_words < words) {
            if (c->d_vp_log) hipFree(c->d_vp_log);
            c->d_vp_log = nullptr; c->vp_log_words = 0;
            HIP_OK(c, hipMalloc(&c->d_vp_log, words * 8));
            c->vp_log_words = words;
        }
    }
    for (int b_lo = 0; b_lo < B; b_lo += refill ? B : per) {
        const int b_hi = std::min(B, b_lo + per);                 // (refill: the rows of the one launch)
        const int n_target = refill ? B : b_hi - b_lo;            // problems that leave this launch
        *h_done = 0;
        // per sub-batch: its tags (the slots are reused by other problems), the pass counter and the count of problems
        // that left the launch (finished or paused) - a sub-batch that stops at the round cap does not keep the later ones
        // from seeing theirs complete.  (The ctx stream is behind the previous sub-batch's last passes here.)
        HIP_OK(c, hipMemsetAsync(R.tag, 0, (size_t)kRingSlots * rb * 4, c->stream));
        HIP_OK(c, hipMemsetAsync(R.pass_done, 0, 4 * kPassWords, c->stream));
        HIP_OK(c, hipMemsetAsync(c->F.n_done + 1, 0, 4, c->stream));
        if (sdf_service) {
            // per sub-batch: no answer yet, and no gate open - a problem opens its own in front of every round's tag (the gates of
            // the problems outside this sub-batch stay shut: the term's kernels cover all problems up to b_hi)
            HIP_OK(c, hipMemsetAsync(c->F.sdf_tag, 0, (size_t)c->Bpad * 4, c->stream));
            HIP_OK(c, hipMemsetAsync(c->F.sdf_gate, 0, (size_t)B * 4, c->stream));
        }
        if (log_on) HIP_OK(c, hipMemsetAsync(c->d_vp_log, 0, c->vp_log_words * 8, c->stream));      // (a profiled fit keeps the last sub-batch's stamps)
        HIP_OK(c, hipEventRecord(c->ev_init, c->stream));
        HIP_OK(c, hipStreamWaitEvent(c->pass_stream, c->ev_init, 0));
        if (refill) {
            c->h_queue0 = b_hi;                                   // problems [0, rows) start on their rows, the queue hands out the rest
            HIP_OK(c, hipMemcpyAsync(c->d_queue, &c->h_queue0, 4, hipMemcpyHostToDevice, c->stream));
        }
        rc = launch_persistent(c, SW, O, launch_cap, R, b_lo, b_hi, n_target, pause_stage, sdf_service, refill ? c->d_queue : nullptr, B);
        if (rc) return rc;
        int k = 0;
        if (tpw) {
            // ---- resident pass: ONE launch serves every closure round of this sub-batch from the ring; it ends when every
            //      problem has left the optimiser kernel (finished, paused at a stage boundary, or the round cap) ----
            ResidentArgs RA{};
            RA.coefH = R.coefH; RA.Amat = R.Amat; RA.tau = R.tau; RA.tag = R.tag;
            RA.done_round = R.done_round; RA.stats = R.stats; RA.wg_round = R.pass_done;
            RA.log = log_on ? c->d_vp_log : nullptr; RA.log_rounds = kVpLogRounds;
            RA.verts = c->d_verts;
            RA.capture_verts = c->capture_verts; RA.capture_round = c->capture_verts ? c->capture_round : -1;
            RA.nslots = kRingSlots; RA.rb = (int)rb;
            RA.b_lo = b_lo; RA.n = b_hi - b_lo;
            RA.flags = (unsigned)debug_hook("MVFIT_DEBUG_NT_OFF");         // (hooks build only) bit 1 = plain vertex stores
            RA.max_rounds = (unsigned)launch_cap;
            hipError_t e = launch_vertex_pass_resident(c->M, RA, tpw, c->pass_stream);
            if (e != hipSuccess) return fail(c, MVFIT_E_HIP, "resident vertex pass launch: %s", hipGetErrorString(e));
            HIP_OK(c, hipEventRecord(c->ev_batch[0], c->pass_stream));
        } else {
        // the passes: one per closure round, queued at most two batches ahead of the ones that have completed
        for (;; ++k) {
            for (int i = 0; i < kPassBatch && !dbg_nopass; ++i) {
                const unsigned r = (unsigned)(k * kPassBatch + i);
                const int slot = (int)(r % (unsigned)kRingSlots);
                DevPose P = c->P;                                          // side outputs / unused fields as in the chained mode
                // the pass addresses its operands by the global problem / chunk index: slot bases shifted by the sub-batch start
                P.coefH = R.coefH + ((ptrdiff_t)slot * (ptrdiff_t)rb - (ptrdiff_t)b_lo) * (KROWS / 4);
                P.coefT = nullptr;
                P.Amat = R.Amat + ((ptrdiff_t)slot * (ptrdiff_t)rb - (ptrdiff_t)b_lo) * 288;
                P.tau = R.tau + ((ptrdiff_t)slot * (ptrdiff_t)rb - (ptrdiff_t)b_lo) * 4;
                P.tag = R.tag + ((ptrdiff_t)slot * (ptrdiff_t)rb - (ptrdiff_t)b_lo);
                P.done_round = R.done_round;
                P.stats = R.stats;
                P.pass_done = R.pass_done;
                P.round = r;
                P.chunk0 = b_lo / 32;
                P.pad_ = (unsigned)debug_hook("MVFIT_DEBUG_NT_OFF");      // (hooks build only) bit 0 = plain basis loads, bit 1 = plain vertex stores
                float* vout = c->d_verts;
                if (c->capture_verts && (int)r == c->capture_round) vout = c->capture_verts;      // test hook
                if (sdf_service && pass_writes_box_parts(c, b_lo, b_hi)) P.box_part = c->d_sdf_boxpart;      // (the term's box from the pass's tile keys)
                hipError_t e = launch_pass_gate(P, b_lo, b_hi, c->pass_stream);
                hipEvent_t ea = nullptr, eb = nullptr;
                if (c->profile && c->ev_vp.size() < 4096) {            // mvfit_profile: the dispatch's own begin / end stamps
                    hipEventCreate(&ea); hipEventCreate(&eb);
                    c->ev_vp.emplace_back(ea, eb);
                }
                if (e == hipSuccess) e = launch_vertex_pass(c->M, P, b_hi, vout, c->opt.pass_kernel, c->pass_stream, ea, eb);
                if (e != hipSuccess) return fail(c, MVFIT_E_HIP, "vertex pass launch: %s", hipGetErrorString(e));
                if (sdf_service) {
                    // the term at the round's vertices for the problems whose gate word is set (written by the optimiser in
                    // front of the round's tag); transforms from the ring slot, float32 coefficients from the chained layout
                    DevPose Ps = P;
                    Ps.coefT = c->P.coefT;
                    e = launch_sdf_term(c->M, Ps, vout, b_hi, c->d_sdf_faces, c->sdf_num_faces, c->sdf_grid, c->F.sdf_gate, c->d_sdf_box,
                                        c->d_sdf_samp, c->d_sdf_entries, c->d_sdf_adj, c->pass_stream, c->d_sdf_cull, c->F.sdf_tag, r + 1u, P.box_part);
                    if (e != hipSuccess) return fail(c, MVFIT_E_HIP, "SDF term launch: %s", hipGetErrorString(e));
                }
            }
            HIP_OK(c, hipEventRecord(c->ev_batch[k & 3], c->pass_stream));
            if (k >= 2) HIP_OK(c, hipEventSynchronize(c->ev_batch[(k - 2) & 3]));
            if (*h_done >= n_target) break;
            if ((k + 1) * kPassBatch >= launch_cap) break;
        }
        }
        // behind the optimiser kernel (all problems of the sub-batch, or the round cap) the ctx stream continues behind the
        // last passes (nothing of the fit's result depends on them: ordering only)
        HIP_OK(c, hipStreamWaitEvent(c->stream, c->ev_batch[k & 3], 0));
    }
    // one host wait for all of it
    HIP_OK(c, hipMemcpyAsync(c->async_stats, R.stats, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(c, hipMemcpyAsync(c->h_done, c->F.n_done, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(c, hipMemcpyAsync(c->h_done + 1, c->F.n_done + 2, 4, hipMemcpyDeviceToHost, c->stream));   // problems that left, all sub-batches
    HIP_OK(c, hipStreamSynchronize(c->stream));
    *seen_out = c->h_done[0];
    if (tpw && c->opt.resident_pass < 0 && c->async_stats[3]) c->resident_auto_off = true;
    if (log_on) {
        // per round: service span = last workgroup's stores drained - first workgroup saw the operands; busy = a workgroup's own
        // drained - seen (wall clock, 100 MHz)
        std::vector<unsigned long long> lg((size_t)kVpLogRounds * res_grid * 2);
        HIP_OK(c, hipMemcpy(lg.data(), c->d_vp_log, lg.size() * 8, hipMemcpyDeviceToHost));
        double span = 0.0, busy = 0.0, slowest = 0.0;
        int n = 0;
        for (int r = 0; r < kVpLogRounds; ++r) {
            unsigned long long lo = ~0ull, hi = 0ull, bsum = 0ull, bmax = 0ull;
            bool all = true;
            for (int w = 0; w < res_grid; ++w) {
                const unsigned long long a = lg[((size_t)r * res_grid + w) * 2], z = lg[((size_t)r * res_grid + w) * 2 + 1];
                if (!z) { all = false; break; }
                lo = std::min(lo, a); hi = std::max(hi, z); bsum += z - a; bmax = std::max(bmax, z - a);
            }
            if (!all) break;
            span += (double)(hi - lo) * 1e-5; busy += (double)bsum / res_grid * 1e-5;      // ticks of 10 ns -> ms
            slowest += (double)bmax * 1e-5;
            ++n;
        }
        c->res_rounds = n;
        c->res_span_ms = n ? span / n : 0.0;
        c->res_busy_ms = n ? busy / n : 0.0;
        c->res_slowest_ms = n ? slowest / n : 0.0;
    }
    return MVFIT_OK;
}

extern "C" int mvfit_debug_capture_pass(mvfit_ctx* c, int round, float* verts) {
    if (!c) return MVFIT_E_ARG;
    c->capture_round = verts ? round : -1;
    c->capture_verts = verts;
    return MVFIT_OK;
}

extern "C" int mvfit_fit_stats(mvfit_ctx* c, uint32_t* out4) {
    if (!c || !out4) return MVFIT_E_ARG;
    for (int i = 0; i < 4; ++i) out4[i] = c->async_stats[i];
    return MVFIT_OK;
}

extern "C" int mvfit_decoder_stats(mvfit_ctx* c, uint32_t* out3) {
    if (!c || !out3) return MVFIT_E_ARG;
    unsigned st[2] = {0, 0};
    if (c->vps_mem && c->vps_stats[0]) {
        HIP_OK(c, hipSetDevice(c->device));
        HIP_OK(c, hipStreamSynchronize(c->stream));
        HIP_OK(c, hipMemcpy(st, c->vps_mem + c->vps_words, 8, hipMemcpyDeviceToHost));
    }
    out3[0] = c->vps_stats[0]; out3[1] = st[0]; out3[2] = st[1];
    return MVFIT_OK;
}

extern "C" int mvfit_fit(mvfit_ctx* c, const mvfit_weights* sw, const mvfit_lbfgs_opts* o, float* params,
                         float* final_loss, int32_t* n_closure, int32_t* n_iter) {
    if (!c || !sw || !o || !params) return MVFIT_E_ARG;
    if (c->B == 0) return fail(c, MVFIT_E_STATE, "call mvfit_set_problems first");
    HIP_OK(c, hipSetDevice(c->device));
    StageWeights SW;
    memset(&SW, 0, sizeof(SW));
    bool any_sdf = false;
    if (o->num_stages <= 0 || o->num_stages > MVFIT_MAX_STAGES) return fail(c, MVFIT_E_ARG, "num_stages");
    for (int s = 0; s < o->num_stages; ++s) {
        int rc = check_flags(c, sw[s].flags);
        if (rc) return rc;
        if (sw[s].flags != sw[0].flags) return fail(c, MVFIT_E_ARG, "flags must be identical for all stages");
        any_sdf = any_sdf || sw[s].coll_loss_weight > 0.f;
        SW.w[s] = to_dev(sw[s]);
    }
    if (any_sdf && !c->sdf_num_faces)
        return fail(c, MVFIT_E_STATE, "coll_loss_weight > 0 needs the SDF term's faces: call mvfit_set_sdf first");
    if (any_sdf) {
        int rc = ensure_sdf_buffers(c);
        if (rc) return rc;
    }
    c->F.sdf_adj = any_sdf ? c->d_sdf_adj : nullptr;
    c->F.trace = c->trace; c->F.trace_cap = c->trace ? c->trace_cap : 0;
    LbOpts O;
    int rc = make_opts(c, o, sw[0].flags, O);
    if (rc) return rc;
    // the interpenetration term reads every vertex: it forces the (vertex pass, step) round structure
    const bool sparse = (sw[0].flags & MVFIT_F_SPARSE_VERTS) != 0 && !any_sdf;
    // mvfit_options::round_mode = 1 keeps the chained (vertex pass -> step kernel) round graph also without the SDF term
    const bool serial = c->opt.round_mode == 1;
    const bool async = !sparse && !any_sdf && c->M.bs_h2 != nullptr && !serial;
    // With the SDF term: the leading stages whose coll_loss_weight is 0 (stages 1-2 of the yaml) do not need the vertices
    // before the loss - they run asynchronously like a fit without the term, every problem leaves at the stage boundary,
    // and the chained rounds take over from the stored optimiser / pose state (a fresh optimiser starts there anyway).
    int lead = 0;
    while (lead < o->num_stages && !(sw[lead].coll_loss_weight > 0.f)) ++lead;
    const bool two_phase = any_sdf && lead >= 1 && lead < o->num_stages && c->M.bs_h2 != nullptr && !serial &&
                           c->opt.sdf_two_phase != 0;
    // Round 6: the stages that carry the term run in the single-launch kernel too, with the term as a service (fit_async with
    // sdf_service; also when the FIRST stage carries it: no lead phase then).  mvfit_options::sdf_service = 0 keeps the chained
    // rounds - pass -> term -> step kernel per round -, which stay the checker of this path and the structure of profiled fits
    // and of MVFIT_F_REUSE_OUTER_VALUE fits; sdf_two_phase = 0 (chained rounds in every stage) switches it off as well
    const bool service = any_sdf && c->M.bs_h2 != nullptr && !serial && c->opt.sdf_two_phase != 0 && c->opt.sdf_service != 0 &&
                         !O.reuse_outer && !c->profile;
    for (unsigned& v : c->async_stats) v = 0;
    for (unsigned& v : c->vps_stats) v = 0;
    if (c->vps_mem) HIP_OK(c, hipMemsetAsync(c->vps_mem + c->vps_words, 0, 8, c->stream));
    const int B = c->B;
    HIP_OK(c, hipMemsetAsync(c->F.n_done, 0, 12, c->stream));
    HIP_OK(c, hipMemsetAsync(c->F.sdf_gate, sw[0].coll_loss_weight > 0.f ? 1 : 0, (size_t)B * 4, c->stream));
    hipLaunchKernelGGL(fit_init_kernel, dim3(B), dim3(STEP_NT), step_lds(), c->stream, c->M, (const ObsBlock*)c->d_obs, c->P, c->F,
                       (const float*)params,
                       sw[0].flags, (sparse || async || two_phase || service) ? 0 : 1);
    HIP_OK(c, hipGetLastError());
    int* h_done = c->h_done;
    *h_done = 0;
    int rounds = 0;
    const int cap = o->max_rounds > 0 ? o->max_rounds : (o->num_stages * o->maxiters * (O.max_eval + 30) + 8);
    if (two_phase) {
        int seen = 0;
        rc = fit_async(c, SW, O, cap, &seen, lead);
        if (rc) return rc;
        // every problem must have LEFT the single-launch kernel at the stage boundary (or finished): one that stopped at the
        // round cap mid-history would be continued by the chained step kernel, whose two-loop direction reads Gram rows the
        // single-launch kernel (compact direction form) does not maintain
        if (c->h_done[1] < B) {
            hipLaunchKernelGGL(fit_finish_kernel, dim3(B), dim3(128), 0, c->stream, c->F, params, final_loss, n_closure, n_iter, B,
                               o->num_stages);
            HIP_OK(c, hipGetLastError());
            return fail(c, MVFIT_E_STATE, "fit hit the round cap (%d) before all problems finished the stages without the SDF term", cap);
        }
    }
    if (service) {
        unsigned lead_stats[4];
        for (int i = 0; i < 4; ++i) lead_stats[i] = c->async_stats[i];
        int seen = 0;
        HIP_OK(c, hipMemsetAsync(c->F.n_done + 1, 0, 8, c->stream));
        rc = fit_async(c, SW, O, cap, &seen, MVFIT_MAX_STAGES + 1, true);
        if (rc) return rc;
        *h_done = seen;
        hipLaunchKernelGGL(fit_finish_kernel, dim3(B), dim3(128), 0, c->stream, c->F, params, final_loss, n_closure, n_iter, B,
                           o->num_stages);
        HIP_OK(c, hipGetLastError());
        // a gate that timed out lets the term's kernels run on another round's operands, a problem whose answer never came ends
        // with a NaN loss: neither is a result
        const unsigned sv_lost = c->async_stats[2], sv_gave_up = c->async_stats[3];
        for (int i = 0; i < 4; ++i) c->async_stats[i] += lead_stats[i];               // mvfit_fit_stats: the whole fit
        if (sv_lost || sv_gave_up)
            return fail(c, MVFIT_E_STATE, "SDF service rounds degraded (%u operand sets lost, %u waits given up): the fit is not valid - "
                        "is the GPU shared?  (mvfit_options::sdf_service = 0 runs these stages as chained rounds)", sv_lost, sv_gave_up);
        if (seen < B) return fail(c, MVFIT_E_STATE, "fit hit the round cap (%d) before all problems finished", cap);
        return MVFIT_OK;
    }
    if (async) {
        int seen = 0;
        rc = fit_async(c, SW, O, cap, &seen);
        if (rc) return rc;
        *h_done = seen;
    } else if (sparse) {
        // (with decoder helpers: sub-batches whose workgroups are all resident, one after the other)
        const int maxb = vps_enabled(c, SW) ? kVpsMaxSparse : B;
        const int nsub = (B + maxb - 1) / maxb, per = (B + nsub - 1) / nsub;
        for (int b_lo = 0; b_lo < B; b_lo += per) {
            const int b_hi = std::min(B, b_lo + per);
            const int done_before = *h_done;          // (synchronised: problems finished by the earlier sub-batches)
            rounds = 0;
            while (rounds < cap) {
                const int chunk = std::min(cap - rounds, 1 << 20);
                rc = launch_persistent(c, SW, O, chunk, AsyncRing{}, b_lo, b_hi, b_hi, MVFIT_MAX_STAGES + 1);
                if (rc) return rc;
                rounds += chunk;
                HIP_OK(c, hipMemcpyAsync(h_done, c->F.n_done, 4, hipMemcpyDeviceToHost, c->stream));
                HIP_OK(c, hipStreamSynchronize(c->stream));
                if (*h_done >= done_before + (b_hi - b_lo)) break;      // this sub-batch is complete (an earlier one may have hit the cap)
            }
        }
    } else if (c->profile) {
        // eager launches bracketed by events (bench.py's per-launch timing of the vertex pass)
        while (rounds < cap) {
            for (int r = 0; r < kGraphRounds; ++r) {
                rc = run_vertex_pass(c, c->d_verts);
                if (!rc && any_sdf) rc = run_sdf_term(c, c->d_verts, c->F.sdf_gate, c->stream);
                if (rc) return rc;
                prof_begin(c, c->ev_step);
                hipLaunchKernelGGL(O.reuse_outer ? fit_step_kernel<true> : fit_step_kernel<false>, dim3(B), dim3(STEP_NT), step_gram_lds(), c->stream, c->M, (const ObsBlock*)c->d_obs, c->V, SW, O,
                                   c->P, c->F);
                prof_end(c, c->ev_step);
            }
            HIP_OK(c, hipGetLastError());
            rounds += kGraphRounds;
            HIP_OK(c, hipMemcpyAsync(h_done, c->F.n_done, 4, hipMemcpyDeviceToHost, c->stream));
            HIP_OK(c, hipStreamSynchronize(c->stream));
            if (*h_done >= B) break;
        }
    } else {
        rc = ensure_round_graph(c, SW, O);
        if (rc) return rc;
        // While at most half of the problems have finished, the next replay is queued before the host looks at the
        // done counter of the current one (the GPU does not idle through the ~30 us host turnaround); later the
        // replays go one at a time, so that no replay runs after the last problem finished.
        if (!c->ev_done[0]) {
            HIP_OK(c, hipEventCreateWithFlags(&c->ev_done[0], hipEventDisableTiming));
            HIP_OK(c, hipEventCreateWithFlags(&c->ev_done[1], hipEventDisableTiming));
        }
        auto enqueue = [&](int slot) -> hipError_t {
            hipError_t e = hipGraphLaunch(c->round_graph, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&h_done[slot], c->F.n_done, 4, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipEventRecord(c->ev_done[slot], c->stream);
            return e;
        };
        int launched = 0, waited = 0, seen = 0;
        h_done[0] = h_done[1] = 0;
        const bool ahead_ok = B >= 8;
        while (true) {
            while (launched - waited < ((ahead_ok && seen <= B / 2) ? 2 : 1) && rounds < cap) {
                HIP_OK(c, enqueue(launched & 1));
                rounds += c->graph_rounds;
                ++launched;
            }
            if (launched == waited) break;                     // round cap reached
            HIP_OK(c, hipEventSynchronize(c->ev_done[waited & 1]));
            seen = h_done[waited & 1];
            ++waited;
            if (seen >= B) break;
        }
        HIP_OK(c, hipStreamSynchronize(c->stream));
        h_done[0] = seen;
    }
    const bool finished = *h_done >= B;
    hipLaunchKernelGGL(fit_finish_kernel, dim3(B), dim3(128), 0, c->stream, c->F, params, final_loss, n_closure, n_iter, B,
                       o->num_stages);
    HIP_OK(c, hipGetLastError());
    if (!finished) return fail(c, MVFIT_E_STATE, "fit hit the round cap (%d) before all problems finished", cap);
    return MVFIT_OK;
}

// The path's only collective: all-gather of the ranks' fitted parameters over RCCL, for hosts that own a
// communicator.  libmvfit does not link RCCL: the communicator belongs to the RCCL copy the host process loaded (PyTorch
// ships its own), so ncclAllGather is bound at run time to THAT library - the one already resident - never to a second one.
extern "C" int mvfit_gather(mvfit_ctx* c, void* rccl_comm, const void* send, void* recv, size_t bytes_per_rank) {
    if (!c) return MVFIT_E_ARG;
    if (!rccl_comm || !send || !recv) return fail(c, MVFIT_E_ARG, "mvfit_gather: null communicator or buffer");
    if (bytes_per_rank == 0) return MVFIT_OK;
    typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);     // ncclAllGather
    static allgather_fn fn = nullptr;
    if (!fn) {
        fn = reinterpret_cast<allgather_fn>(dlsym(RTLD_DEFAULT, "ncclAllGather"));
        for (const char* so : {"librccl.so", "librccl.so.1"}) {
            if (fn) break;
            if (void* h = dlopen(so, RTLD_NOW | RTLD_NOLOAD)) fn = reinterpret_cast<allgather_fn>(dlsym(h, "ncclAllGather"));
        }
    }
    if (!fn) return fail(c, MVFIT_E_STATE, "mvfit_gather: no RCCL library is loaded in this process (ncclAllGather not found)");
    HIP_OK(c, hipSetDevice(c->device));
    const int rc = fn(send, recv, bytes_per_rank, /* ncclInt8 */ 0, rccl_comm, c->stream);
    if (rc != 0) return fail(c, MVFIT_E_HIP, "mvfit_gather: ncclAllGather returned %d", rc);
    return MVFIT_OK;
}

extern "C" int mvfit_fit_trace(mvfit_ctx* c, float* trace, int max_closures) {
    if (!c || max_closures < 0 || (trace && max_closures == 0)) return MVFIT_E_ARG;
    c->trace = trace;
    c->trace_cap = trace ? max_closures : 0;
    return MVFIT_OK;
}

#ifdef MVFIT_LB_CHECK
// check build: [0] fast optimiser transitions cross-checked against the general state machine, [1] mismatches, [2] first word
extern "C" __attribute__((visibility("default"))) int mvfit_debug_lb_check(unsigned* out4, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out4, HIP_SYMBOL(mvfit::g_lb_check), sizeof(unsigned) * 4);
    if (reset) { unsigned z[4] = {0, 0, 0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(mvfit::g_lb_check), z, sizeof(z)); }
    return 0;
}
#endif
#ifdef MVFIT_TIMING
extern "C" __attribute__((visibility("default"))) int mvfit_debug_timing(long long* out32, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out32, HIP_SYMBOL(mvfit::g_dbg), sizeof(long long) * 32);
    if (reset) { long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(mvfit::g_dbg), z, sizeof(z)); }
    return 0;
}
extern "C" __attribute__((visibility("default"))) int mvfit_debug_timing_adv(long long* out16, int reset) {           // g_dbg[48..63]: inside lbfgs_advance
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(mvfit::g_dbg), sizeof(long long) * 16, sizeof(long long) * 48);
    if (reset) { long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(mvfit::g_dbg), z, sizeof(z), sizeof(long long) * 48); }
    return 0;
}
extern "C" __attribute__((visibility("default"))) int mvfit_debug_timing_calls(long long* out16, int reset) {          // g_dbg[64..79]: optimiser calls by kind (lbfgs_round)
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(mvfit::g_dbg), sizeof(long long) * 16, sizeof(long long) * 64);
    if (reset) { long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(mvfit::g_dbg), z, sizeof(z), sizeof(long long) * 64); }
    return 0;
}
extern "C" __attribute__((visibility("default"))) int mvfit_debug_timing_helpers(long long* out16, int reset) {       // g_dbg[32..47]: decoder helper (set 0, slice 0)
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(mvfit::g_dbg), sizeof(long long) * 16, sizeof(long long) * 32);
    if (reset) { long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(mvfit::g_dbg), z, sizeof(z), sizeof(long long) * 32); }
    return 0;
}
#endif

extern "C" int mvfit_sdf(mvfit_ctx* c, const int32_t* faces, int num_faces, const float* vertices, int B, int num_vertices,
                         int G, float* phi) {
    if (!c) return MVFIT_E_ARG;
    if (!faces || !vertices || !phi || num_faces < 0 || B <= 0 || num_vertices <= 0 || G < 2 || G > 1024)
        return fail(c, MVFIT_E_ARG, "mvfit_sdf: bad argument (num_faces=%d B=%d num_vertices=%d G=%d)", num_faces, B, num_vertices, G);
    HIP_OK(c, hipSetDevice(c->device));
    // long face lists: exact culling on face lists (sdf_term.hip), bit-identical to the walk; mvfit_options::sdf_face_lists = 0
    // keeps the walk
    c->sdf_op_path = 0;
    if (sdf_op_uses_lists(num_faces) && c->opt.sdf_face_lists) {
        if (c->sdf_op_B != B || c->sdf_op_F != num_faces) {       // a new shape: decide once (the decision, also a refusal, is kept)
            HIP_OK(c, hipStreamSynchronize(c->stream));
            if (c->d_sdf_op_ws) { hipFree(c->d_sdf_op_ws); c->d_sdf_op_ws = nullptr; }
            size_t free_b = 0, total_b = 0;
            HIP_OK(c, hipMemGetInfo(&free_b, &total_b));
            if (sdf_op_ws_bytes(B, num_faces) < free_b / 2) {
                HIP_OK(c, hipMalloc(&c->d_sdf_op_ws, sdf_op_ws_bytes(B, num_faces)));
                HIP_OK(c, hipMemsetAsync(reinterpret_cast<unsigned char*>(c->d_sdf_op_ws) + sdf_cull_zero_offset(B, num_faces), 0,
                                         sdf_cull_zero_bytes(B), c->stream));
            }
            c->sdf_op_B = B; c->sdf_op_F = num_faces;
        }
        if (c->d_sdf_op_ws) {
            hipError_t e = launch_sdf_voxelize_culled(faces, num_faces, vertices, B, num_vertices, G, phi, c->d_sdf_op_ws, c->stream);
            if (e != hipSuccess) return fail(c, MVFIT_E_HIP, "sdf launch: %s", hipGetErrorString(e));
            c->sdf_op_path = 1;
            return MVFIT_OK;
        }
        c->sdf_op_path = 2;                                       // the workspace did not fit: the walk
    }
    hipError_t e = launch_sdf_voxelize(faces, num_faces, vertices, B, num_vertices, G, phi, c->stream);
    if (e != hipSuccess) return fail(c, MVFIT_E_HIP, "sdf launch: %s", hipGetErrorString(e));
    return MVFIT_OK;
}

extern "C" int mvfit_triangulate(mvfit_ctx* c, int B, int V, const float* keypoints, const double* intris, const double* extris,
                                 double* joints3d) {
    if (!c) return MVFIT_E_ARG;
    if (B <= 0 || V <= 0 || !keypoints || !intris || !extris || !joints3d)
        return fail(c, MVFIT_E_ARG, "mvfit_triangulate: bad argument (B=%d V=%d)", B, V);
    HIP_OK(c, hipSetDevice(c->device));
    hipError_t e = launch_triangulate(keypoints, intris, extris, B, V, NKP, joints3d, c->stream);
    if (e != hipSuccess) return fail(c, MVFIT_E_HIP, "triangulate launch: %s", hipGetErrorString(e));
    return MVFIT_OK;
}

extern "C" int mvfit_depth_guess(mvfit_ctx* c, int B, const double* rest_joints, const double* extri, const double* intri,
                                 const float* keypoints, double* joints3d) {
    if (!c) return MVFIT_E_ARG;
    if (B <= 0 || !rest_joints || !extri || !intri || !keypoints || !joints3d)
        return fail(c, MVFIT_E_ARG, "mvfit_depth_guess: bad argument (B=%d)", B);
    HIP_OK(c, hipSetDevice(c->device));
    hipError_t e = launch_depth_guess(rest_joints, extri, intri, keypoints, B, NKP, joints3d, c->stream);
    if (e != hipSuccess) return fail(c, MVFIT_E_HIP, "depth guess launch: %s", hipGetErrorString(e));
    return MVFIT_OK;
}

extern "C" int mvfit_umeyama(mvfit_ctx* c, int B, int npts, const double* src, const double* dst, int estimate_scale,
                             double* rot, double* rvec, double* trans, double* scale) {
    if (!c) return MVFIT_E_ARG;
    if (B <= 0 || npts < 3 || !src || !dst || !rot || !rvec || !trans || !scale)
        return fail(c, MVFIT_E_ARG, "mvfit_umeyama: bad argument (B=%d npts=%d)", B, npts);
    HIP_OK(c, hipSetDevice(c->device));
    hipError_t e = launch_umeyama(src, dst, B, npts, estimate_scale, rot, rvec, trans, scale, c->stream);
    if (e != hipSuccess) return fail(c, MVFIT_E_HIP, "umeyama launch: %s", hipGetErrorString(e));
    return MVFIT_OK;
}

extern "C" int mvfit_project_points(mvfit_ctx* c, const float* points, int num_points, float* uv) {
    if (!c) return MVFIT_E_ARG;
    if (!points || !uv || num_points <= 0) return fail(c, MVFIT_E_ARG, "mvfit_project_points: bad argument (num_points=%d)", num_points);
    if (c->B == 0) return fail(c, MVFIT_E_STATE, "call mvfit_set_problems first (the cameras come from there)");
    HIP_OK(c, hipSetDevice(c->device));
    hipError_t e = launch_project_points(c->Q, points, num_points, uv, c->stream);
    if (e != hipSuccess) return fail(c, MVFIT_E_HIP, "projection launch: %s", hipGetErrorString(e));
    return MVFIT_OK;
}

extern "C" int mvfit_profile(mvfit_ctx* c, int enable) {
    if (!c) return MVFIT_E_ARG;
    c->profile = enable != 0;
    return MVFIT_OK;
}

// n back-to-back launches of the vertex pass on the pose operands the last closure / fit left behind,
// bracketed by ONE hipEvent pair on the ctx stream: elapsed / n is the per-launch duration with the event
// markers' own ~2-4 us amortised away (a pair around a single launch over-reports by about that much).
static int profile_vertex_pass(mvfit_ctx* c, int launches, int flavour, double* avg_ms) {
    if (!c || !avg_ms || launches <= 0) return MVFIT_E_ARG;
    if (c->B == 0) return fail(c, MVFIT_E_STATE, "call mvfit_set_problems first");
    HIP_OK(c, hipSetDevice(c->device));
    DevPose P = c->P;
    if (flavour == 1) {
        // the pass as the asynchronous fit launches it: operands from ring slot 0 (whatever trial points the last fit
        // left there), non-temporal basis stream / vertex stores, no side outputs; round 0 is live for every problem
        if (!c->ring.tag) return fail(c, MVFIT_E_STATE, "no asynchronous fit has run on this batch yet");
        P.coefH = c->ring.coefH; P.coefT = nullptr; P.Amat = c->ring.Amat; P.tau = c->ring.tau;
        P.tag = c->ring.tag; P.done_round = c->ring.done_round; P.stats = c->ring.stats; P.round = 0;
    }
    if (flavour == 2) {
        // the RESIDENT pass alone: `launches` (<= ring slots) closure rounds whose operands are already in the ring (whatever
        // trial points the last fit left in slots 0 .. launches - 1) and whose tags say so - ONE kernel launch inside one
        // hipEvent pair serves them back to back; elapsed / launches = the service time of a round with no optimiser next door
        if (!c->ring.tag || !c->resident_tpw) return fail(c, MVFIT_E_STATE, "no asynchronous fit with the resident pass has run on this batch yet");
        const AsyncRing& R = c->ring;
        const int n = std::min(c->B, R.Bpad), rounds = std::min(launches, R.nslots);
        std::vector<unsigned> tg((size_t)R.nslots * R.Bpad, 0u), dn((size_t)c->Bpad, 0u);
        for (int sl = 0; sl < rounds; ++sl) for (int q = 0; q < n; ++q) tg[(size_t)sl * R.Bpad + q] = (unsigned)sl + 1u;
        for (int q = 0; q < n; ++q) dn[q] = (unsigned)rounds;
        HIP_OK(c, hipStreamSynchronize(c->stream));
        HIP_OK(c, hipMemcpy(R.tag, tg.data(), tg.size() * 4, hipMemcpyHostToDevice));
        HIP_OK(c, hipMemcpy(R.done_round, dn.data(), dn.size() * 4, hipMemcpyHostToDevice));
        HIP_OK(c, hipMemset(R.pass_done, 0, 4 * kPassWords));
        ResidentArgs RA{};
        RA.coefH = R.coefH; RA.Amat = R.Amat; RA.tau = R.tau; RA.tag = R.tag; RA.done_round = R.done_round; RA.stats = R.stats;
        RA.wg_round = R.pass_done; RA.verts = c->d_verts; RA.capture_round = -1; RA.nslots = R.nslots; RA.rb = R.Bpad;
        RA.b_lo = 0; RA.n = n; RA.max_rounds = (unsigned)rounds + 1u;
        hipEvent_t a, b;
        HIP_OK(c, hipEventCreate(&a)); HIP_OK(c, hipEventCreate(&b));
        HIP_OK(c, hipEventRecord(a, c->stream));
        hipError_t e = launch_vertex_pass_resident(c->M, RA, c->resident_tpw, c->stream);
        HIP_OK(c, hipEventRecord(b, c->stream));
        HIP_OK(c, hipStreamSynchronize(c->stream));
        float ms = 0.f;
        hipError_t e2 = hipEventElapsedTime(&ms, a, b);
        hipEventDestroy(a); hipEventDestroy(b);
        if (e != hipSuccess || e2 != hipSuccess) return fail(c, MVFIT_E_HIP, "resident vertex pass timing failed");
        *avg_ms = (double)ms / rounds;
        return MVFIT_OK;
    }
    hipEvent_t a, b;
    HIP_OK(c, hipEventCreate(&a)); HIP_OK(c, hipEventCreate(&b));
    hipError_t e = launch_vertex_pass(c->M, P, c->B, c->d_verts, c->opt.pass_kernel, c->stream);      // warm
    HIP_OK(c, hipEventRecord(a, c->stream));
    for (int i = 0; i < launches && e == hipSuccess; ++i) e = launch_vertex_pass(c->M, P, c->B, c->d_verts, c->opt.pass_kernel, c->stream);
    HIP_OK(c, hipEventRecord(b, c->stream));
    HIP_OK(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    hipError_t e2 = hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    if (e != hipSuccess || e2 != hipSuccess) return fail(c, MVFIT_E_HIP, "vertex pass timing failed");
    *avg_ms = (double)ms / launches;
    return MVFIT_OK;
}

extern "C" int mvfit_profile_vertex_pass(mvfit_ctx* c, int launches, double* avg_ms) {
    return profile_vertex_pass(c, launches, 0, avg_ms);
}

extern "C" int mvfit_profile_vertex_pass_ex(mvfit_ctx* c, int launches, int flavour, double* avg_ms) {
    return profile_vertex_pass(c, launches, flavour, avg_ms);
}

static double drain(std::vector<std::pair<hipEvent_t, hipEvent_t>>& evs, int* n) {
    double tot = 0.0;
    int cnt = 0;
    for (auto& e : evs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { tot += ms; ++cnt; }
        hipEventDestroy(e.first); hipEventDestroy(e.second);
    }
    evs.clear();
    *n = cnt;
    return cnt ? tot / cnt : 0.0;
}

extern "C" int mvfit_pass_profile(mvfit_ctx* c, int* tiles_per_wg, int* workgroups, int* rounds, double* span_ms, double* busy_ms,
                                  double* slowest_ms) {
    if (!c) return MVFIT_E_ARG;
    if (tiles_per_wg) *tiles_per_wg = c->resident_tpw;
    if (workgroups) *workgroups = resident_grid(c, c->resident_tpw);
    if (rounds) *rounds = c->res_rounds;
    if (span_ms) *span_ms = c->res_span_ms;
    if (busy_ms) *busy_ms = c->res_busy_ms;
    if (slowest_ms) *slowest_ms = c->res_slowest_ms;
    return MVFIT_OK;
}

extern "C" int mvfit_profile_read(mvfit_ctx* c, double* vp_ms, int* launches, double* step_ms, int* step_launches) {
    if (!c) return MVFIT_E_ARG;
    HIP_OK(c, hipStreamSynchronize(c->stream));
    int n1 = 0, n2 = 0;
    double a = drain(c->ev_vp, &n1), b = drain(c->ev_step, &n2);
    // resident pass (one launch per fit): the per-round service span stamped inside the kernel stands for the launch duration
    if (n1 == 0 && c->res_rounds > 0) { a = c->res_span_ms; n1 = c->res_rounds; }
    if (vp_ms) *vp_ms = a;
    if (launches) *launches = n1;
    if (step_ms) *step_ms = b;
    if (step_launches) *step_launches = n2;
    return MVFIT_OK;
}

extern "C" int mvfit_lbfgs_kat(int device, int kind, int D, const int32_t* segs, int nseg, const mvfit_lbfgs_opts* o,
                               double* x_inout, double* trace, int max_trace, int* n_closure, double* final_loss) {
    if (!o || !x_inout || D <= 1 || D > LB_D || nseg < 1 || nseg > 8 || !segs) return MVFIT_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return MVFIT_E_HIP;
    LbOpts O;
    memset(&O, 0, sizeof(O));
    O.lr = o->lr; O.tol_grad = o->tolerance_grad; O.tol_change = o->tolerance_change; O.ftol = o->ftol; O.gtol = o->gtol;
    O.max_iter = o->max_iter; O.max_eval = o->max_iter * 5 / 4; O.history = o->history; O.maxiters = o->maxiters;
    O.num_stages = 1; O.nseg = nseg;
    for (int i = 0; i < nseg; ++i) { O.seg_lo[i] = segs[i]; O.seg_hi[i] = segs[i + 1]; }
    double *dx, *dtrace, *dfl, *ddirs, *dstps, *dro, *dgrow, *dgcol, *dcmat;
    int* dn;
    const size_t tb = (size_t)std::max(max_trace, 1) * (D + 1) * 8;
    if (hipMalloc(&dx, LB_D * 8) || hipMalloc(&dtrace, tb) || hipMalloc(&dfl, 8) || hipMalloc(&dn, 4) ||
        hipMalloc(&ddirs, LB_HIST * LB_D * 8) || hipMalloc(&dstps, LB_HIST * LB_D * 8) || hipMalloc(&dro, LB_HIST * 8) ||
        hipMalloc(&dgrow, LB_GSIZE * 8) || hipMalloc(&dgcol, LB_GSIZE * 8) || hipMalloc(&dcmat, 3 * LB_HIST * LB_HIST * 8))
        return MVFIT_E_HIP;
    hipMemcpy(dx, x_inout, D * 8, hipMemcpyHostToDevice);
    hipMemset(dtrace, 0, tb);
    hipMemset(dgrow, 0, LB_GSIZE * 8);
    hipMemset(dgcol, 0, LB_GSIZE * 8);
    hipMemset(ddirs, 0, LB_HIST * LB_D * 8);
    hipMemset(dstps, 0, LB_HIST * LB_D * 8);
    hipMemset(dcmat, 0, 3 * LB_HIST * LB_HIST * 8);
    hipLaunchKernelGGL(lbfgs_kat_kernel, dim3(1), dim3(64), 0, 0, kind, D, O, dx, dtrace, max_trace, dn, dfl, ddirs, dstps, dro,
                       dgrow, dgcol, dcmat);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(x_inout, dx, D * 8, hipMemcpyDeviceToHost);
    if (trace && max_trace > 0) hipMemcpy(trace, dtrace, tb, hipMemcpyDeviceToHost);
    if (n_closure) hipMemcpy(n_closure, dn, 4, hipMemcpyDeviceToHost);
    if (final_loss) hipMemcpy(final_loss, dfl, 8, hipMemcpyDeviceToHost);
    hipFree(dx); hipFree(dtrace); hipFree(dfl); hipFree(dn); hipFree(ddirs); hipFree(dstps); hipFree(dro); hipFree(dgrow); hipFree(dgcol); hipFree(dcmat);
    return e == hipSuccess ? MVFIT_OK : MVFIT_E_HIP;
}

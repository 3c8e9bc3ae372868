// Shared device-side structures of libmvfit (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mvfit.h"
#include "wave_ops.h"
#include "lbfgs_device.h"
#include "vposer_service.h"

namespace mvfit {

constexpr int NJ = 24;            // SMPL joints
constexpr int NKP = 17;           // dataset keypoints
constexpr int KROWS = 224;        // blendshape rows: 207 pose + 10 shape, padded to 28*8
constexpr int KGROUPS = 28;       // groups of 4 MFMA k-steps (8 rows) each
constexpr int TILE_V = 32;        // vertices per MFMA tile
constexpr int DV = MVFIT_D;       // 118
constexpr int DPAD = 128;
constexpr int NS_MAX = 96;        // selected (objective-relevant) vertices
constexpr int NS_STRIDE = 112;    // row stride of the transposed skinning weights (== 16 mod 32: the four
                                  // 16-lane rows of a wave hit disjoint LDS banks)
constexpr int NC_MAX = NS_MAX * 3;
constexpr int KNNZ_MAX = 160;     // non-zeros of the 17 x ns keypoint selection
constexpr int KP_NZ = 12;         // padded per-keypoint list length (LSP regressor rows have 4-9 non-zeros)
constexpr int VS_NZ = 2;          // padded per-vertex list length (a vertex usually feeds one keypoint)
constexpr int A_STRIDE = 288;     // per-problem stride of the 24x12 skinning transforms in LDS (vertex pass): the HBM stride, so that a chunk's
                                  // transforms are one linear global -> LDS copy
constexpr int STEP_NT = 512;      // threads of the per-problem kernels (8 waves)
constexpr int STEP_NW = STEP_NT / 64;

// flat parameter layout (include/mvfit.h)
constexpr int X_BETAS = 0, X_GO = 10, X_BP = 13, X_TR = 82, X_SC = 85, X_EMB = 86;

// Model constants every per-problem workgroup keeps in LDS (bulk-copied once per launch).
struct ModelLds {
    float wT[NJ][NS_STRIDE];          // lbs_weights of the selected vertices, transposed: wT[j][s]
    float J_t[NJ * 3];                // J_regressor . v_template
    float J_S[NJ * 3][11];            // J_regressor . shapedirs  (row padded to 11: conflict-free by lane)
    float vt_sub[NC_MAX];             // v_template of the selected vertices, c = 3 s + a
    int sel_v[NS_MAX];                // vertex id of selected vertex s
    int kp_start[NKP + 1];            // keypoint k = sum_t kp_w[t] * xs[kp_s[t]]  (ascending s)
    int kp_s[KNNZ_MAX];
    float kp_w[KNNZ_MAX];
    int vs_start[NS_MAX + 1];         // transpose: selected vertex s feeds keypoints vs_k[t] (ascending k)
    int vs_k[KNNZ_MAX];
    float vs_w[KNNZ_MAX];
    // the same selection as fixed-length zero-padded lists (all index loads of a thread in one LDS round
    // trip instead of one per CSR entry); padded = 0 when a row is longer than the padding (CSR is used)
    int kpp_s[NKP][KP_NZ];
    float kpp_w[NKP][KP_NZ];
    int vsp_k[NS_MAX][VS_NZ];
    float vsp_w[NS_MAX][VS_NZ];
    int padded, padx0, padx1, padx2;
    int parents[NJ];
    int nlevels;
    int level_start[NJ + 1];
    int level_joints[NJ];
    int child_start[NJ + 1];
    int child_list[NJ];
    // kinematic chain schedules for ONE wave (12 lanes per joint, 5 joints per pass):
    //   fwd_tab[pass][q] = j | parent << 8 (or -1): joints whose parent transform is complete
    //   bwd_tab[pass][q] = parent | 0x80 if not its first entry | c0 << 8 | c1 << 16 | c2 << 24 (or -1), child 31 = none
    int n_fwd, n_bwd;
    int fwd_tab[NJ][5];
    int bwd_tab[NJ][5];
    // pointer-jumping form of the forward chain: anc_tab[s][j] = the 2^s-th ancestor of joint j (-1: above the root);
    // n_jump = steps until every path product is complete (2^n_jump >= joints on the longest path)
    int anc_tab[5][NJ];
    int n_jump, jpad0, jpad1, jpad2;
    int ns, nc, nc_pad, pad0;
    // the selected vertices' skinning weights as <= 4 (weight, joint) pairs in ascending joint order, zero-padded (the
    // non-zero products of the dense row in the same order: the same bits); sel_sparse = 0 when a row has more than 4
    float selw[NS_MAX][4];
    unsigned selj[NS_MAX];            // four joint indices, one per byte
    int sel_sparse, spad0, spad1, spad2;
};
static_assert(sizeof(ModelLds) % 16 == 0, "ModelLds is bulk-copied as 16-byte words");

struct DevModel {
    int nv, nv_pad, ntiles;
    const float* bs4;        // [ntiles][3][KGROUPS][64][4]   MFMA-B-operand order (see vertex pass)
    // the same basis split into fp16 pairs for the fp16 matrix pipe (null: exact-fp32 contraction):
    // [ntiles][3][14 blocks][hi, lo][64 lanes] 16-byte words = 8 fp16 of rows 16 G + 8 (lane >> 5) + t, vertex
    // 32 T + (lane & 31), values scaled by bs_scale (a power of two)
    const float4* bs_h2;
    float bs_scale;
    int half_basis;          // 1: the contraction reads only the hi halves of the split basis (2 bytes per element, configs[4])
    const float* vt_planes;  // [3][nv_pad]
    const float* wt_tiles;   // [ntiles][24][32]
    // sparse skinning (null unless every vertex has <= 4 non-zero weights): per padded vertex 4 weights and
    // their joint indices in ascending order, zero-weight padding
    const float4* wsp_w;     // [nv_pad]
    const int4* wsp_j;       // [nv_pad]
    // vertex-major copies for the per-vertex pull-back of the SDF term (sdf_term.hip)
    const float* bs_vm;      // [nv][3][KROWS]   same row order as the coefficient vector
    const float* w_vm;       // [nv][24]
    // objective-relevant vertex subset
    int ns, nc, nc_pad;      // nc = 3*ns, nc_pad multiple of 4
    const ModelLds* mlds;    // LDS image (global copy)
    const int* sel_v;        // [ns]
    const float* pd_sub;     // [KROWS][nc_pad]
    const float* pd_subT;    // [nc_pad][KROWS]
    // vertex-pass side outputs for the selected vertices: per tile, which local vertices are selected
    const int* tile_sel_start;   // [ntiles + 1]
    const int* tile_sel_local;   // [ns] local vertex index inside its tile
    const int* tile_sel_slot;    // [ns] selected-vertex slot s
    // VPoser decoder (null if absent)
    const float* vp_w1; const float* vp_b1;     // [512][32]
    const float* vp_w2; const float* vp_b2;     // [512][512]
    const float* vp_w3; const float* vp_b3;     // [138][512]
    const float* vp_w1T;                        // [32][512]
    const float* vp_w2T;                        // [512][512]
    const float* vp_w3T;                        // [512][138->144]
    VpTiles vpt;                                // the same weights as register tiles of the decoder helpers (vposer_service.h)
    VpService vps;                              // per launch: the helpers of THIS launch (nsets == 0: none)
    // GMM
    int gmm_M;
    const float* gmm_means;      // [M][69]
    const float* gmm_prec;       // [M][69][72]   rows padded to 72 floats (16-byte aligned)
    const float* gmm_precT;      // [M][69][72]   transposed precisions
    const float* gmm_lognw;      // [M]  log(nll_weights)
};

struct DevProblems {
    int B, V, cam_batched;
    const float* cam_R;   // [.,V,9]
    const float* cam_t;   // [.,V,3]
    const float* cam_f;   // [.,V]
    const float* cam_c;   // [.,V,2]
    const float* gt_xy;   // [B,V,17,2]
    const float* w_conf;  // [B,V,17]
};

struct DevWeights {          // one stage
    float data_w2;           // data_weight^2
    float pose_w;            // body_pose_weight
    float shape_w;
    float bend_w;
    float coll_w;
    float rho2;
    uint32_t flags;
    uint32_t pad;
};

// per-problem pose workspace handed from the step kernel to the vertex pass
//   coefT [B/32][KROWS][32]   (pose_feature | betas | 0), transposed per 32-problem chunk
//   coefH the same coefficients, split into fp16 (hi, lo) pairs in MFMA A-operand order
//   Amat  [B][24][12]
//   tau   [B][4]
// and back (vertex pass -> step kernel), for the selected vertices only:
//   vposed_sel [B][NC_MAX]   blendshaped rest positions  (lbs.py:203), c = 3 s + a
//   xs_sel     [B][NC_MAX]   skinned positions before "+ transl"
struct DevPose {
    float4* coefH;           // [B/32][14 blocks][hi, lo][64 lanes] 16-byte words: the coefficients of a chunk as
                             // split-fp16 MFMA A operands (problem = lane & 31, rows 16 G + 8 (lane >> 5) + t)
    float* coefT;
    float* Amat;
    float* tau;
    float* vposed_sel;
    float* xs_sel;
    // asynchronous fit (mvfit_fit without the SDF term): the operands are a ring slot that the optimiser kernel
    // publishes while this pass is already queued; tag != nullptr makes the pass wait for them (vertex_pass.hip)
    const unsigned* tag;         // [Bpad] of this slot: round + 1 of the trial point each problem's operands belong to
    const unsigned* done_round;  // [Bpad] closures evaluated by a finished problem (0xffffffff while it is running)
    unsigned* stats;             // [4] passes run / chunk passes skipped (all problems finished) / missed / timed out
    unsigned round;              // closure round this pass belongs to
    unsigned pad_;
    // sub-batch of the asynchronous fit (mvfit_fit time-slices batches with more problems than the optimiser gets CUs):
    // this launch covers the 32-problem chunks from chunk0 on, up to the problem count it is given
    int chunk0;
    unsigned* pass_done;         // [1] rounds whose pass has completed (written by the gate of the next round)
    unsigned long long* box_part; // [B][ntiles][6] min / max keys of the tile's vertices per axis (single-chunk split kernel; the SDF
                                 // term's bounding box is reduced from them), or null
};

// Pose-operand ring of the asynchronous fit: slot (r % nslots) holds, per problem, the vertex-pass operands of the
// trial point of that problem's closure round r.  Producer: fit_persistent_kernel (write-through sc1 stores, then
// the problem's tag); consumer: the vertex pass launched for round r (polls the tags of its 32 problems).
struct AsyncRing {
    float4* coefH;               // [nslots][Bpad / 32][KROWS / 16][hi, lo][64]
    float* Amat;                 // [nslots][Bpad][288]
    float* tau;                  // [nslots][Bpad][4]
    unsigned* tag;               // [nslots][Bpad]
    unsigned* done_round;        // [Bpad]
    unsigned* stats;             // [4]
    int* host_done;              // pinned host word: set to the number of problems by the last one to finish
    int nslots, Bpad;
    // back-pressure: rounds whose vertex pass has completed.  A problem publishes round r into slot r % nslots only once
    // the pass of round r - nslots has run (pass_done > r - nslots): a slot is never overwritten before its pass read it,
    // however slow the passes are (the optimiser is throttled to their rate instead).
    // npass words, the minimum counts: ONE written by the gate kernels of the per-round launches, or one per workgroup of
    // the resident pass (rounds whose operands that workgroup has read) - no atomics, every writer owns its word.
    unsigned* pass_done;
    int npass, rpad_;
};

// Resident vertex pass of the asynchronous fit (lbs_vertex_pass_resident_kernel, vertex_pass.hip): ONE launch per
// (sub-batch) fit; every workgroup keeps the basis of its vertex tile(s) in registers and serves closure round after
// closure round from the ring.
struct ResidentArgs {
    const float4* coefH;         // ring bases: slot 0, problem 0 of the sub-batch
    const float* Amat;
    const float* tau;
    const unsigned* tag;
    const unsigned* done_round;  // [Bpad] by global problem index
    unsigned* stats;             // [4] chunk passes run / skipped / missed / timed out
    unsigned* wg_round;          // [grid] back-pressure words (AsyncRing::pass_done)
    unsigned long long* log;     // mvfit_profile: [log_rounds][grid][2] wall-clock stamps {operands seen, stores drained}, or null
    float* verts;                // [B][nv][3] by global problem index
    float* capture_verts;        // test hook: the round capture_round is written here instead
    int capture_round;
    int nslots, rb;              // ring geometry: slots, problems per slot
    int b_lo, n;                 // the sub-batch: global index of its first problem, number of problems
    int log_rounds;
    unsigned flags;              // bit 1: plain (not non-temporal) vertex stores
    unsigned max_rounds;
};

// SDF interpenetration term (fitting.py:352-393), per problem
struct SdfBox {              // bounding box of the vertices: centre, scale, arg indices (fitting.py:282-288,356-359)
    float c[3];
    float s;
    int imin[3];
    int imax[3];
    int amax;                // axis of the largest extent
    int pad;
};
struct SdfAdj {              // S = sum_v phi_v and its adjoint w.r.t. the vertex pass operands
    float S;
    float gtau[3];
    float gA[NJ * 12];       // per joint [g_Ar (a-major 3x3) | g_At (3)]: the accumulator order of the closure's E6
    float gcoef[KROWS];
};
static_assert(sizeof(SdfAdj) % 16 == 0, "16-byte block");

}  // namespace mvfit

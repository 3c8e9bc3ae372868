// Shared device-side structures of libmvfit (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mvfit.h"
#include "lbfgs_device.h"

namespace mvfit {

constexpr int NJ = 24;            // SMPL joints
constexpr int NKP = 17;           // dataset keypoints
constexpr int KROWS = 224;        // blendshape rows: 207 pose + 10 shape, padded to 28*8
constexpr int KGROUPS = 28;       // groups of 4 MFMA k-steps (8 rows) each
constexpr int TILE_V = 32;        // vertices per MFMA tile
constexpr int DV = MVFIT_D;       // 118
constexpr int DPAD = 128;
constexpr int NS_MAX = 128;       // selected (objective-relevant) vertices
constexpr int NC_MAX = NS_MAX * 3;
constexpr int A_STRIDE = 292;     // per-problem stride of the 24x12 skinning transforms in LDS

// flat parameter layout (include/mvfit.h)
constexpr int X_BETAS = 0, X_GO = 10, X_BP = 13, X_TR = 82, X_SC = 85, X_EMB = 86;

struct DevModel {
    int nv, nv_pad, ntiles;
    const float* bs4;        // [ntiles][3][KGROUPS][64][4]   MFMA-B-operand order (see vertex pass)
    const float* vt_planes;  // [3][nv_pad]
    const float* wt_tiles;   // [ntiles][24][32]
    const float* J_t;        // [24][3]
    const float* J_S;        // [24][3][10]
    // objective-relevant vertex subset
    int ns, nc, nc_pad;      // nc = 3*ns, nc_pad multiple of 4
    const int* sel_v;        // [ns]
    const float* vt_sub;     // [nc_pad]
    const float* pd_sub;     // [KROWS][nc_pad]
    const float* pd_subT;    // [nc_pad][KROWS]
    const float* w_sub;      // [ns][24]
    const float* ksel_sub;   // [17][NS_MAX]
    // kinematic tree
    int parents[NJ];
    int nlevels;
    int level_start[NJ + 1];
    int level_joints[NJ];
    int child_start[NJ + 1];
    int child_list[NJ];
    // VPoser decoder (null if absent)
    const float* vp_w1; const float* vp_b1;     // [512][32]
    const float* vp_w2; const float* vp_b2;     // [512][512]
    const float* vp_w3; const float* vp_b3;     // [138][512]
    const float* vp_w1T;                        // [32][512]
    const float* vp_w2T;                        // [512][512]
    const float* vp_w3T;                        // [512][138->144]
    // GMM
    int gmm_M;
    const float* gmm_means;      // [M][69]
    const float* gmm_prec;       // [M][69][69]
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
//   Amat  [B][24][12]
//   tau   [B][4]
struct DevPose {
    float* coefT;
    float* Amat;
    float* tau;
};

}  // namespace mvfit

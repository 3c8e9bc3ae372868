/*
 * mvfit.h - C ABI of the MI355X-native multi-view SMPL fitting hot path.
 *
 * The reference (boycehbz/MvSMPLfitting) has no FFI on this path except the SDF op; its
 * seams are Python callables.  Each entry point below names the reference interface it
 * replaces (file:line relative to the reference root).  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative MVFIT_E_* code on failure, never throws;
 *     mvfit_last_error(ctx) returns a NUL-terminated message owned by the ctx.
 *   - a ctx is bound to one HIP device and one hipStream_t, is not thread-safe; different
 *     ctxs are independent.  Calls enqueue their work on the ctx stream and return; the ones that
 *     block until the device has finished are mvfit_create, mvfit_set_problems, mvfit_set_sdf,
 *     mvfit_sync, mvfit_destroy, the profiling readers and mvfit_fit (it watches the problems'
 *     completion to stop queueing rounds; its outputs are complete when it returns).
 *   - "dev|host" pointers may be either (copied with hipMemcpyDefault); "dev" pointers must be
 *     device memory (e.g. a torch CUDA tensor's data_ptr()); all arrays row-major float32
 *     unless noted.  Caller-owned; nothing is retained beyond the call except by
 *     mvfit_create, which copies (and re-tiles) the model constants into HBM.
 *   - flat parameter vector x[D], D = MVFIT_D = 86 + 32:  the reference's final_params order
 *     (code/utils/non_linear_solver.py:164-170, code/smplx/body_models_scale.py:202-268)
 *       betas[0:10] global_orient[10:13] body_pose[13:82] transl[82:85] scale[85]
 *       pose_embedding[86:118]
 *     With MVFIT_F_VPOSER the body_pose slots are ignored on input (decoded from the
 *     embedding) and receive zero gradient; without it the embedding slots are ignored.
 *     MVFIT_F_FIX_SHAPE / MVFIT_F_FIX_SCALE freeze betas / scale (gradient forced to 0:
 *     code/utils/init_guess.py:205-210).
 */
#ifndef MVFIT_H_
#define MVFIT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libmvfit.so is built with -fvisibility=hidden: the functions declared in this header are its only exports */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define MVFIT_NUM_JOINTS 24
#define MVFIT_NUM_BETAS 10
#define MVFIT_NUM_POSE_BASIS 207
#define MVFIT_NUM_KP 17
#define MVFIT_D 118          /* 86 model scalars + 32 latent */
#define MVFIT_D_MODEL 86
#define MVFIT_MAX_VIEWS 16
#define MVFIT_MAX_STAGES 8
#define MVFIT_HISTORY 100

/* error codes */
#define MVFIT_OK 0
#define MVFIT_E_ARG (-1)
#define MVFIT_E_HIP (-2)
#define MVFIT_E_STATE (-3)
#define MVFIT_E_UNSUPPORTED (-4)

/* flags (mvfit_weights.flags) */
#define MVFIT_F_VPOSER 1u      /* use_vposer: code/utils/fitting.py:166-168,327-329 */
#define MVFIT_F_PRIOR_GMM 2u   /* body_prior_type 'gmm' (code/prior.py:100-231) instead of 'l2' */
#define MVFIT_F_FIX_SHAPE 4u   /* fix_shape: code/utils/fitting.py:340, init_guess.py:208-210 */
#define MVFIT_F_FIX_SCALE 8u   /* fix_scale: init_guess.py:205-207 */
#define MVFIT_F_USE_3D 32u      /* use_3d: 3-D joint term (code/utils/fitting.py:319-324); needs mvfit_set_joints3d */
#define MVFIT_F_SPARSE_VERTS 16u /* evaluate only the vertices the objective reads (same loss /
                                    gradient; skips the full 6890-vertex pass inside the closure) */
#define MVFIT_F_REUSE_OUTER_VALUE 64u /* mvfit_fit only, opt-in, NOT the reference's closure count: LBFGS.step() opens with a
                                    closure call (lbfgs_ls.py:279-283) at the point the previous step() of the stage ended on;
                                    with this flag the optimiser feeds the loss / gradient it still holds instead of evaluating
                                    again - same iterates and eval accounting, 8-10 % fewer closure evaluations */

typedef struct mvfit_ctx mvfit_ctx;

/* Host-side description of the body model; replaces the buffers registered by
 * SMPL.__init__ (code/smplx/body_models_scale.py:197-305). */
typedef struct mvfit_model {
    int32_t num_verts;               /* 6890 */
    int32_t num_faces;               /* 13776 (0 if faces == NULL) */
    const float* v_template;         /* [Nv,3] */
    const float* shapedirs;          /* [Nv,3,10]   (beta index fastest) */
    const float* posedirs;           /* [207, Nv*3] (column = 3*vertex + coord) */
    const float* J_regressor;        /* [24,Nv] dense */
    const int32_t* parents;          /* [24], parents[0] = -1 */
    const float* lbs_weights;        /* [Nv,24] dense */
    const float* kp_regressor;       /* [14,Nv] dense ('smpllsp' joint_regressor, :283-286) */
    const int32_t* face_vertex_ids;  /* [5]  (code/smplx/vertex_joint_selector.py:38-43) */
    const int32_t* joint_map;        /* [17] (code/utils/utils.py:453-457) */
    const int32_t* faces;            /* [Nf,3] or NULL */
    /* optional VPoser decoder (code/model/VPoser.py:188-195), NULL if unused */
    const float* vp_fc1_w; const float* vp_fc1_b;   /* [512,32],[512] */
    const float* vp_fc2_w; const float* vp_fc2_b;   /* [512,512],[512] */
    const float* vp_out_w; const float* vp_out_b;   /* [138,512],[138] */
    /* optional max-mixture prior (code/prior.py:135-160), gmm_M = 0 if unused */
    int32_t gmm_M;
    const float* gmm_means;          /* [M,69] */
    const float* gmm_precisions;     /* [M,69,69] */
    const float* gmm_nll_weights;    /* [M] */
} mvfit_model;

/* One stage's loss weights; replaces SMPLifyLoss.reset_loss_weights
 * (code/utils/fitting.py:270-280) + the per-stage dict of non_linear_solver.py:109-124,177-180. */
typedef struct mvfit_weights {
    float data_weight;           /* 500/H, enters squared (fitting.py:315) */
    float body_pose_weight;      /* enters squared (fitting.py:329,333,337) */
    float shape_weight;          /* enters squared (fitting.py:342) */
    float bending_prior_weight;  /* 3.17*body_pose_weight, NOT squared (fitting.py:348) */
    float coll_loss_weight;      /* SDF term (fitting.py:354,392); 0 = off */
    float rho;                   /* GMoF rho (code/utils/utils.py:427-438) */
    uint32_t flags;              /* MVFIT_F_* */
} mvfit_weights;

/* Optimiser settings; replaces create_optimizer(..., 'lbfgsls') (code/optimizers/optim_factory.py:50-52,
 * lbfgs_ls.py:199-207) and FittingMonitor(maxiters, ftol, gtol) (code/utils/fitting.py:38-47). */
typedef struct mvfit_lbfgs_opts {
    float lr;                /* 1.0 */
    int32_t max_iter;        /* 30 (max_eval = max_iter*5/4) */
    int32_t history;         /* <= MVFIT_HISTORY (100) */
    float tolerance_grad;    /* 1e-5 */
    float tolerance_change;  /* 1e-9 */
    int32_t maxiters;        /* outer run_fitting iterations, 30 */
    float ftol;              /* 1e-9 */
    float gtol;              /* 1e-9 */
    int32_t num_stages;      /* <= MVFIT_MAX_STAGES */
    int32_t max_rounds;      /* safety cap on closure rounds per call (0 = no cap) */
} mvfit_lbfgs_opts;

/* SMPL.__init__ + .to(device) (body_models_scale.py:98-305, code/init.py:143-151): copies and
 * re-tiles the constants into HBM.  hip_stream may be NULL (default stream). */
/* Precision and path selectors of a ctx.  The released library reads NO environment variable: what used to be MVFIT_*
 * switches are fields here (mvfit_options_default fills the defaults; mvfit_create(...) = mvfit_create_ex(..., NULL) = the
 * defaults).  The first group decides what mvfit_create_ex uploads and is fixed for the ctx's life; the second group may be
 * changed between calls with mvfit_set_options. */
#define MVFIT_CONTRACTION_SPLIT_FP16 0   /* default: every fp32 product of the blendshape contraction as error-compensated
                                          * split-fp16 pairs on the fp16 matrix pipe, fp32 accumulate (5e-7 from float64) */
#define MVFIT_CONTRACTION_EXACT_FP32 1   /* the contraction as an exact fp32 MFMA chain (bitwise an fmaf chain) */
#define MVFIT_CONTRACTION_HALF_BASIS 2   /* BASELINE configs[4], half-width blendshape operands: only the fp16 hi halves of the
                                          * basis are streamed (2 bytes per element; vertices within ~2e-5 of the fp32 result) */
typedef struct mvfit_options {
    uint32_t struct_size;            /* sizeof(mvfit_options) of the caller */
    /* ---- fixed at mvfit_create_ex ---- */
    int32_t contraction;             /* MVFIT_CONTRACTION_* */
    int32_t dense_skinning;          /* 1: the dense 24-column skinning blend even when every vertex has <= 4 weights (0) */
    /* ---- mvfit_set_options ---- */
    int32_t round_mode;              /* 0: automatic (asynchronous single-launch fit wherever the objective allows);
                                      * 1: chained rounds always (vertex pass -> step kernel per closure round) */
    int32_t resident_pass;           /* vertex passes of the asynchronous fit: -1 automatic (resident when its workgroups fit
                                      * next to the optimiser's), 0 a gate + a pass launch per closure round, 1 / 2 resident with
                                      * that many vertex tiles per workgroup, 3 resident with two tiles per workgroup and the
                                      * workgroup split into contraction and worker waves (a forced value that does not fit
                                      * stalls the fit; round 6: forms 1 and 3 are the automatic choices - 216 workgroups beside
                                      * <= 36 optimiser workgroups, 108 beside <= 144 -, form 2 was dropped: it maps to 3).
                                      * The resident pass assumes what the path's deployment gives it - one process per GPU
                                      * (SURVEY 8(e)): a fit's ~250 workgroups are resident together.  Processes (or concurrent
                                      * ctxs) that SHARE a device should set 0: waiting for one another's CUs they would exhaust
                                      * the ring's patience (20 ms) and lose passes - counted by mvfit_fit_stats, never silent,
                                      * and without effect on the fitted parameters */
    int32_t sdf_two_phase;           /* 1 (default): a fit with the SDF term runs its leading coll_loss_weight == 0 stages
                                      * asynchronously and hands over to chained rounds; 0: chained rounds in every stage */
    int32_t sdf_face_lists;          /* 1 (default): long face lists are culled exactly on per-round face lists (bit-identical
                                      * to the walk); 0: the walk over every face for every vertex / voxel */
    int32_t vposer_helpers;          /* 1 (default): the single-launch fits decode VPoser on helper workgroups; 0: in the
                                      * problems' own workgroups (another summation order: last-bit differences) */
    int32_t vposer_sets;             /* 0: automatic (16 sets for <= 32 problems, else 8; asynchronous fits take 8 where 16 would
                                      * leave no CUs for the resident vertex pass); n: helper sets of a launch (clamped to what
                                      * the problems need / fit).  Results do not depend on it */
    int32_t closure_vposer_helpers;  /* 1: mvfit_closure (MVFIT_F_VPOSER, no SDF term, B <= 160) decodes on helper workgroups
                                      * of its own launch - the decoder arithmetic of the fits, for parity tests (0) */
    int32_t pass_kernel;             /* per-round launch kernels at more than 32 problems: 0 automatic (two-role pipeline; dense
                                      * skinning rows: the lock-step chunk loop), 1 one workgroup per (tile, chunk); 2 = 0 (the
                                      * lock-step loop for <= 4 weights per vertex was dropped in round 6) */
    int32_t sdf_service;             /* 1 (default): in a two-phase fit (sdf_two_phase) the stages WITH the SDF term also run in the
                                      * single-launch optimiser kernel, which asks for the term every closure round - gate ->
                                      * vertex pass -> term kernels per round on the pass stream, the pull-back answers through
                                      * memory (fitting.py:352-393 unchanged: the same kernels compute the same S and adjoint);
                                      * 0: those stages as chained rounds (pass -> term -> step kernel launch per round) */
    int32_t work_queue;              /* 1 (default): an asynchronous fit of more problems than optimiser workgroups (128 beside the
                                      * resident vertex pass) is ONE launch whose workgroups take the next unfitted problem when
                                      * theirs has finished; 0: sub-batches one after the other.  A problem's result does not depend
                                      * on it (problems are independent) */
} mvfit_options;
void mvfit_options_default(mvfit_options* opts);

/* Error contract: MVFIT_E_ARG for a null / incomplete model leaves *out = NULL.  Any later failure (unsupported
 * model, device allocation) still stores a ctx in *out: it carries the message (mvfit_last_error) and owns whatever
 * was allocated so far - release it with mvfit_destroy, it is not usable for anything else. */
int mvfit_create(mvfit_ctx** out, int device, void* hip_stream, const mvfit_model* model);
/* mvfit_create with explicit options (NULL = defaults).  MVFIT_E_ARG for an unknown selector value. */
int mvfit_create_ex(mvfit_ctx** out, int device, void* hip_stream, const mvfit_model* model, const mvfit_options* opts);
/* Change the second group of options between calls; the first group must equal what the ctx was created with
 * (MVFIT_E_ARG otherwise).  mvfit_get_options returns what is in force. */
int mvfit_set_options(mvfit_ctx* ctx, const mvfit_options* opts);
int mvfit_get_options(const mvfit_ctx* ctx, mvfit_options* opts);
/* Which path served the last mvfit_sdf call (*op_path) and the SDF term of the last mvfit_fit / mvfit_closure (*term_path):
 * 0 the walk over every face (short face list, or sdf_face_lists = 0), 1 face lists, 2 the walk because the lists'
 * workspace (11.6 MB per problem at 13,776 faces) did not fit in half of the free device memory - decided once per shape
 * and remembered, a ~10x slower path that is never taken silently: the Python adapter warns. */
int mvfit_sdf_info(const mvfit_ctx* ctx, int* op_path, int* term_path);
void mvfit_destroy(mvfit_ctx* ctx);
const char* mvfit_last_error(const mvfit_ctx* ctx);
int mvfit_sync(mvfit_ctx* ctx);

/* The per-frame inputs of create_fitting_closure (code/utils/fitting.py:144-156; shapes from
 * non_linear_solver.py:77-84, code/init.py:112-131):
 *   cameras: cam_batched = 0 -> one rig [V,...] shared by all problems; 1 -> [B,V,...].
 *   gt_xy[B,V,17,2] ; w_conf[B,V,17] = joint_weights * conf (0 for missing views, main.py:49-57).
 * B = number of independent (subject x frame) problems. */
int mvfit_set_problems(mvfit_ctx* ctx, int B, int V, int cam_batched,
                       const float* cam_R /*[.,V,3,3] dev|host*/, const float* cam_t /*[.,V,3]*/,
                       const float* cam_f /*[.,V]*/, const float* cam_c /*[.,V,2]*/,
                       const float* gt_xy /*dev|host*/, const float* w_conf /*dev|host*/);

/* Optional 3-D joint targets of the use_3d term (code/utils/non_linear_solver.py:86-99):
 * gt3d[B,17,3], conf3d[B,17] (dev|host).  Call after mvfit_set_problems (which clears them). */
int mvfit_set_joints3d(mvfit_ctx* ctx, const float* gt3d, const float* conf3d);

/* One closure evaluation for all B problems: fitting_func(backward=True)
 * (code/utils/fitting.py:162-203) = SMPL.forward + SMPLifyLoss.forward + backward.
 *   params[B,MVFIT_D] dev ; loss[B] dev ; grad[B,MVFIT_D] dev or NULL (forward only) ;
 *   verts[B,Nv,3] dev or NULL ; joints[B,17,3] dev or NULL.
 * mvfit_options::closure_vposer_helpers = 1 (with MVFIT_F_VPOSER, no SDF term, B <= 160) decodes the body pose on helper
 * workgroups of the closure's own launch - the decoder arithmetic of the single-launch fits (another summation order than
 * the in-workgroup decoder, ~1e-7 relative) - so that the parity tests can hold the shipping decoder against the
 * closure-level goldens (tests/test_gpu_closure_helpers.py); mvfit_decoder_stats reports that launch. */
int mvfit_closure(mvfit_ctx* ctx, const mvfit_weights* w, const float* params,
                  float* loss, float* grad, float* verts, float* joints);

/* ModelOutput.full_pose of SMPL.forward (body_models_scale.py:392-412): [B,72] = global_orient | body_pose, the body pose
 * decoded from the embedding with MVFIT_F_VPOSER (fitting.py:170-173, VPoser.decode 'aa') - what the reference's
 * save_results stores as 'pose' / 'body_pose' (code/utils/utils.py:744-766).  params[B,MVFIT_D] dev, full_pose[B,72] dev. */
int mvfit_full_pose(mvfit_ctx* ctx, const float* params, uint32_t flags, float* full_pose);

/* SMPL.forward only (body_models_scale.py:327-412): vertices (+transl) and the 17 keypoints. */
int mvfit_vertices(mvfit_ctx* ctx, const float* params /*[B,MVFIT_D] dev*/, uint32_t flags,
                   float* verts /*[B,Nv,3] dev*/, float* joints /*[B,17,3] dev or NULL*/);

/* The whole staged fit, device resident: for each stage (non_linear_solver.py:156-211) a fresh
 * LBFGS (lbfgs_ls.py:256-445, strong-Wolfe :39-167) driven by run_fitting (fitting.py:99-142),
 * every problem advancing its own state machine, no host synchronisation per closure.
 *   params[B,MVFIT_D] dev, in/out ; stage_weights[num_stages] host ;
 *   final_loss[B] dev (run_fitting's return of the last stage; NaN where the reference returns None)
 *   n_closure[B], n_iter[B] dev int32 (closure evaluations / L-BFGS iterations spent), may be NULL.
 * The vertices of the trial points (the reference's return_verts=True) are computed per closure round into an internal
 * buffer and are NOT an output of this call - mvfit_vertices(params) gives the vertices of the result.  In the
 * asynchronous mode (the default; with the SDF term: for the stages whose coll_loss_weight is 0) those per-round passes
 * run beside the optimiser and nothing of the result depends on them; mvfit_fit_stats reports how many ran and whether
 * any was lost (none: the operand ring has back-pressure).  Any number of problems: batches beyond what is resident at
 * once are fitted in sub-batches, a problem's result does not depend on the slicing. */
int mvfit_fit(mvfit_ctx* ctx, const mvfit_weights* stage_weights, const mvfit_lbfgs_opts* opts,
              float* params, float* final_loss, int32_t* n_closure, int32_t* n_iter);

/* Counters of the vertex passes of the last mvfit_fit in its asynchronous mode (all zero in the other modes):
 *   out4[0]  chunk passes (32 problems x 6890 vertices) run;
 *   out4[1]  chunk passes skipped because all their problems had finished;
 *   out4[2]  pose operands lost (expected 0): per-round launches - (problem, round) operand sets overwritten before their pass
 *            read them; resident pass - the LARGEST number of (problem, round) operand sets any one of its workgroups found
 *            overwritten (every workgroup reports; such a round is skipped by that workgroup, never computed from another
 *            round's operands);
 *   out4[3]  waits given up (expected 0): gate kernels / resident workgroups that waited 20 ms for operands and left + problems
 *            whose optimiser waited 20 ms for the ring's back-pressure and stopped honouring it.
 * A non-zero out4[2] or out4[3] means "every closure round got its full vertex pass" does not hold for that fit (the fitted
 * parameters never depend on the passes); with mvfit_options::resident_pass = -1 the ctx then runs its later fits with
 * per-round launches (until mvfit_set_options is called). */
int mvfit_fit_stats(mvfit_ctx* ctx, uint32_t* out4);

/* Counters of the decoder helpers of the last mvfit_fit.  With MVFIT_F_VPOSER the single-launch fits (asynchronous and
 * objective-vertices-only) run the VPoser decoder's layers (VPoser.py:218-232) on helper workgroups of the same launch
 * that keep the weights in registers (csrc/vposer_service.h; mvfit_options::vposer_helpers = 0 keeps them in the
 * problems' own workgroups):
 * out3 = { launches that carried helpers, answers that did not arrive within 50 ms (expected 0: that problem decodes
 * locally from then on), helpers that gave up after 0.2 s without a request (expected 0) }.  Waits for the ctx stream. */
int mvfit_decoder_stats(mvfit_ctx* ctx, uint32_t* out3);

/* Test hook for the asynchronous fit: the vertex pass that belongs to closure round `round` (0-based, of every
 * problem) writes its vertices to verts[B,Nv,3] (dev) instead of the internal buffer; together with mvfit_fit_trace
 * (the trial points) this lets a test check that the pass of round r really computed the trial point of round r.
 * verts = NULL switches it off.  The buffer is sized for the CURRENT batch: mvfit_set_problems with another B (or V)
 * switches the hook off. */
int mvfit_debug_capture_pass(mvfit_ctx* ctx, int round, float* verts);

/* Optional closure trace of the NEXT mvfit_fit calls (test / debugging hook; the reference equivalent is printing
 * inside fitting_func): for every problem the first max_closures closure evaluations are recorded as
 *   trace[b][k][0:MVFIT_D] = the trial point the closure was evaluated at, trace[b][k][MVFIT_D] = its loss.
 * trace[B, max_closures, MVFIT_D + 1] dev, caller-owned, must stay valid until tracing is switched off with
 * mvfit_fit_trace(ctx, NULL, 0) (or the ctx is destroyed).  Rows beyond a problem's closure count are not written.
 * The buffer is sized for the CURRENT batch: mvfit_set_problems with another B (or V) switches tracing off. */
int mvfit_fit_trace(mvfit_ctx* ctx, float* trace, int max_closures);

/* The SDF voxelisation op (reference sdf/sdf/sdf.py:21-26 -> sdf_cuda.cpp:14-28 -> sdf_cuda_kernel.cu:242-335):
 *   faces[num_faces,3] int32 dev ; vertices[B,num_vertices,3] dev, coordinates in [-1,1] ; phi[B,G,G,G] dev out
 *   (phi[b,k,j,i]: i fastest = x).  num_faces is the caller's faces.size(0), exactly as the reference launcher
 *   takes it (sdf_cuda_kernel.cu:314; the reference's own call site passes a [1,F,3] tensor, i.e. ONE triangle).
 * Stand-alone op; the loss term below evaluates the same voxel function without materialising phi.
 * Face lists of 512 faces and more are voxelised on per-call face lists (exact culling: the same bits as the walk
 * over every face for every voxel, which mvfit_options::sdf_face_lists = 0 keeps; the workspace, 11.6 MB per batch
 * element at 13,776 faces, is kept in the ctx between calls of one shape). */
int mvfit_sdf(mvfit_ctx* ctx, const int32_t* faces, int num_faces, const float* vertices, int B,
              int num_vertices, int G, float* phi);

/* The interpenetration term of SMPLifyLoss.forward (code/utils/fitting.py:352-393, boxes :282-288):
 *   pen = (coll_loss_weight * sum_v grid_sample(phi, (v - c) / s))^2,  phi = SDF(faces, (v - c) / s, grid_size)
 * switched on for mvfit_closure / mvfit_fit whenever a weight set has coll_loss_weight > 0 (fitting.py:354).
 *   faces[num_faces,3] int32 (host or device), copied; num_faces = what the reference's call site makes the
 *   op see: it passes body_model_faces.reshape(1, -1, 3) (fitting.py:367-368), so the op's faces.size(0)
 *   is 1 and only the FIRST triangle is voxelised - pass num_faces = 1 for the reference's behaviour, the
 *   full face count for the behaviour its author presumably intended.  grid_size: 128 in the reference (:368).
 *   faces = NULL or num_faces = 0 removes the term.
 * Every problem is one person (the reference asserts batch size 1, :366): boxes, phi and the sum are per problem.
 * The term reads all vertices, so MVFIT_F_SPARSE_VERTS is ignored while it is active.
 * With 512 faces and more the sampled corners take their values from per-round face lists (same bits as the walk over
 * every face; mvfit_options::sdf_face_lists = 0 keeps the walk; a batch whose workspace would not fit in half of the free
 * memory keeps it too - mvfit_sdf_info says which path ran). */
int mvfit_set_sdf(mvfit_ctx* ctx, const int32_t* faces, int num_faces, int grid_size);

/* Diagnostics of the last evaluated interpenetration term (after mvfit_closure with coll_loss_weight > 0):
 *   samples[B,num_verts,4] dev out = (phi_v, d phi_v / d local x, y, z) per vertex (may be NULL),
 *   sums[B] dev out = S = sum_v phi_v (may be NULL). */
int mvfit_sdf_term_read(mvfit_ctx* ctx, float* samples, float* sums);

/* Per-frame initial guess, stage 1 (code/utils/init_guess.py:80-83 -> code/utils/recompute3D.py:22-62): weighted linear
 * triangulation of the 17 keypoints from V calibrated views, batched over B frames.
 *   keypoints[B,V,17,3] float32 dev (u, v, confidence) ; intris[V,3,3], extris[V,4,4] float64 dev (the reference
 *   keeps the camera file in float64, code/utils/utils.py:352-394) ; joints3d[B,17,3] float64 dev out.
 * Same arithmetic as the reference: float64 accumulation, AtA rounded to float32 before the float64 solve (:54). */
int mvfit_triangulate(mvfit_ctx* ctx, int B, int V, const float* keypoints, const double* intris, const double* extris,
                      double* joints3d);

/* Per-frame initial guess, stage 1 for single-view input (code/utils/init_guess.py:54-74): the depth guess that replaces
 * the triangulation when a frame has ONE view - the model's rest-pose keypoints pushed along the camera's z axis by
 * est_d = fx * (torso height in camera space) / (torso height in the image) and mapped back with inv(extri); the
 * reference's arithmetic is kept (the left shoulder-hip pair taken twice in the 2-D height, over (u, v, confidence)
 * rows in float32; everything else float64).  Batched over B frames seen by the same camera.
 *   rest_joints[17,3] float64 dev (the 17 keypoints of mvfit_vertices at zero pose / shape / translation and the start
 *   scale, init_guess.py:31-52) ; extri[4,4], intri[3,3] float64 dev ; keypoints[B,17,3] float32 dev (u, v, confidence)
 *   -> joints3d[B,17,3] float64 dev: what mvfit_umeyama takes as dst. */
int mvfit_depth_guess(mvfit_ctx* ctx, int B, const double* rest_joints, const double* extri, const double* intri,
                      const float* keypoints, double* joints3d);

/* Per-frame initial guess, stage 2 (code/utils/init_guess.py:95-106): similarity alignment src -> dst by the
 * reference's umeyama (code/utils/umeyama.py:16-109, incl. its full-rank formula U diag(d) Vh^T and the two-candidate
 * choice with the translation of the second candidate) and cv2.Rodrigues of the chosen rotation; batched over B frames
 * that share the source points (the rest-pose keypoints).  All float64 like the reference's NumPy.
 *   src[npts,3] dev, dst[B,npts,3] dev (npts = 4: the torso joints 5, 6, 11, 12 with use_torso, or 17) ->
 *   rot[B,3,3], rvec[B,3] (the model's global_orient), trans[B,3], scale[B] dev.
 * The signs of the singular-vector pairs - which the reference's formula is sensitive to and LAPACK chooses for it -
 * are LAPACK's own: the device SVD walks dgesdd's path for a 3 x 3 matrix (dgebd2, dbdsqr, dormbr; csrc/lapack_svd3.h)
 * and returns numpy's pairs, so rot / trans / scale equal the reference's (tests/golden/init_guess_ref.npz). */
int mvfit_umeyama(mvfit_ctx* ctx, int B, int npts, const double* src, const double* dst, int estimate_scale,
                  double* rot, double* rvec, double* trans, double* scale);

/* Per-view projection of point sets with the cameras of mvfit_set_problems: the reference's visualisation path
 * cam(verts) / cam(joints) per view (code/utils/utils.py:581-583,603-607; PerspectiveCamera.forward code/camera.py:93-117).
 *   points[B,num_points,3] dev (e.g. the vertices of mvfit_vertices, num_points = 6890) ->
 *   uv[B,V,num_points,2] dev, float pixels (the reference truncates to int32 on the host afterwards). */
int mvfit_project_points(mvfit_ctx* ctx, const float* points, int num_points, float* uv);

/* The path's only collective (north_star: "RCCL over xGMI only for the final gather"; in the Python adapters it is one
 * torch.distributed.all_gather, mvsmplfitting_amd/sharding.py): all-gather over the caller's RCCL communicator on the ctx
 * stream - rank r's bytes_per_rank bytes at `send` land at recv + r * bytes_per_rank on every rank.  For hosts that own
 * a communicator (a C / C++ driver); the reference has no counterpart (single process, code/main.py:60-120).
 *   rccl_comm: the host's ncclComm_t.  libmvfit does not link RCCL - ncclAllGather is bound at run time to the RCCL
 *   library already loaded in the process (the one the communicator belongs to); MVFIT_E_STATE if there is none.
 *   send[bytes_per_rank], recv[nranks * bytes_per_rank] dev; asynchronous like every other call (mvfit_sync). */
int mvfit_gather(mvfit_ctx* ctx, void* rccl_comm, const void* send, void* recv, size_t bytes_per_rank);

/* Timing hook for bench.py: average duration (ms) of the LBS vertex-pass kernel launches since
 * the last call, measured with hipEvents on the ctx stream; *launches = number measured.
 * Enable with mvfit_profile(ctx, 1) (adds two event records per launch). */
int mvfit_profile(mvfit_ctx* ctx, int enable);
int mvfit_profile_read(mvfit_ctx* ctx, double* vertex_pass_ms_avg, int* launches,
                       double* step_kernel_ms_avg, int* step_launches);
/* `launches` back-to-back launches of the vertex pass (pose operands as the last closure / fit left them)
 * inside ONE hipEvent pair on the ctx stream; *avg_ms = elapsed / launches.  A pair around a single launch
 * (mvfit_profile_read) contains the markers' own few microseconds; this amortises them. */
int mvfit_profile_vertex_pass(mvfit_ctx* ctx, int launches, double* avg_ms);
/* flavour 0: as above; 1: the pass exactly as the asynchronous fit launches it (operands from its ring, non-temporal
 * basis stream and vertex stores, no side outputs) - needs a preceding asynchronous mvfit_fit on this batch. */
int mvfit_profile_vertex_pass_ex(mvfit_ctx* ctx, int launches, int flavour, double* avg_ms);
/* How the vertex passes of the last asynchronous mvfit_fit ran:
 *   *tiles_per_wg  the form of mvfit_options::resident_pass that ran.  1 / 2 / 3: the RESIDENT pass - one launch per (sub-batch)
 *                  fit whose workgroups keep the blendshape basis of their one / two / two vertex tile(s) in registers and serve
 *                  closure round after closure round from the operand ring (the basis crosses the memory system once per fit;
 *                  3 = two tiles with the workgroup split into contraction and worker waves, the automatic choice beside more
 *                  than 36 optimiser workgroups); 0: one gate + one pass launch per closure round (dense skinning rows,
 *                  exact-fp32 contraction, or launches that leave no CUs for resident workgroups);
 *   *workgroups    workgroups of the resident pass (ceil(tiles / tiles per workgroup));
 * and, when the fit ran under mvfit_profile(ctx, 1) with the resident pass, what its workgroups stamped per closure round
 * (wall clock, 10 ns ticks; the first 1024 rounds of the last sub-batch):
 *   *rounds        rounds stamped;
 *   *span_ms       mean over the rounds of (last workgroup's vertex stores acknowledged - first workgroup saw the round's
 *                  operands): the in-fit service time of a round - what mvfit_profile_read reports as the launch duration;
 *   *busy_ms       mean over rounds and workgroups of a workgroup's own (stores acknowledged - operands seen);
 *   *slowest_ms    mean over the rounds of the SLOWEST workgroup's (stores acknowledged - operands seen): the rate at which the
 *                  pass can serve rounds (when the passes are slower than the optimiser the workgroups drift apart by up to the
 *                  ring's depth and the span of a round says nothing about that rate).
 * Any pointer may be NULL.  flavour 2 of mvfit_profile_vertex_pass_ex runs the resident pass ALONE over `launches` (<= 128)
 * rounds whose operands the last fit left in the ring: one kernel launch inside one hipEvent pair, avg_ms = elapsed / rounds. */
int mvfit_pass_profile(mvfit_ctx* ctx, int* tiles_per_wg, int* workgroups, int* rounds, double* span_ms, double* busy_ms,
                       double* slowest_ms);

/* Known-answer test entry for the device L-BFGS state machine (same template as production,
 * instantiated in float64) on the analytic objectives of oracle/lbfgs_np.py:kat_objective.
 *   kind: 0 quad, 1 rosen, 2 gmof ; D <= 96 ; x_inout[D] host ; trace[max_trace,(D+1)] host
 *   (x_trial, loss per closure) ; segs[nseg+1] parameter-tensor boundaries for the gtol test. */
int mvfit_lbfgs_kat(int device, int kind, int D, const int32_t* segs, int nseg,
                    const mvfit_lbfgs_opts* opts, double* x_inout, double* trace, int max_trace,
                    int* n_closure, double* final_loss);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MVFIT_H_ */

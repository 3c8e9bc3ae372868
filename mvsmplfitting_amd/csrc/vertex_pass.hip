// LBS vertex pass: verts[b] = skin( v_template + shapedirs.beta_b + posedirs.pose_feature_b ) + transl_b
// for every vertex and every problem of a 32-problem chunk.
//
// Restates lbs() steps 1,3,5 of the reference (code/smplx/lbs.py:179,192-203,207-220) and the
// "+ transl" of SMPL.forward (code/smplx/body_models_scale.py:401-403); the joint regression
// J(beta) and the kinematic chain (lbs.py:183,205) are done per problem by the step kernel.
//
// MI355X mapping
//   * one workgroup = one tile of 32 vertices x one chunk of 32 problems; 3*KSPLIT waves.
//   * the blendshape contraction [32 problems x 224] . [224 x (32 verts x 3 coords)] runs on the
//     matrix cores in exact fp32 (v_mfma_f32_32x32x2_f32): wave (coord, kslice) owns one 32x32
//     accumulator.  The basis is pre-tiled in HBM in B-operand order so every wave load is one
//     contiguous 1 KiB global_load_dwordx4 covering 4 k-steps; each basis element is read once
//     per chunk.  The A operand (coefficients, transposed per chunk by the step kernel) is
//     staged once in LDS and read conflict-free (64 consecutive floats per k-step).
//   * k-slice partials meet in LDS in a fixed order (deterministic), then the skinning blend
//     T = W.A and the affine apply run on the VALU while other workgroups' MFMAs proceed.
//   * tile index = blockIdx.x, so the tile -> XCD assignment (block b -> XCD b%8) is the same in
//     every launch: each XCD keeps its 1/8 of the 18 MB basis in its own 4 MiB L2 across the
//     closure rounds of a fit.
#include "mvfit_device.h"

namespace mvfit {

typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int KSPLIT>
__global__ __launch_bounds__(192 * KSPLIT) void lbs_vertex_pass_kernel(DevModel M, DevPose P, int B,
                                                                       float* __restrict__ verts) {
    constexpr int NT = 192 * KSPLIT;
    constexpr int GPS = KGROUPS / KSPLIT;           // k-groups per slice
    static_assert(KGROUPS % KSPLIT == 0, "KSPLIT must divide 28");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* coefT_l = smem;                          // [KROWS][32]
    float* A_l = coefT_l + KROWS * 32;              // [32][A_STRIDE]
    float* Wt_l = A_l + 32 * A_STRIDE;              // [24][32]
    float* tau_l = Wt_l + NJ * 32;                  // [32][4]
    float* part = tau_l + 32 * 4;                   // [KSPLIT][3][32][33]
    float* vp_l = part + KSPLIT * 3 * 32 * 33;      // [32 b][32 v][4]  (x,y,z,pad)
    float* out_l = part;                            // alias: [32 b][96] after the partials are consumed

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int tile = blockIdx.x;
    const int chunk = blockIdx.y;
    const int b0 = chunk * 32;

    // ---- stage the per-chunk operands in LDS ----
    {
        const float4* src = reinterpret_cast<const float4*>(P.coefT + (size_t)chunk * KROWS * 32);
        float4* dst = reinterpret_cast<float4*>(coefT_l);
        for (int i = tid; i < KROWS * 32 / 4; i += NT) dst[i] = src[i];
        for (int i = tid; i < 32 * 288; i += NT) {
            int b = i / 288, e = i - b * 288;
            A_l[b * A_STRIDE + e] = (b0 + b < B) ? P.Amat[(size_t)(b0 + b) * 288 + e] : 0.f;
        }
        const float* wsrc = M.wt_tiles + (size_t)tile * NJ * 32;
        for (int i = tid; i < NJ * 32; i += NT) Wt_l[i] = wsrc[i];
        if (tid < 128) {
            int b = tid >> 2;
            tau_l[tid] = (b0 + b < B) ? P.tau[(size_t)(b0 + b) * 4 + (tid & 3)] : 0.f;
        }
    }
    __syncthreads();

    // ---- blendshape contraction on the matrix cores ----
    {
        const int k = wave % 3;           // coordinate plane
        const int ks = wave / 3;          // k-slice
        floatx16 acc;
        float init = (ks == 0) ? M.vt_planes[k * M.nv_pad + tile * TILE_V + (lane & 31)] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = init;
        const float4* bsrc = reinterpret_cast<const float4*>(M.bs4) +
                             ((size_t)(tile * 3 + k) * KGROUPS + ks * GPS) * 64 + lane;
        const float* asrc = coefT_l + (lane >> 5) * 32 + (lane & 31);
        float4 bv[GPS];
#pragma unroll
        for (int g = 0; g < GPS; ++g) bv[g] = bsrc[g * 64];
#pragma unroll
        for (int g = 0; g < GPS; ++g) {
            const int kk0 = (ks * GPS + g) * 4;
            float a0 = asrc[(2 * (kk0 + 0)) * 32];
            float a1 = asrc[(2 * (kk0 + 1)) * 32];
            float a2 = asrc[(2 * (kk0 + 2)) * 32];
            float a3 = asrc[(2 * (kk0 + 3)) * 32];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[g].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[g].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, bv[g].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, bv[g].w, acc, 0, 0, 0);
        }
        // D layout: col (vertex) = lane&31, row (problem) = (r&3) + 8*(r>>2) + 4*(lane>>5)
        float* pdst = part + ((ks * 3 + k) * 32) * 33 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int b = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            pdst[b * 33] = acc[r];
        }
    }
    __syncthreads();

    // ---- combine k-slices (fixed order) -> v_posed[b][v][coord] ----
    for (int t = tid; t < 32 * 3 * 8; t += NT) {
        const int vg = t & 7, bk = t >> 3, k = bk % 3, b = bk / 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = 4 * vg + i;
            float s = part[((0 * 3 + k) * 32 + b) * 33 + v];
#pragma unroll
            for (int q = 1; q < KSPLIT; ++q) s += part[((q * 3 + k) * 32 + b) * 33 + v];
            vp_l[(b * 32 + v) * 4 + k] = s;
        }
    }
    __syncthreads();

    // ---- skinning: row k of T = sum_j W[v][j] A_b[j], applied to v_posed; + transl ----
    for (int t = tid; t < 32 * 3 * 8; t += NT) {
        const int vg = t & 7, bk = t >> 3, k = bk % 3, b = bk / 3;
        float tr[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) tr[i][e] = 0.f;
        const float* arow = A_l + b * A_STRIDE + 4 * k;
#pragma unroll 4
        for (int j = 0; j < NJ; ++j) {
            const float4 w = *reinterpret_cast<const float4*>(Wt_l + j * 32 + 4 * vg);
            const float4 a = *reinterpret_cast<const float4*>(arow + j * 12);
            const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                tr[i][0] = fmaf(wv[i], a.x, tr[i][0]);
                tr[i][1] = fmaf(wv[i], a.y, tr[i][1]);
                tr[i][2] = fmaf(wv[i], a.z, tr[i][2]);
                tr[i][3] = fmaf(wv[i], a.w, tr[i][3]);
            }
        }
        float xo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 vp = *reinterpret_cast<const float4*>(vp_l + (b * 32 + 4 * vg + i) * 4);
            xo[i] = fmaf(tr[i][0], vp.x, fmaf(tr[i][1], vp.y, fmaf(tr[i][2], vp.z, tr[i][3])));
        }
        // out_l aliases the (now dead) partial slabs only after every thread passed the combine
        // barrier above; the slabs are not read again in this loop.  It holds the skinned position
        // BEFORE "+ transl" (the step kernel wants that value; the store below adds transl).
#pragma unroll
        for (int i = 0; i < 4; ++i) out_l[b * 96 + (4 * vg + i) * 3 + k] = xo[i];
    }
    __syncthreads();

    // ---- coalesced store of x + transl: 32 rows of 96 floats (8-byte aligned: 12*6890 % 8 == 0) ----
    {
        const int vbase = tile * TILE_V;
        const int nvalid = min(TILE_V, M.nv - vbase) * 3;      // floats valid in this tile row
        for (int i = tid; i < 32 * 48; i += NT) {
            const int b = i / 48, q = i - b * 48;
            if (b0 + b >= B) continue;
            float* dst = verts + ((size_t)(b0 + b) * M.nv + vbase) * 3 + 2 * q;
            const int k0 = (2 * q) % 3, k1 = (2 * q + 1) % 3;
            float2 o = *reinterpret_cast<const float2*>(out_l + b * 96 + 2 * q);
            o.x += tau_l[b * 4 + k0];
            o.y += tau_l[b * 4 + k1];
            if (2 * q + 1 < nvalid) {
                *reinterpret_cast<float2*>(dst) = o;
            } else if (2 * q < nvalid) {
                dst[0] = o.x;
            }
        }
    }
    // ---- side outputs for the vertices the objective reads (consumed by the step kernel) ----
    {
        const int s0 = M.tile_sel_start[tile], nsel = M.tile_sel_start[tile + 1] - s0;
        for (int i = tid; i < nsel * 96; i += NT) {
            const int sl = i / 96, rem = i - sl * 96, b = rem / 3, k = rem - 3 * b;
            if (b0 + b >= B) continue;
            const int lv = M.tile_sel_local[s0 + sl], slot = M.tile_sel_slot[s0 + sl];
            P.vposed_sel[(size_t)(b0 + b) * NC_MAX + 3 * slot + k] = vp_l[(b * 32 + lv) * 4 + k];
            P.xs_sel[(size_t)(b0 + b) * NC_MAX + 3 * slot + k] = out_l[b * 96 + lv * 3 + k];
        }
    }
}

size_t vertex_pass_lds_bytes(int ksplit) {
    return sizeof(float) * (size_t)(KROWS * 32 + 32 * A_STRIDE + NJ * 32 + 32 * 4 +
                                    ksplit * 3 * 32 * 33 + 32 * 32 * 4);
}

hipError_t launch_vertex_pass(const DevModel& M, const DevPose& P, int B, float* verts, int ksplit,
                              hipStream_t stream) {
    dim3 grid(M.ntiles, (B + 31) / 32);
    size_t lds = vertex_pass_lds_bytes(ksplit);
    switch (ksplit) {
        case 1:
            hipLaunchKernelGGL(lbs_vertex_pass_kernel<1>, grid, dim3(192), lds, stream, M, P, B, verts);
            break;
        case 2:
            hipLaunchKernelGGL(lbs_vertex_pass_kernel<2>, grid, dim3(384), lds, stream, M, P, B, verts);
            break;
        default:
            hipLaunchKernelGGL(lbs_vertex_pass_kernel<4>, grid, dim3(768), lds, stream, M, P, B, verts);
            break;
    }
    return hipGetLastError();
}

hipError_t vertex_pass_configure() {
    hipError_t e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(lbs_vertex_pass_kernel<1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)vertex_pass_lds_bytes(1));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(lbs_vertex_pass_kernel<2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)vertex_pass_lds_bytes(2));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(lbs_vertex_pass_kernel<4>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)vertex_pass_lds_bytes(4));
}

}  // namespace mvfit

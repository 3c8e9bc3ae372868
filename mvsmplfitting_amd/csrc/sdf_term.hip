// The SDF interpenetration term of SMPLifyLoss.forward (reference code/utils/fitting.py:282-288, :352-393)
// on the device, for a batch of independent one-person problems:
//
//   boxes / centre / scale (:356-359)  ->  sdf_bbox_kernel      one workgroup per problem
//   phi = SDF(faces, (v-c)/s, G) (:361-369) and phi_v = grid_sample(phi, (v-c)/s) (:375-383)
//                                      ->  sdf_sample_kernel    thread per (problem, vertex)
//   adjoint of S = sum_v phi_v w.r.t. the pose operands of the vertex pass
//                                      ->  sdf_adjoint_kernel   one workgroup per problem
//
// The reference voxelises G^3 = 2 M voxels per closure and then samples 6890 x 8 of them; phi is a pure
// function of the voxel index, so the sample kernel evaluates exactly those <= 55 k voxels on the fly with the
// op's own per-voxel code (sdf_device.h: bit-identical values, 38x fewer voxel evaluations, no 8 MB grid per
// problem).  pen = (w S)^2 and the factor 2 w^2 S are applied by the closure kernel, which knows the stage
// weight; this file produces S and dS/d(A, coef, transl).
//
// dS/dvertex has two parts: grid_sample's coordinate gradient / s, and the bounding box (centre and scale
// are differentiable functions of the arg-min / arg-max vertices).  Only vertices whose 8 corners touch a
// non-zero voxel - plus the <= 6 box vertices - carry gradient; they are compacted into an entry list and
// pulled back through skinning and the blendshape basis (vertex-major copies M.bs_vm / M.w_vm) entry by
// entry: phase 1 (wave per entry) recomputes v_posed and the blended transform, phase 2 (thread per output)
// accumulates g_A[24][12] and g_coef[224] in entry order, i.e. deterministically.
#include "sdf_device.h"
#include "mvfit_device.h"
#include "wave_ops.h"

namespace mvfit {

#pragma clang fp contract(off)

constexpr int SDF_ADJ_NT = 512;
constexpr int SDF_EB = 256;          // entries staged per batch in phase 2

__device__ __forceinline__ void better_min(float& v, int& i, float ov, int oi) { if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; } }
__device__ __forceinline__ void better_max(float& v, int& i, float ov, int oi) { if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; } }

// fitting.py:282-288 + :356-359.  First-occurrence arg indices (ties: lowest vertex index).
__global__ __launch_bounds__(512) void sdf_bbox_kernel(const float* __restrict__ verts, int nv, const int* __restrict__ gate,
                                                       SdfBox* __restrict__ box) {
    __shared__ float s_v[8][6];
    __shared__ int s_i[8][6];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (gate && !gate[b]) return;                       // the problem's current stage has no SDF term (uniform)
    const float* vb = verts + (size_t)b * nv * 3;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    int ilo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, ihi[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
    for (int v = tid; v < nv; v += 512)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float x = vb[3 * v + a];
            better_min(lo[a], ilo[a], x, v);
            better_max(hi[a], ihi[a], x, v);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            better_min(lo[a], ilo[a], __shfl_xor(lo[a], off), __shfl_xor(ilo[a], off));
            better_max(hi[a], ihi[a], __shfl_xor(hi[a], off), __shfl_xor(ihi[a], off));
        }
    if (lane == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_v[wave][a] = lo[a]; s_i[wave][a] = ilo[a]; s_v[wave][3 + a] = hi[a]; s_i[wave][3 + a] = ihi[a]; }
    __syncthreads();
    if (tid == 0) {
        SdfBox o;
        float ext[3];
        for (int a = 0; a < 3; ++a) {
            float l = s_v[0][a], h = s_v[0][3 + a];
            int il = s_i[0][a], ih = s_i[0][3 + a];
            for (int w = 1; w < 8; ++w) { better_min(l, il, s_v[w][a], s_i[w][a]); better_max(h, ih, s_v[w][3 + a], s_i[w][3 + a]); }
            o.c[a] = (l + h) / 2.f;                                  // boxes.mean(dim=1)
            o.imin[a] = il; o.imax[a] = ih;
            ext[a] = h - l;
        }
        int am = 0;
        if (ext[1] > ext[am]) am = 1;
        if (ext[2] > ext[am]) am = 2;
        o.amax = am;
        o.s = (float)((1 + 0.2) * 0.5) * ext[am];                    // "(1+0.2) * 0.5 * (...)": Python double meets a float tensor
        o.pad = 0;
        box[b] = o;
    }
}

// thread per (problem, vertex): samp[b][v] = (phi_v, dphi_v/dloc x, y, z).
__global__ __launch_bounds__(SDF_NT) void sdf_sample_kernel(const float* __restrict__ verts, int nv, const SdfBox* __restrict__ box,
                                                            const int32_t* __restrict__ faces, int num_faces, int G,
                                                            const int* __restrict__ gate, float4* __restrict__ samp) {
    __shared__ SdfTri tri[SDF_CH];
    const int b = blockIdx.y, v = blockIdx.x * SDF_NT + threadIdx.x;
    if (gate && !gate[b]) return;
    const bool live = v < nv;
    const SdfBox bx = box[b];
    const float* vb = verts + (size_t)b * nv * 3;
    // sampling coordinates (fitting.py:377-379) and grid_sample's unnormalisation (align_corners=False)
    float loc[3] = {0.f, 0.f, 0.f}, fr[3];
    int i0[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (live) loc[a] = (vb[3 * v + a] - bx.c[a]) / bx.s;
        const float pix = ((loc[a] + 1.f) * (float)G - 1.f) / 2.f;
        const float fl = floorf(pix);
        i0[a] = (int)fl;
        fr[a] = pix - fl;
    }
    // the 8 corners: bit q of corner index = +1 along axis q
    unsigned inrange = 0, parity = 0;
#pragma unroll
    for (int cn = 0; cn < 8; ++cn) {
        const int ix = i0[0] + (cn & 1), iy = i0[1] + ((cn >> 1) & 1), iz = i0[2] + (cn >> 2);
        if (live && ix >= 0 && ix < G && iy >= 0 && iy < G && iz >= 0 && iz < G) inrange |= 1u << cn;
    }
    // pass 1: crossing parity of every in-range corner over all faces (sdf_cuda_kernel.cu:281-287)
    for (int f0 = 0; f0 < num_faces; f0 += SDF_CH) {
        const int nf = min(SDF_CH, num_faces - f0);
        __syncthreads();
        for (int t = threadIdx.x; t < nf; t += SDF_NT) {
            float p[3][3];
            for (int m = 0; m < 3; ++m) {
                const int vi = faces[3 * (f0 + t) + m];
                for (int a = 0; a < 3; ++a) p[m][a] = (vb[3 * vi + a] - bx.c[a]) / bx.s;     // fitting.py:362-363
            }
            sdf_tri_setup(tri[t], p[0], p[1], p[2]);
        }
        __syncthreads();
        if (inrange) {
            for (int cn = 0; cn < 8; ++cn) {
                if (!((inrange >> cn) & 1)) continue;
                const float c[3] = {sdf_voxel_coord(i0[0] + (cn & 1), G), sdf_voxel_coord(i0[1] + ((cn >> 1) & 1), G),
                                    sdf_voxel_coord(i0[2] + (cn >> 2), G)};
                int n = 0;
                for (int t = 0; t < nf; ++t) n += sdf_ray_hit(tri[t], c) ? 1 : 0;
                if (n & 1) parity ^= 1u << cn;
            }
        }
    }
    // pass 2: min distance over all faces, only for the corners that are inside (odd parity)
    float pv[8];
#pragma unroll
    for (int cn = 0; cn < 8; ++cn) pv[cn] = 1000.f;
    const bool any_inside = __syncthreads_or(parity != 0);
    if (any_inside) {
        for (int f0 = 0; f0 < num_faces; f0 += SDF_CH) {
            const int nf = min(SDF_CH, num_faces - f0);
            if (num_faces > SDF_CH) {                    // a single chunk is still staged from pass 1
                __syncthreads();
                for (int t = threadIdx.x; t < nf; t += SDF_NT) {
                    float p[3][3];
                    for (int m = 0; m < 3; ++m) {
                        const int vi = faces[3 * (f0 + t) + m];
                        for (int a = 0; a < 3; ++a) p[m][a] = (vb[3 * vi + a] - bx.c[a]) / bx.s;
                    }
                    sdf_tri_setup(tri[t], p[0], p[1], p[2]);
                }
                __syncthreads();
            }
            if (parity) {
#pragma unroll
                for (int cn = 0; cn < 8; ++cn) {
                    if (!((parity >> cn) & 1)) continue;
                    const float c[3] = {sdf_voxel_coord(i0[0] + (cn & 1), G), sdf_voxel_coord(i0[1] + ((cn >> 1) & 1), G),
                                        sdf_voxel_coord(i0[2] + (cn >> 2), G)};
                    float md = pv[cn];
                    for (int t = 0; t < nf; ++t) { const float d = sdf_tri_distance(tri[t], c); if (d < md) md = d; }
                    pv[cn] = md;
                }
            }
        }
    }
    // trilinear interpolation and its coordinate gradient (zeros padding: out-of-range corners are 0)
    float val = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int cn = 0; cn < 8; ++cn) {
        const float p = ((parity >> cn) & 1) ? pv[cn] : 0.f;
        const float wx = (cn & 1) ? fr[0] : 1.f - fr[0];
        const float wy = (cn & 2) ? fr[1] : 1.f - fr[1];
        const float wz = (cn & 4) ? fr[2] : 1.f - fr[2];
        val += p * wx * wy * wz;
        gx += ((cn & 1) ? p : -p) * wy * wz;
        gy += ((cn & 2) ? p : -p) * wx * wz;
        gz += ((cn & 4) ? p : -p) * wx * wy;
    }
    const float hg = (float)G / 2.f;
    if (live) samp[(size_t)b * nv + v] = make_float4(val, gx * hg, gy * hg, gz * hg);
}

// block-wide sum of a double, same bits in every thread (fixed order: lanes, then waves ascending)
__device__ __forceinline__ double block_sum(double v, double* sh, int tid) {
    v = wave64_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < SDF_ADJ_NT / 64; ++w) s += sh[w];
    return s;
}

// entry list of one problem: [nv] SdfEntry written by the compaction, then [nv] SdfPull written by phase 1
// (separate arrays: no cache line is read before another wave writes into it)
struct SdfEntry { int v; float g[3]; };                                  // vertex, dS/dvertex
struct SdfPull { float vposed[3]; float gvp[3]; float pad[2]; };         // v_posed, Tr^T g
static_assert(sizeof(SdfEntry) == 16 && sizeof(SdfPull) == 32, "entry layout");

__global__ __launch_bounds__(SDF_ADJ_NT) void sdf_adjoint_kernel(DevModel M, const float* __restrict__ verts,
                                                                 const SdfBox* __restrict__ box, const float4* __restrict__ samp,
                                                                 DevPose P, const int* __restrict__ gate,
                                                                 unsigned char* __restrict__ entries, SdfAdj* __restrict__ out) {
    __shared__ double sh_d[8];
    __shared__ int sh_cnt[SDF_ADJ_NT];
    __shared__ float sh_coef[KROWS];
    __shared__ float sh_A[NJ * 12];
    __shared__ __attribute__((aligned(16))) SdfEntry sh_e[SDF_EB];
    __shared__ __attribute__((aligned(16))) SdfPull sh_p[SDF_EB];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (gate && !gate[b]) return;
    const int nv = M.nv;
    const SdfBox bx = box[b];
    const float* vb = verts + (size_t)b * nv * 3;
    const float4* sb = samp + (size_t)b * nv;
    SdfEntry* eb = reinterpret_cast<SdfEntry*>(entries + (size_t)b * nv * 48);
    SdfPull* pb = reinterpret_cast<SdfPull*>(entries + (size_t)b * nv * 48 + (size_t)nv * 16);
    if (tid < KROWS) sh_coef[tid] = P.coefT[(size_t)(b >> 5) * KROWS * 32 + (size_t)tid * 32 + (b & 31)];
    if (tid >= 224 && tid < 224 + NJ * 12) sh_A[tid - 224] = P.Amat[(size_t)b * NJ * 12 + (tid - 224)];
    // ---- reductions over the vertices: S, g_c = -sum g/s, g_s = -sum g.loc/s ----
    const int per = (nv + SDF_ADJ_NT - 1) / SDF_ADJ_NT;
    const int v0 = tid * per, v1 = min(nv, v0 + per);
    double S = 0.0, gc[3] = {0.0, 0.0, 0.0}, gs = 0.0;
    int cnt = 0;
    for (int v = v0; v < v1; ++v) {
        const float4 q = sb[v];
        S += (double)q.x;
        const float g[3] = {q.y, q.z, q.w};
        bool act = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float loc = (vb[3 * v + a] - bx.c[a]) / bx.s;
            gc[a] -= (double)(g[a] / bx.s);
            gs -= (double)(g[a] * loc / bx.s);
            act |= g[a] != 0.f;
            act |= (v == bx.imin[a]) | (v == bx.imax[a]);
        }
        cnt += act ? 1 : 0;
    }
    S = block_sum(S, sh_d, tid);
    gc[0] = block_sum(gc[0], sh_d, tid); gc[1] = block_sum(gc[1], sh_d, tid); gc[2] = block_sum(gc[2], sh_d, tid);
    gs = block_sum(gs, sh_d, tid);
    // ---- compaction: exclusive scan of the per-thread counts (threads own ascending vertex ranges) ----
    sh_cnt[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < SDF_ADJ_NT; off <<= 1) {
        const int add = tid >= off ? sh_cnt[tid - off] : 0;
        __syncthreads();
        sh_cnt[tid] += add;
        __syncthreads();
    }
    const int n = sh_cnt[SDF_ADJ_NT - 1];
    int pos = sh_cnt[tid] - cnt;
    const float gcf[3] = {(float)gc[0], (float)gc[1], (float)gc[2]};
    const float gsf = (float)gs * (float)((1 + 0.2) * 0.5);
    for (int v = v0; v < v1; ++v) {
        const float4 q = sb[v];
        float g[3] = {q.y / bx.s, q.z / bx.s, q.w / bx.s};
        bool act = (q.y != 0.f) | (q.z != 0.f) | (q.w != 0.f);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (v == bx.imin[a]) { g[a] += gcf[a] / 2.f; act = true; if (a == bx.amax) g[a] -= gsf; }
            if (v == bx.imax[a]) { g[a] += gcf[a] / 2.f; act = true; if (a == bx.amax) g[a] += gsf; }
        }
        if (act) { eb[pos].v = v; eb[pos].g[0] = g[0]; eb[pos].g[1] = g[1]; eb[pos].g[2] = g[2]; ++pos; }
    }
    __syncthreads();            // entries of this workgroup are visible to it (workgroup-scope release/acquire)
    // ---- phase 1: wave per entry - v_posed (lbs.py:179,203), blended transform (lbs.py:209-213), g_vposed = Tr^T g ----
    for (int e = wave; e < n; e += SDF_ADJ_NT / 64) {
        const int v = eb[e].v;
        const float g0 = eb[e].g[0], g1 = eb[e].g[1], g2 = eb[e].g[2];
        const float* bsv = M.bs_vm + (size_t)v * 3 * KROWS;
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = lane + 64 * q;
            if (p < KROWS) {
                const float cf = sh_coef[p];
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[k] += cf * bsv[k * KROWS + p];
            }
        }
        float vp[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) vp[k] = M.vt_planes[(size_t)k * M.nv_pad + v] + wave64_sum(acc[k]);
        float tl = 0.f;
        if (lane < 12)
            for (int j = 0; j < NJ; ++j) tl += M.w_vm[(size_t)v * NJ + j] * sh_A[j * 12 + lane];
        float T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = lane_read(tl, i);
        if (lane == 0) {
            pb[e].vposed[0] = vp[0]; pb[e].vposed[1] = vp[1]; pb[e].vposed[2] = vp[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) pb[e].gvp[k] = T[k] * g0 + T[4 + k] * g1 + T[8 + k] * g2;
        }
    }
    __syncthreads();
    // ---- phase 2: thread per output, entries in ascending order ----
    float acc = 0.f;
    const int oj = tid >= KROWS ? (tid - KROWS) / 12 : 0, oe = tid >= KROWS ? (tid - KROWS) % 12 : 0;
    float gt[3] = {0.f, 0.f, 0.f};
    for (int e0 = 0; e0 < n; e0 += SDF_EB) {
        const int ne = min(SDF_EB, n - e0);
        __syncthreads();
        for (int i = tid; i < ne * 4; i += SDF_ADJ_NT) reinterpret_cast<float*>(sh_e)[i] = reinterpret_cast<const float*>(eb + e0)[i];
        for (int i = tid; i < ne * 8; i += SDF_ADJ_NT) reinterpret_cast<float*>(sh_p)[i] = reinterpret_cast<const float*>(pb + e0)[i];
        __syncthreads();
        for (int e = 0; e < ne; ++e) {
            const SdfEntry& E = sh_e[e];
            const SdfPull& Q = sh_p[e];
            if (tid < KROWS) {
                const float* bsv = M.bs_vm + (size_t)E.v * 3 * KROWS + tid;
                acc += bsv[0] * Q.gvp[0] + bsv[KROWS] * Q.gvp[1] + bsv[2 * KROWS] * Q.gvp[2];
                if (tid < 3) gt[tid] += E.g[tid];
            } else {
                const float w = M.w_vm[(size_t)E.v * NJ + oj];
                acc += w * (oe < 9 ? E.g[oe / 3] * Q.vposed[oe % 3] : E.g[oe - 9]);
            }
        }
    }
    SdfAdj& O = out[b];
    if (tid < KROWS) O.gcoef[tid] = acc; else O.gA[tid - KROWS] = acc;
    if (tid < 3) O.gtau[tid] = gt[tid];
    if (tid == 0) { O.S = (float)S; }
}

hipError_t launch_sdf_term(const DevModel& M, const DevPose& P, const float* verts, int B, const int32_t* faces, int num_faces,
                           int G, const int* gate, SdfBox* box, float4* samp, void* entries, SdfAdj* adj, hipStream_t stream) {
    hipLaunchKernelGGL(sdf_bbox_kernel, dim3(B), dim3(512), 0, stream, verts, M.nv, gate, box);
    hipLaunchKernelGGL(sdf_sample_kernel, dim3((M.nv + SDF_NT - 1) / SDF_NT, B), dim3(SDF_NT), 0, stream, verts, M.nv,
                       (const SdfBox*)box, faces, num_faces, G, gate, samp);
    hipLaunchKernelGGL(sdf_adjoint_kernel, dim3(B), dim3(SDF_ADJ_NT), 0, stream, M, verts, (const SdfBox*)box,
                       (const float4*)samp, P, gate, reinterpret_cast<unsigned char*>(entries), adj);
    return hipGetLastError();
}

size_t sdf_entry_bytes() { return 48; }     // per vertex: SdfEntry + SdfPull

}  // namespace mvfit

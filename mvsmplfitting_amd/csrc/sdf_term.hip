// The SDF interpenetration term of SMPLifyLoss.forward (reference code/utils/fitting.py:282-288, :352-393)
// on the device, for a batch of independent one-person problems:
//
//   boxes / centre / scale (:356-359)  ->  sdf_bbox_kernel      one workgroup per problem
//   phi = SDF(faces, (v-c)/s, G) (:361-369) and phi_v = grid_sample(phi, (v-c)/s) (:375-383)
//                                      ->  sdf_sample_kernel    thread per (problem, vertex)
//   adjoint of S = sum_v phi_v w.r.t. the pose operands of the vertex pass
//                                      ->  sdf_entries_kernel   8 workgroups per problem: S, box-adjoint sums, entry lists
//                                          sdf_pullback_kernel  8 workgroups per problem: entries through skinning / basis
//                                          (slice partials added in slice order by the problem's last-arriving pull-back workgroup)
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
// entry, a batch of 256 entries at a time in LDS: thread per (entry, coordinate) recomputes v_posed, thread per
// (entry, element) the blended transform, then thread per output accumulates g_A[24][12] and g_coef[224] in
// entry order, i.e. deterministically.
#include "sdf_device.h"
#include "mvfit_device.h"
#include "wave_ops.h"

namespace mvfit {

#pragma clang fp contract(off)

constexpr int SDF_ADJ_NT = 512;
constexpr int SDF_EB = 256;          // entries staged per batch in phase 2

// (ord_bits / ord_float / dpp_u64 / row16_key: wave_ops.h - shared with the vertex pass, which reduces its tile's keys itself)
typedef unsigned long long u64;
// whole-wave min / max of 64-bit keys on the DPP path (wave_ops.h), same result in every lane
template <bool MIN>
__device__ __forceinline__ u64 wave64_key(u64 v) {
    auto pick = [](u64 p, u64 q) { return MIN ? (p < q ? p : q) : (p > q ? p : q); };
    v = row16_key<MIN>(v);
    double p, q;
    swap_pair<false>(__builtin_bit_cast(double, v), p, q); v = pick(__builtin_bit_cast(u64, p), __builtin_bit_cast(u64, q));
    swap_pair<true>(__builtin_bit_cast(double, v), p, q);  v = pick(__builtin_bit_cast(u64, p), __builtin_bit_cast(u64, q));
    return v;
}

// fitting.py:282-288 + :356-359.  First-occurrence arg indices (ties: lowest vertex index): the reductions
// run on keys (ordered value << 32 | index) for the minima and (ordered value << 32 | ~index) for the maxima.
// the reductions of one workgroup of 512 threads over all vertices of problem b; the result in *out (LDS or global), valid
// for the whole workgroup after its next barrier
// the tail of the reduction: a thread's keys -> wave keys -> the workgroup's box (512 threads)
__device__ __forceinline__ void sdf_box_finish(u64 (&kmin)[3], u64 (&kmax)[3], int tid, u64 (*s_k)[6], SdfBox* out);

// Round 6: the box from the vertex pass's own per-tile keys (vertex_pass.hip: vp_box_parts; [ntiles][6] per problem: min keys of
// x, y, z, max keys of x, y, z) instead of from all 6890 vertices - min / max of keys are associative, the keys carry the vertex
// index: the SAME box, arg indices included.  (The 16 workgroups of a problem each reduced the whole vertex list: ~5 us of the
// front kernel's 16.7.)
__device__ __forceinline__ void sdf_box_from_parts(const u64* __restrict__ parts, int ntiles, int b, int tid, u64 (*s_k)[6], SdfBox* out) {
    u64 kmin[3] = {~0ull, ~0ull, ~0ull}, kmax[3] = {0ull, 0ull, 0ull};
    for (int t = tid; t < ntiles; t += SDF_ADJ_NT) {
        const u64* q = parts + ((size_t)b * ntiles + t) * 6;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const u64 lo = q[a], hi = q[3 + a];
            kmin[a] = lo < kmin[a] ? lo : kmin[a];
            kmax[a] = hi > kmax[a] ? hi : kmax[a];
        }
    }
    sdf_box_finish(kmin, kmax, tid, s_k, out);
}

__device__ __forceinline__ void sdf_box_reduce(const float* __restrict__ verts, int nv, int b, int tid, u64 (*s_k)[6], SdfBox* out) {
    const float* vb = verts + (size_t)b * nv * 3;
    u64 kmin[3] = {~0ull, ~0ull, ~0ull}, kmax[3] = {0ull, 0ull, 0ull};
    for (int vbase = 0; vbase < nv; vbase += 512 * 16) {
        // 16 rows of 512 vertices with every load issued before the first compare (clamped index: a vertex
        // seen twice does not change a min / max, and its index is the same)
        float x[16][3];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int v = min(vbase + i * 512 + tid, nv - 1);
#pragma unroll
            for (int a = 0; a < 3; ++a) x[i][a] = vb[3 * v + a];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int v = min(vbase + i * 512 + tid, nv - 1);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const u64 o = (u64)ord_bits(x[i][a]) << 32;
                const u64 lo = o | (unsigned)v, hi = o | (unsigned)~v;
                kmin[a] = lo < kmin[a] ? lo : kmin[a];
                kmax[a] = hi > kmax[a] ? hi : kmax[a];
            }
        }
    }
    sdf_box_finish(kmin, kmax, tid, s_k, out);
}

__device__ __forceinline__ void sdf_box_finish(u64 (&kmin)[3], u64 (&kmax)[3], int tid, u64 (*s_k)[6], SdfBox* out) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) { kmin[a] = wave64_key<true>(kmin[a]); kmax[a] = wave64_key<false>(kmax[a]); }
    if (lane == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_k[wave][a] = kmin[a]; s_k[wave][3 + a] = kmax[a]; }
    __syncthreads();
    if (tid == 0) {
        SdfBox o;
        float ext[3];
        for (int a = 0; a < 3; ++a) {
            u64 lo = s_k[0][a], hi = s_k[0][3 + a];
            for (int w = 1; w < 8; ++w) { lo = s_k[w][a] < lo ? s_k[w][a] : lo; hi = s_k[w][3 + a] > hi ? s_k[w][3 + a] : hi; }
            const float l = ord_float((unsigned)(lo >> 32)), h = ord_float((unsigned)(hi >> 32));
            o.c[a] = (l + h) / 2.f;                                  // boxes.mean(dim=1)
            o.imin[a] = (int)(unsigned)lo; o.imax[a] = (int)~(unsigned)hi;
            ext[a] = h - l;
        }
        int am = 0;
        if (ext[1] > ext[am]) am = 1;
        if (ext[2] > ext[am]) am = 2;
        o.amax = am;
        o.s = (float)((1 + 0.2) * 0.5) * ext[am];                    // "(1+0.2) * 0.5 * (...)": Python double meets a float tensor
        o.pad = 0;
        *out = o;
    }
}

__global__ __launch_bounds__(512) void sdf_bbox_kernel(const float* __restrict__ verts, int nv, const int* __restrict__ gate,
                                                       SdfBox* __restrict__ box, int* __restrict__ cull_flag) {
    __shared__ u64 s_k[8][6];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (gate && !gate[b]) return;                       // the problem's current stage has no SDF term (uniform)
    if (cull_flag && tid == 0) cull_flag[b] = 0;        // "walk all faces" is decided anew by this round's count / scan kernels
    sdf_box_reduce(verts, nv, b, tid, s_k, box + b);
}

// One vertex against a face list staged in LDS (<= SDF_CH faces): the arithmetic of sdf_sample_kernel for that case -
// parity of every in-range corner over the list, minimum distance of the inside corners, trilinear value and gradient.
__device__ __forceinline__ float4 sdf_sample_staged(const SdfTri* tri, int nf, int G, const float (&loc)[3], bool live) {
    float fr[3];
    int i0[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float pix = ((loc[a] + 1.f) * (float)G - 1.f) / 2.f;
        const float fl = floorf(pix);
        i0[a] = (int)fl;
        fr[a] = pix - fl;
    }
    float vc[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a) { vc[a][0] = sdf_voxel_coord(i0[a], G); vc[a][1] = sdf_voxel_coord(i0[a] + 1, G); }
    float pv[8];
#pragma unroll
    for (int cn = 0; cn < 8; ++cn) {
        pv[cn] = 0.f;
        const int ix = i0[0] + (cn & 1), iy = i0[1] + ((cn >> 1) & 1), iz = i0[2] + (cn >> 2);
        if (!(live && ix >= 0 && ix < G && iy >= 0 && iy < G && iz >= 0 && iz < G)) continue;
        const float c[3] = {vc[0][cn & 1], vc[1][(cn >> 1) & 1], vc[2][cn >> 2]};
        int n = 0;
        for (int t = 0; t < nf; ++t) n += sdf_ray_hit(tri[t], c) ? 1 : 0;
        if (!(n & 1)) continue;
        float md = 1000.f;
        for (int t = 0; t < nf; ++t) { const float d = sdf_tri_distance(tri[t], c); if (d < md) md = d; }
        pv[cn] = md;
    }
    float val = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int cn = 0; cn < 8; ++cn) {
        const float p = pv[cn];
        const float wx = (cn & 1) ? fr[0] : 1.f - fr[0];
        const float wy = (cn & 2) ? fr[1] : 1.f - fr[1];
        const float wz = (cn & 4) ? fr[2] : 1.f - fr[2];
        val += p * wx * wy * wz;
        gx += ((cn & 1) ? p : -p) * wy * wz;
        gy += ((cn & 2) ? p : -p) * wx * wz;
        gz += ((cn & 4) ? p : -p) * wx * wy;
    }
    const float hg = (float)G / 2.f;
    return make_float4(val, gx * hg, gy * hg, gz * hg);
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
    // the 8 corners: bit q of corner index = +1 along axis q; their voxel-centre coordinates per axis
    float vc[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a) { vc[a][0] = sdf_voxel_coord(i0[a], G); vc[a][1] = sdf_voxel_coord(i0[a] + 1, G); }
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
#pragma unroll
            for (int cn = 0; cn < 8; ++cn) {
                if (!((inrange >> cn) & 1)) continue;
                const float c[3] = {vc[0][cn & 1], vc[1][(cn >> 1) & 1], vc[2][cn >> 2]};
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
                    const float c[3] = {vc[0][cn & 1], vc[1][(cn >> 1) & 1], vc[2][cn >> 2]};
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

// ---------------------------------------------------------------------------------------------------------------
// All faces (sdf_cuda_kernel.cu:258-287 walks every triangle for every voxel): exact culling.
//
// Both per-voxel quantities admit it.  Crossing parity: every test segment ends in the one point P = (-1,-1,-1), so the
// rays form a pencil through P; with q = x - P (all components positive: the mesh is normalised into |x| <= 1/1.2, the
// voxel centres lie above -1) the map x -> (q_x, q_y) / (q_x + q_y + q_z) sends a whole ray to ONE point of the unit
// square and a triangle to the triangle of its projected vertices.  Triangles are binned by the bounding box of that
// footprint - of the triangle thickened by SDF_CULL_DELTA in every direction, which covers by a wide margin the distance
// at which the floating-point test (sdf_ray_hit) can still call a near miss a hit - into SDF_NB x SDF_NB bins; a voxel
// tests the triangles of its bin whose nearest vertex (in q_x + q_y + q_z, which grows monotonically from P along a ray)
// is not beyond the voxel.  A triangle is in a bin at most once, so the count of hits - the parity - is the brute-force one.
// Minimum distance: triangles are binned by bounding box into SDF_NC^3 cells of [-1, 1]^3.  A corner the term samples
// belongs to the cell of a mesh vertex v, so |corner - v| bounds the minimum from above (when v is a vertex of a listed
// face); the triangles of the cells the box corner +- that radius overlaps contain every triangle closer than the radius,
// the minimum over them is the minimum over all - the same sdf_tri_distance on the same SdfTri, so the same bits.  A
// corner whose minimum comes out at the radius or above (v in no listed face) walks all triangles, as does a problem whose
// lists overflow the workspace.
// One counting sort per problem and round (count, scan, fill) builds both structures in one bin space.
// ---------------------------------------------------------------------------------------------------------------
constexpr int SDF_NB = 256;                         // projective bins per axis (ray structure)
constexpr int SDF_NC3 = 64;                         // cells per axis (distance structure)
constexpr int SDF_NBINS = SDF_NB * SDF_NB + SDF_NC3 * SDF_NC3 * SDF_NC3;
constexpr int SDF_OFFS_LD = SDF_NBINS + 4;           // row of the offsets (NBINS + 1 used), 16-byte aligned
constexpr int SDF_CULL_CAP = 64;                    // list entries per face the workspace holds
constexpr float SDF_CULL_DELTA = 2e-3f;
constexpr int SDF_CULL_MIN_FACES = 4 * SDF_CH;      // below: the staged brute-force kernel

struct SdfCullWs {               // per-problem slices of one allocation
    SdfTri* tri;                 // [B][F]       pad[0] = min over the vertices of q_x + q_y + q_z
    int* offs;                   // [B][SDF_OFFS_LD]
    int* cnt;                    // [B][NBINS]   zero between rounds (count adds, fill takes away)
    int2* ent;                   // [B][CAP * F]  {face, its pad[0] as bits}
    int* flag;                   // [B]          1 = lists do not fit / a vertex outside the normalised box: walk all faces
};
static size_t cull_align(size_t x) { return (x + 255) & ~(size_t)255; }
size_t sdf_cull_bytes(int B, int num_faces) {
    return cull_align((size_t)B * num_faces * sizeof(SdfTri)) + cull_align((size_t)B * SDF_OFFS_LD * 4) +
           cull_align((size_t)B * SDF_NBINS * 4) + cull_align((size_t)B * SDF_CULL_CAP * num_faces * 8) + cull_align((size_t)B * 4);
}
size_t sdf_cull_zero_offset(int B, int num_faces) {          // the part that must be zero before the first round: cnt
    return cull_align((size_t)B * num_faces * sizeof(SdfTri)) + cull_align((size_t)B * SDF_OFFS_LD * 4);
}
size_t sdf_cull_zero_bytes(int B) { return cull_align((size_t)B * SDF_NBINS * 4); }
int sdf_cull_min_faces() { return SDF_CULL_MIN_FACES; }
static SdfCullWs cull_views(void* ws, int B, int F) {
    unsigned char* p = reinterpret_cast<unsigned char*>(ws);
    SdfCullWs w;
    w.tri = reinterpret_cast<SdfTri*>(p); p += cull_align((size_t)B * F * sizeof(SdfTri));
    w.offs = reinterpret_cast<int*>(p);   p += cull_align((size_t)B * SDF_OFFS_LD * 4);
    w.cnt = reinterpret_cast<int*>(p);    p += cull_align((size_t)B * SDF_NBINS * 4);
    w.ent = reinterpret_cast<int2*>(p);   p += cull_align((size_t)B * SDF_CULL_CAP * F * 8);
    w.flag = reinterpret_cast<int*>(p);
    return w;
}

struct SdfBinBox { int a0, a1, b0, b1, c0[3], c1[3]; bool bad; float min_s; };
__device__ __forceinline__ int sdf_bin2(float a) { return min(SDF_NB - 1, max(0, (int)(a * (float)SDF_NB))); }
__device__ __forceinline__ int sdf_cell3(float x) { return min(SDF_NC3 - 1, max(0, (int)((x + 1.0f) * (0.5f * (float)SDF_NC3)))); }
// bins / cells of one triangle from its normalised vertices (the same code in the count and in the fill kernel)
__device__ __forceinline__ SdfBinBox sdf_tri_bins(const float (&p)[3][3]) {
    SdfBinBox o;
    float amin = 2.f, amax = -1.f, bmin = 2.f, bmax = -1.f, smin = 1e30f;
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    bool bad = false;
    const float dl = SDF_CULL_DELTA;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const float qx = p[m][0] + 1.0f, qy = p[m][1] + 1.0f, qz = p[m][2] + 1.0f;
        const float sq = qx + qy + qz;
        bad = bad || !(qx > 8.f * dl && qy > 8.f * dl && qz > 8.f * dl && sq < 16.f);          // also catches NaN
        amin = fminf(amin, (qx - dl) / (sq + dl)); amax = fmaxf(amax, (qx + dl) / (sq - dl));
        bmin = fminf(bmin, (qy - dl) / (sq + dl)); bmax = fmaxf(bmax, (qy + dl) / (sq - dl));
        smin = fminf(smin, sq);
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], p[m][a]); hi[a] = fmaxf(hi[a], p[m][a]); }
    }
    o.a0 = sdf_bin2(amin); o.a1 = sdf_bin2(amax); o.b0 = sdf_bin2(bmin); o.b1 = sdf_bin2(bmax);
#pragma unroll
    for (int a = 0; a < 3; ++a) { o.c0[a] = sdf_cell3(lo[a]); o.c1[a] = sdf_cell3(hi[a]); }
    o.bad = bad; o.min_s = smin;
    return o;
}
__device__ __forceinline__ void sdf_tri_points(const float* vb, const SdfBox& bx, const int32_t* faces, int f, float (&p)[3][3]) {
    for (int m = 0; m < 3; ++m) {
        const int vi = faces[3 * f + m];
        for (int a = 0; a < 3; ++a) p[m][a] = (vb[3 * vi + a] - bx.c[a]) / bx.s;     // fitting.py:362-363
    }
}

// thread per (problem, face): the face's record and its bin counts
__global__ __launch_bounds__(256) void sdf_cull_count_kernel(const float* __restrict__ verts, int nv, const SdfBox* __restrict__ box,
                                                             const int32_t* __restrict__ faces, int F, const int* __restrict__ gate,
                                                             SdfCullWs W) {
    const int b = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
    if (gate && !gate[b]) return;
    if (f >= F) return;
    const SdfBox bx = box[b];
    float p[3][3];
    sdf_tri_points(verts + (size_t)b * nv * 3, bx, faces, f, p);
    SdfTri T;
    sdf_tri_setup(T, p[0], p[1], p[2]);
    for (int q = 0; q < 3; ++q) { T.d12[q] = 0.f; T.pad[q] = 0.f; }
    T.m12 = 0.f;
    const SdfBinBox bb = sdf_tri_bins(p);
    T.pad[0] = bb.min_s;
    W.tri[(size_t)b * F + f] = T;
    if (bb.bad) { atomicOr(&W.flag[b], 1); return; }               // (the flag was cleared by this round's box kernel)
    int* cnt = W.cnt + (size_t)b * SDF_NBINS;
    for (int ib = bb.b0; ib <= bb.b1; ++ib)
        for (int ia = bb.a0; ia <= bb.a1; ++ia) atomicAdd(&cnt[ib * SDF_NB + ia], 1);
    for (int kz = bb.c0[2]; kz <= bb.c1[2]; ++kz)
        for (int ky = bb.c0[1]; ky <= bb.c1[1]; ++ky)
            for (int kx = bb.c0[0]; kx <= bb.c1[0]; ++kx) atomicAdd(&cnt[SDF_NB * SDF_NB + (kz * SDF_NC3 + ky) * SDF_NC3 + kx], 1);
}

// one workgroup per problem: exclusive scan of the counts; lists that do not fit raise the flag
constexpr int SDF_SCAN_NT = 1024;
constexpr int SDF_SCAN_PER = SDF_NBINS / SDF_SCAN_NT;
static_assert(SDF_NBINS % SDF_SCAN_NT == 0 && SDF_SCAN_PER % 4 == 0, "bins per scan thread");
__global__ __launch_bounds__(SDF_SCAN_NT) void sdf_cull_scan_kernel(int F, const int* __restrict__ gate, SdfCullWs W) {
    // (64-bit sums: a face may be counted in up to 256 * 256 + 64^3 bins, so >= 8192 large faces - user meshes handed to
    // mvfit_sdf - wrap 2^31; a wrapped total could pass the overflow check below and the fill kernel would write out of
    // bounds.  The offsets themselves stay 32-bit: they are only used when the total fits SDF_CULL_CAP * F entries.)
    __shared__ long long wsum[SDF_SCAN_NT / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (gate && !gate[b]) return;
    const int4* cnt = reinterpret_cast<const int4*>(W.cnt + (size_t)b * SDF_NBINS + (size_t)tid * SDF_SCAN_PER);
    int* offs = W.offs + (size_t)b * SDF_OFFS_LD + (size_t)tid * SDF_SCAN_PER;
    long long mine = 0;
#pragma unroll 8
    for (int k = 0; k < SDF_SCAN_PER / 4; ++k) { const int4 v = cnt[k]; mine += ((long long)v.x + v.y) + ((long long)v.z + v.w); }
    long long incl = mine;                               // inclusive scan across the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const long long o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    long long base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SDF_SCAN_NT / 64; ++w) { const long long s = wsum[w]; if (w < wave) base += s; total += s; }
    int run = (int)(base + incl - mine);
#pragma unroll 8
    for (int k = 0; k < SDF_SCAN_PER / 4; ++k) {
        const int4 v = cnt[k];
        int4 o;
        o.x = run; run += v.x; o.y = run; run += v.y; o.z = run; run += v.z; o.w = run; run += v.w;
        *reinterpret_cast<int4*>(offs + 4 * k) = o;
    }
    if (tid == 0) {
        const bool over = total > (long long)SDF_CULL_CAP * F;
        W.offs[(size_t)b * SDF_OFFS_LD + SDF_NBINS] = over ? 0 : (int)total;
        if (over) atomicOr(&W.flag[b], 2);
    }
}

// thread per (problem, face): the face's index into the lists of its bins (order within a list: arrival - the
// results taken from a list, a count and a minimum, do not depend on it)
__global__ __launch_bounds__(256) void sdf_cull_fill_kernel(const float* __restrict__ verts, int nv, const SdfBox* __restrict__ box,
                                                            const int32_t* __restrict__ faces, int F, const int* __restrict__ gate,
                                                            SdfCullWs W) {
    const int b = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
    if (gate && !gate[b]) return;
    if (f >= F) return;
    const SdfBox bx = box[b];
    float p[3][3];
    sdf_tri_points(verts + (size_t)b * nv * 3, bx, faces, f, p);
    const SdfBinBox bb = sdf_tri_bins(p);
    if (bb.bad) return;
    const bool keep = W.flag[b] == 0;
    int* cnt = W.cnt + (size_t)b * SDF_NBINS;
    const int* offs = W.offs + (size_t)b * SDF_OFFS_LD;
    int2* ent = W.ent + (size_t)b * SDF_CULL_CAP * F;
    const int2 rec = make_int2(f, __builtin_bit_cast(int, bb.min_s));
    for (int ib = bb.b0; ib <= bb.b1; ++ib)
        for (int ia = bb.a0; ia <= bb.a1; ++ia) {
            const int bin = ib * SDF_NB + ia;
            const int k = atomicSub(&cnt[bin], 1) - 1;
            if (keep) ent[offs[bin] + k] = rec;
        }
    for (int kz = bb.c0[2]; kz <= bb.c1[2]; ++kz)
        for (int ky = bb.c0[1]; ky <= bb.c1[1]; ++ky)
            for (int kx = bb.c0[0]; kx <= bb.c1[0]; ++kx) {
                const int bin = SDF_NB * SDF_NB + (kz * SDF_NC3 + ky) * SDF_NC3 + kx;
                const int k = atomicSub(&cnt[bin], 1) - 1;
                if (keep) ent[offs[bin] + k] = rec;
            }
}

#ifdef MVFIT_SDF_STATS
__device__ unsigned long long g_sdf_stats[8];      // corners in range, ray tests, inside corners, distance tests, corners that walked all faces, ray tests skipped by depth
#define SDF_STAT(k, v) atomicAdd(&g_sdf_stats[k], (unsigned long long)(v))
#else
#define SDF_STAT(k, v) do { } while (0)
#endif
// The sample kernel on the lists.  A vertex is served by 32 consecutive lanes (half a wave).  Parity: 4 lanes per corner
// walk the corner's bin, strided.  Distance: the 32 lanes take the vertex' inside corners one after the other and walk the
// cells around the corner, strided; a cell is left out when its box is farther than the best distance so far (or than the
// sampled vertex, which bounds the minimum), the corner's own cell goes first.  A count and a minimum do not depend on the
// order or on which lane met which face.
constexpr int SDF_VPB = SDF_NT / 32;                // vertices per workgroup
__device__ __forceinline__ float sdf_group_min32(float v) {
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) v = fminf(v, __shfl_xor(v, d));
    return v;
}
__global__ __launch_bounds__(SDF_NT) void sdf_sample_culled_kernel(const float* __restrict__ verts, int nv, const SdfBox* __restrict__ box,
                                                                   int F, int G, const int* __restrict__ gate, SdfCullWs W,
                                                                   float4* __restrict__ samp) {
    const int b = blockIdx.y, v = blockIdx.x * SDF_VPB + (threadIdx.x >> 5), l = threadIdx.x & 31, cn = l >> 2, sub = l & 3;
    if (gate && !gate[b]) return;
    const bool live = v < nv;
    const SdfBox bx = box[b];
    const float* vb = verts + (size_t)b * nv * 3;
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
    const SdfTri* tri = W.tri + (size_t)b * F;
    const int* offs = W.offs + (size_t)b * SDF_OFFS_LD;
    const int2* ent = W.ent + (size_t)b * SDF_CULL_CAP * F;
    const bool walk_all = W.flag[b] != 0;
    // A box with a non-finite centre or scale (a fit that ran off to infinity) makes one coordinate of EVERY normalised
    // face vertex NaN: each comparison of sdf_ray_hit then fails, no corner is inside, whatever the walk.  (And a finite box
    // means finite vertices inside it: the only way to the walk-all flag that is left is a list overflow.)
    const bool no_hits = !(fabsf(bx.c[0]) < INFINITY && fabsf(bx.c[1]) < INFINITY && fabsf(bx.c[2]) < INFINITY && fabsf(bx.s) < INFINITY) || bx.s == 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) { SDF_STAT(6, W.flag[b] & 1); SDF_STAT(4, (W.flag[b] >> 1) & 1); SDF_STAT(7, 1); }
    // ---- crossing parity of this lane's corner ----
    bool inside;
    {
        const int ix = i0[0] + (cn & 1), iy = i0[1] + ((cn >> 1) & 1), iz = i0[2] + (cn >> 2);
        const bool inrange = live && ix >= 0 && ix < G && iy >= 0 && iy < G && iz >= 0 && iz < G;
        const float c[3] = {sdf_voxel_coord(ix, G), sdf_voxel_coord(iy, G), sdf_voxel_coord(iz, G)};
        int n = 0;
        if (inrange && !no_hits) {
            if (!walk_all) {
                const float qx = c[0] + 1.0f, qy = c[1] + 1.0f, qz = c[2] + 1.0f, sq = qx + qy + qz;
                const int bin = sdf_bin2(qy / sq) * SDF_NB + sdf_bin2(qx / sq);
                const float s_lim = sq + 3.f * SDF_CULL_DELTA;
                const int e0 = offs[bin], e1 = offs[bin + 1];
                if (sub == 0) SDF_STAT(0, 1);
                for (int e = e0 + sub; e < e1; e += 4) {
                    const int2 en = ent[e];
                    if (__builtin_bit_cast(float, en.y) > s_lim) { SDF_STAT(5, 1); continue; }
                    SDF_STAT(1, 1);
                    n += sdf_ray_hit(tri[en.x], c) ? 1 : 0;
                }
            } else {
                for (int f = sub; f < F; f += 4) n += sdf_ray_hit(tri[f], c) ? 1 : 0;
            }
        }
        n += __shfl_xor(n, 1);
        n += __shfl_xor(n, 2);
        inside = inrange && (n & 1);
    }
    const unsigned long long bal = __ballot(inside && sub == 0);
    const unsigned half = (unsigned)(bal >> (threadIdx.x & 32));          // this vertex' 32 lanes: bit 4 k = corner k
    // ---- minimum distance of the inside corners, one corner at a time on the 32 lanes ----
    float pv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        pv[k] = 0.f;
        if (!((half >> (4 * k)) & 1u)) continue;
        const float c[3] = {sdf_voxel_coord(i0[0] + (k & 1), G), sdf_voxel_coord(i0[1] + ((k >> 1) & 1), G), sdf_voxel_coord(i0[2] + (k >> 2), G)};
        float md = 1000.f;
        bool done = false;
        if (!walk_all) {
            // the sampled vertex is at most this far away (in the normalised frame the records are in)
            const float rad = sdf_dist(c, loc) * 1.001f + 1e-6f;
            float bound = rad;
            int k0[3], k1[3], kc[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) { k0[a] = sdf_cell3(c[a] - rad); k1[a] = sdf_cell3(c[a] + rad); kc[a] = sdf_cell3(c[a]); }
            if (l == 0) SDF_STAT(2, 1);
            auto cell = [&](int kx, int ky, int kz) {
                const int bin = SDF_NB * SDF_NB + (kz * SDF_NC3 + ky) * SDF_NC3 + kx;
                const int e0 = offs[bin], e1 = offs[bin + 1];
                if (l == 0) SDF_STAT(3, e1 - e0);
                for (int e = e0 + l; e < e1; e += 32) { const float d = sdf_tri_distance(tri[ent[e].x], c); if (d < md) md = d; }
                md = sdf_group_min32(md);
                bound = fminf(bound, md);
            };
            cell(kc[0], kc[1], kc[2]);
            constexpr float hc = 2.0f / (float)SDF_NC3;
            for (int kz = k0[2]; kz <= k1[2]; ++kz)
                for (int ky = k0[1]; ky <= k1[1]; ++ky)
                    for (int kx = k0[0]; kx <= k1[0]; ++kx) {
                        if (kx == kc[0] && ky == kc[1] && kz == kc[2]) continue;
                        // distance from the corner to the cell's box (the outermost cells also hold what lies beyond them)
                        const int kk[3] = {kx, ky, kz};
                        float d2 = 0.f;
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            const float lo = -1.0f + (float)kk[a] * hc, hi = lo + hc;
                            float da = 0.f;
                            if (c[a] < lo && kk[a] > 0) da = lo - c[a];
                            if (c[a] > hi && kk[a] < SDF_NC3 - 1) da = c[a] - hi;
                            d2 += da * da;
                        }
                        const float lim = bound * 1.0001f + 2e-6f;
                        if (d2 > lim * lim) continue;
                        cell(kx, ky, kz);
                    }
            done = md <= rad * 0.9999f;        // everything not examined is farther than rad
        }
        if (!done) {
            if (l == 0) SDF_STAT(4, 1);
            md = 1000.f;
            for (int f = l; f < F; f += 32) { const float d = sdf_tri_distance(tri[f], c); if (d < md) md = d; }
            md = sdf_group_min32(md);
        }
        pv[k] = md;
    }
    // trilinear interpolation and its coordinate gradient, in the corner order of sdf_sample_kernel
    float val = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float p = pv[k];
        const float wx = (k & 1) ? fr[0] : 1.f - fr[0];
        const float wy = (k & 2) ? fr[1] : 1.f - fr[1];
        const float wz = (k & 4) ? fr[2] : 1.f - fr[2];
        val += p * wx * wy * wz;
        gx += ((k & 1) ? p : -p) * wy * wz;
        gy += ((k & 2) ? p : -p) * wx * wz;
        gz += ((k & 4) ? p : -p) * wx * wy;
    }
    const float hg = (float)G / 2.f;
    if (live && l == 0) samp[(size_t)b * nv + v] = make_float4(val, gx * hg, gy * hg, gz * hg);
}

// ---------------------------------------------------------------------------------------------------------------
// The stand-alone op (mvfit_sdf: every voxel of the G^3 grid, sdf_cuda_kernel.cu:242-335) on the same face lists.  The op's
// vertices arrive normalised by the caller: the lists are built with the identity box (x - 0) / 1 = x.  Parity as in the
// term; minimum distance of an inside voxel - which may lie deep inside the mesh - by rings of cells around the voxel's
// cell: ring k (Chebyshev distance k) is visited after ring k - 1, a cell farther than the best distance so far is left
// out, and the search ends when the best distance is at most k h: everything outside the block of rings 0..k is at least
// that far away.  A mesh that leaves the box the lists are made for (a vertex within 0.016 of -1 on any axis, or not
// finite) raises the element's flag and its voxels walk all faces.
// ---------------------------------------------------------------------------------------------------------------
__global__ void sdf_identity_box_kernel(SdfBox* __restrict__ box, int* __restrict__ flag, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    SdfBox o;
    for (int a = 0; a < 3; ++a) { o.c[a] = 0.f; o.imin[a] = 0; o.imax[a] = 0; }
    o.s = 1.f; o.amax = 0; o.pad = 0;
    box[b] = o;
    flag[b] = 0;
}

__global__ __launch_bounds__(SDF_NT) void sdf_voxelize_culled_kernel(int F, int G, SdfCullWs W, float* __restrict__ phi) {
    const int bn = blockIdx.y;
    const int nvox = G * G * G;
    const int vid = blockIdx.x * SDF_NT + threadIdx.x;
    if (vid >= nvox) return;
    const int i = vid % G, j = (vid / G) % G, k = (vid / (G * G)) % G;
    const float c[3] = {sdf_voxel_coord(i, G), sdf_voxel_coord(j, G), sdf_voxel_coord(k, G)};
    const SdfTri* tri = W.tri + (size_t)bn * F;
    const int* offs = W.offs + (size_t)bn * SDF_OFFS_LD;
    const int2* ent = W.ent + (size_t)bn * SDF_CULL_CAP * F;
    const bool walk_all = W.flag[bn] != 0;
    int n = 0;
    if (!walk_all) {
        const float qx = c[0] + 1.0f, qy = c[1] + 1.0f, qz = c[2] + 1.0f, sq = qx + qy + qz;
        const int bin = sdf_bin2(qy / sq) * SDF_NB + sdf_bin2(qx / sq);
        const float s_lim = sq + 3.f * SDF_CULL_DELTA;
        const int e0 = offs[bin], e1 = offs[bin + 1];
        for (int e = e0; e < e1; ++e) {
            const int2 en = ent[e];
            if (__builtin_bit_cast(float, en.y) > s_lim) continue;
            n += sdf_ray_hit(tri[en.x], c) ? 1 : 0;
        }
    } else {
        for (int f = 0; f < F; ++f) n += sdf_ray_hit(tri[f], c) ? 1 : 0;
    }
    float md = 1000.f;
    if (n & 1) {
        if (!walk_all) {
            constexpr float hc = 2.0f / (float)SDF_NC3;
            const int kc[3] = {sdf_cell3(c[0]), sdf_cell3(c[1]), sdf_cell3(c[2])};
            auto visit = [&](int kx, int ky, int kz) {
                const int kk[3] = {kx, ky, kz};
                float d2 = 0.f;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float lo = -1.0f + (float)kk[a] * hc, hi = lo + hc;
                    float da = 0.f;
                    if (c[a] < lo && kk[a] > 0) da = lo - c[a];
                    if (c[a] > hi && kk[a] < SDF_NC3 - 1) da = c[a] - hi;
                    d2 += da * da;
                }
                const float lim = md * 1.0001f + 2e-6f;
                if (d2 > lim * lim) return;
                const int bin = SDF_NB * SDF_NB + (kz * SDF_NC3 + ky) * SDF_NC3 + kx;
                const int e0 = offs[bin], e1 = offs[bin + 1];
                for (int e = e0; e < e1; ++e) { const float d = sdf_tri_distance(tri[ent[e].x], c); if (d < md) md = d; }
            };
            for (int ring = 0; ring < SDF_NC3; ++ring) {
                const int x0 = max(kc[0] - ring, 0), x1 = min(kc[0] + ring, SDF_NC3 - 1);
                const int y0 = max(kc[1] - ring, 0), y1 = min(kc[1] + ring, SDF_NC3 - 1);
                const int z0 = max(kc[2] - ring, 0), z1 = min(kc[2] + ring, SDF_NC3 - 1);
                for (int kz = z0; kz <= z1; ++kz)
                    for (int ky = y0; ky <= y1; ++ky) {
                        const bool shell = (kz - kc[2] == ring) || (kc[2] - kz == ring) || (ky - kc[1] == ring) || (kc[1] - ky == ring);
                        if (shell) {
                            for (int kx = x0; kx <= x1; ++kx) visit(kx, ky, kz);
                        } else {
                            if (kc[0] - ring >= 0) visit(kc[0] - ring, ky, kz);
                            if (ring > 0 && kc[0] + ring <= SDF_NC3 - 1) visit(kc[0] + ring, ky, kz);
                        }
                    }
                if (md <= (float)ring * hc * 0.9999f) break;          // nothing outside the visited block is closer
                if (x0 == 0 && y0 == 0 && z0 == 0 && x1 == SDF_NC3 - 1 && y1 == SDF_NC3 - 1 && z1 == SDF_NC3 - 1) break;
            }
        } else {
            for (int f = 0; f < F; ++f) { const float d = sdf_tri_distance(tri[f], c); if (d < md) md = d; }
        }
    }
    phi[(size_t)bn * nvox + vid] = (n & 1) ? md : 0.f;
}

size_t sdf_op_ws_bytes(int B, int num_faces) { return sdf_cull_bytes(B, num_faces) + cull_align((size_t)B * sizeof(SdfBox)); }
bool sdf_op_uses_lists(int num_faces) { return num_faces >= SDF_CULL_MIN_FACES; }
// ws: sdf_op_ws_bytes(B, num_faces) bytes whose count area (sdf_cull_zero_offset / sdf_cull_zero_bytes) was zeroed once
hipError_t launch_sdf_voxelize_culled(const int32_t* faces, int num_faces, const float* vertices, int B, int num_vertices, int G,
                                      float* phi, void* ws, hipStream_t stream) {
    const SdfCullWs W = cull_views(ws, B, num_faces);
    SdfBox* box = reinterpret_cast<SdfBox*>(reinterpret_cast<unsigned char*>(ws) + sdf_cull_bytes(B, num_faces));
    hipLaunchKernelGGL(sdf_identity_box_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, box, W.flag, B);
    const dim3 gf((num_faces + 255) / 256, B);
    hipLaunchKernelGGL(sdf_cull_count_kernel, gf, dim3(256), 0, stream, vertices, num_vertices, (const SdfBox*)box, faces, num_faces,
                       (const int*)nullptr, W);
    hipLaunchKernelGGL(sdf_cull_scan_kernel, dim3(B), dim3(SDF_SCAN_NT), 0, stream, num_faces, (const int*)nullptr, W);
    hipLaunchKernelGGL(sdf_cull_fill_kernel, gf, dim3(256), 0, stream, vertices, num_vertices, (const SdfBox*)box, faces, num_faces,
                       (const int*)nullptr, W);
    const int nvox = G * G * G;
    hipLaunchKernelGGL(sdf_voxelize_culled_kernel, dim3((nvox + SDF_NT - 1) / SDF_NT, B), dim3(SDF_NT), 0, stream, num_faces, G, W, phi);
    return hipGetLastError();
}

// entry list of one problem: the vertices that carry gradient, in ascending vertex order
struct SdfEntry { int v; float g[3]; };                                  // vertex, dS/dvertex
static_assert(sizeof(SdfEntry) == 16, "entry layout");

constexpr int SDF_NC = 16;           // vertex chunks per problem in the entry kernel (one workgroup each)
constexpr int SDF_NIT = 1;           // 64-vertex rows per wave of a chunk: nv <= SDF_NC * 8 waves * SDF_NIT * 64 = 8192

// per (problem, vertex chunk): partial sums of S and of the box adjoint, entries written (at the chunk's own offset)
struct SdfChunk { double S, gc0, gc1, gc2, gs; int cnt, pad; };
static_assert(sizeof(SdfChunk) == 48, "chunk record");
#ifndef SDF_NS_
#define SDF_NS_ 8                   // (-DSDF_NS_=1 reproduces the single-chain summation order bit for bit: the control experiment)
#endif
constexpr int SDF_NS = SDF_NS_;      // workgroups (entry slices) per problem in the pull-back

// Kernel 1 of the adjoint, grid (SDF_NC, B): a workgroup scans one sixteenth (1 / SDF_NC) of a problem's vertices (the whole list
// through one CU took 9 us): partial S and box-adjoint sums in float64, and the chunk's entries - the vertices that carry
// gradient plus the box's arg-min / arg-max vertices - compacted in ascending order at the chunk's offset of the entry
// buffer, with the gradient of the sampling only (the box adjoint needs the sums of ALL chunks: sdf_pullback_kernel
// adds it when it loads an entry).
// FUSED (face lists of at most SDF_CH faces, i.e. the term as the reference wires it): the same workgroups first reduce
// the problem's box themselves - every chunk the whole vertex list, the same code, hence the same box in all SDF_NC chunks
// (SDF_NC redundant reductions over nv: 16 x 83 KB of L2-resident reads per problem, cheaper than a launch boundary) - and
// sample their own vertices against the list staged in LDS: box, sample and entry kernels in one launch (three launches
// of 6 + 10 + 5 us, mostly latency, became one).  The samples are still written for mvfit_sdf_term_read.
template <bool FUSED>
__global__ __launch_bounds__(SDF_ADJ_NT) void sdf_entries_kernel(int nv, const float* __restrict__ verts,
                                                                 SdfBox* __restrict__ box, float4* __restrict__ samp,
                                                                 const int* __restrict__ gate,
                                                                 SdfEntry* __restrict__ entries, SdfChunk* __restrict__ chunks,
                                                                 const int32_t* __restrict__ faces, int num_faces, int G,
                                                                 const u64* __restrict__ box_parts, int ntiles) {
    __shared__ double sh_d[8][5];
    __shared__ int sh_cnt[8];
    __shared__ u64 s_k[FUSED ? 8 : 1][6];
    __shared__ SdfBox sh_box;
    __shared__ SdfTri tri[FUSED ? SDF_CH : 1];
    const int b = blockIdx.y, y = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (gate && !gate[b]) return;
    const float* vb = verts + (size_t)b * nv * 3;
    float fraw[3][3];                                               // FUSED: the raw vertices of this thread's face (num_faces <= SDF_CH < threads)
    if constexpr (FUSED) {
        // requested before the box is reduced: the two dependent round trips (indices, then vertices) of the list's setup
        // then run under the reduction instead of behind it
        if (tid < num_faces)
            for (int m = 0; m < 3; ++m) {
                const int vi = faces[3 * tid + m];
                for (int a = 0; a < 3; ++a) fraw[m][a] = vb[3 * vi + a];
            }
        if (box_parts) sdf_box_from_parts(box_parts, ntiles, b, tid, s_k, &sh_box);      // (uniform) the pass's own tile keys
        else sdf_box_reduce(verts, nv, b, tid, s_k, &sh_box);
        __syncthreads();
    }
    const SdfBox bx = FUSED ? sh_box : box[b];
    if constexpr (FUSED) {
        if (y == 0 && tid == 0) box[b] = bx;
        if (tid < num_faces) {
            float p[3][3];
            for (int m = 0; m < 3; ++m)
                for (int a = 0; a < 3; ++a) p[m][a] = (fraw[m][a] - bx.c[a]) / bx.s;         // fitting.py:362-363
            sdf_tri_setup(tri[tid], p[0], p[1], p[2]);
        }
        __syncthreads();
    }
    float4* sb = samp + (size_t)b * nv;
    const int csz = (nv + SDF_NC - 1) / SDF_NC, k0 = y * csz, k1 = min(nv, k0 + csz);
    SdfEntry* eb = entries + (size_t)b * nv + k0;
    // ---- pass A: every wave owns an ascending run of the chunk, 64 vertices per row; everything stays in registers ----
    const int wsz = (csz + 7) / 8, c0 = k0 + wave * wsz, c1 = min(k1, c0 + wsz);
    float4 q[SDF_NIT];
    double S = 0.0, gc0 = 0.0, gc1 = 0.0, gc2 = 0.0, gs = 0.0;
    unsigned actmask = 0;
    int cnt = 0;
    {
        float px[SDF_NIT], py[SDF_NIT], pz[SDF_NIT];
#pragma unroll
        for (int i = 0; i < SDF_NIT; ++i) {
            const int v = c0 + i * 64 + lane;
            const bool in = v < c1;
            const int vc = in ? v : min(k0, nv - 1);      // padding lanes re-read a vertex that exists (small / odd nv: k0 may be >= nv)
            px[i] = vb[3 * vc]; py[i] = vb[3 * vc + 1]; pz[i] = vb[3 * vc + 2];
            if constexpr (FUSED) {
                const float loc[3] = {(px[i] - bx.c[0]) / bx.s, (py[i] - bx.c[1]) / bx.s, (pz[i] - bx.c[2]) / bx.s};
                q[i] = sdf_sample_staged(tri, num_faces, G, loc, true);
                if (in) sb[v] = q[i];
            } else {
                q[i] = sb[vc];
            }
            if (!in) q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < SDF_NIT; ++i) {
            const int v = c0 + i * 64 + lane;
            const bool in = v < c1;
            const float4 qq = q[i];
            S += (double)qq.x;
            bool act = (qq.y != 0.f) | (qq.z != 0.f) | (qq.w != 0.f);
            if (act) {                                  // rare: only vertices next to a non-zero voxel
                const float l0 = (px[i] - bx.c[0]) / bx.s, l1 = (py[i] - bx.c[1]) / bx.s, l2 = (pz[i] - bx.c[2]) / bx.s;
                gc0 -= (double)(qq.y / bx.s); gc1 -= (double)(qq.z / bx.s); gc2 -= (double)(qq.w / bx.s);
                gs -= (double)(qq.y * l0 / bx.s); gs -= (double)(qq.z * l1 / bx.s); gs -= (double)(qq.w * l2 / bx.s);
            }
            act |= (v == bx.imin[0]) | (v == bx.imin[1]) | (v == bx.imin[2]) | (v == bx.imax[0]) | (v == bx.imax[1]) | (v == bx.imax[2]);
            act &= in;
            if (act) actmask |= 1u << i;
            cnt += __popcll(__ballot(act));
        }
    }
    S = wave64_sum(S);
    if (__any(gc0 != 0.0 || gc1 != 0.0 || gc2 != 0.0 || gs != 0.0)) {      // (rare: only vertices next to a non-zero voxel carry these; sums of zeros are zeros)
        gc0 = wave64_sum(gc0); gc1 = wave64_sum(gc1); gc2 = wave64_sum(gc2); gs = wave64_sum(gs);
    }
    if (lane == 0) { sh_d[wave][0] = S; sh_d[wave][1] = gc0; sh_d[wave][2] = gc1; sh_d[wave][3] = gc2; sh_d[wave][4] = gs; sh_cnt[wave] = cnt; }
    __syncthreads();
    S = 0.0; gc0 = 0.0; gc1 = 0.0; gc2 = 0.0; gs = 0.0;
    int base = 0, n = 0;
    for (int w = 0; w < 8; ++w) {
        S += sh_d[w][0]; gc0 += sh_d[w][1]; gc1 += sh_d[w][2]; gc2 += sh_d[w][3]; gs += sh_d[w][4];
        if (w < wave) base += sh_cnt[w];
        n += sh_cnt[w];
    }
    // ---- pass B: the chunk's entries (ballot compaction, ascending order) ----
#pragma unroll
    for (int i = 0; i < SDF_NIT; ++i) {
        const int v = c0 + i * 64 + lane;
        const bool act = (actmask >> i) & 1u;
        const unsigned long long bal = __ballot(act);
        if (act) {
            const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
            *reinterpret_cast<float4*>(&eb[pos]) = make_float4(__builtin_bit_cast(float, v), q[i].y / bx.s, q[i].z / bx.s, q[i].w / bx.s);
        }
        base += __popcll(bal);
    }
    if (tid == 0) {
        SdfChunk c;
        c.S = S; c.gc0 = gc0; c.gc1 = gc1; c.gc2 = gc2; c.gs = gs; c.cnt = n; c.pad = 0;
        chunks[(size_t)b * SDF_NC + y] = c;
    }
}

// Kernel 2: pull-back of the entries through skinning and the blendshape basis (~5.4 KB of basis rows and weights per
// entry: one CU ingests ~10 B/clk, so a problem's list is cut into SDF_NS contiguous slices, one workgroup each; the
// slice partials are added in slice order by kernel 3 - deterministic).  Grid (SDF_NS, B).
__global__ __launch_bounds__(SDF_ADJ_NT) void sdf_pullback_kernel(DevModel M, DevPose P, const int* __restrict__ gate,
                                                                  const SdfBox* __restrict__ box,
                                                                  const SdfEntry* __restrict__ entries,
                                                                  const SdfChunk* __restrict__ chunks, SdfAdj* __restrict__ part,
                                                                  int* __restrict__ tickets, SdfAdj* __restrict__ out,
                                                                  unsigned* __restrict__ answer_tag, unsigned answer) {
    __shared__ int sh_pref[SDF_NC + 1];
    __shared__ float sh_box[4];                                      // box adjoint: d S / d centre (3), d S / d scale
    __shared__ __attribute__((aligned(16))) float sh_coef[KROWS];
    __shared__ float sh_A[NJ * 12];
    __shared__ __attribute__((aligned(16))) SdfEntry sh_e[SDF_EB];
    __shared__ float sh_T[SDF_EB][12];
    __shared__ float sh_vp[SDF_EB][3];
    __shared__ float sh_part[SDF_EB][3][4];
    __shared__ float sh_gvp[SDF_EB][3];
    const int b = blockIdx.y, y = blockIdx.x, tid = threadIdx.x;
    if (gate && !gate[b]) return;
    if (tid == 0) {                                                  // chunk records -> prefix of the entry counts, box adjoint
        double g0 = 0.0, g1 = 0.0, g2 = 0.0, gs = 0.0;
        int acc = 0;
        for (int c = 0; c < SDF_NC; ++c) {
            const SdfChunk ch = chunks[(size_t)b * SDF_NC + c];
            sh_pref[c] = acc; acc += ch.cnt;
            g0 += ch.gc0; g1 += ch.gc1; g2 += ch.gc2; gs += ch.gs;
        }
        sh_pref[SDF_NC] = acc;
        sh_box[0] = (float)g0; sh_box[1] = (float)g1; sh_box[2] = (float)g2;
        sh_box[3] = (float)gs * (float)((1 + 0.2) * 0.5);
    }
    __syncthreads();
    const SdfBox bx = box[b];
    const int ntot = sh_pref[SDF_NC];
    // a very short list (little more than the six vertices of the box) goes through slice 0 alone, which writes the result
    // itself - no partials, no ticket.  (Not for the term as wired: the one triangle shadows ~200 vertices, 1 MB of basis
    // rows per problem - through a single CU the kernel took 25 us instead of 11.)
    const bool small = ntot <= 16;
    if (small && y != 0) return;                                     // uniform
    const int lo = small ? 0 : (int)((long long)ntot * y / SDF_NS), n = small ? ntot : (int)((long long)ntot * (y + 1) / SDF_NS);
    const int csz = (M.nv + SDF_NC - 1) / SDF_NC;
    const SdfEntry* eb = entries + (size_t)b * M.nv;
    if (lo < n) {                                                    // (uniform) an empty slice writes zeros
        if (tid < KROWS) sh_coef[tid] = P.coefT[(size_t)(b >> 5) * KROWS * 32 + (size_t)tid * 32 + (b & 31)];
        if (tid >= 224 && tid < 224 + NJ * 12) sh_A[tid - 224] = P.Amat[(size_t)b * NJ * 12 + (tid - 224)];
    }
    float acc = 0.f;
    const int oj = tid >= KROWS ? (tid - KROWS) / 12 : 0, oe = tid >= KROWS ? (tid - KROWS) % 12 : 0;
    float gtv = 0.f;
    for (int e0 = lo; e0 < n; e0 += SDF_EB) {
        const int ne = min(SDF_EB, n - e0);
        __syncthreads();
        if (tid < ne) {
            // entry e0 + tid of the problem = entry (e0 + tid - prefix[c]) of chunk c; the box adjoint reaches the vertices
            // that define the box: half of d S / d centre each, -+ d S / d scale on the largest axis (fitting.py:282-288,356-359)
            const int idx = e0 + tid;
            int c = 0;
#pragma unroll
            for (int k = 1; k < SDF_NC; ++k) c += idx >= sh_pref[k] ? 1 : 0;
            const float4 raw = reinterpret_cast<const float4*>(eb + (size_t)c * csz)[idx - sh_pref[c]];
            const int v = __builtin_bit_cast(int, raw.x);
            float g[3] = {raw.y, raw.z, raw.w};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (v == bx.imin[a]) { g[a] += sh_box[a] / 2.f; if (a == bx.amax) g[a] -= sh_box[3]; }
                if (v == bx.imax[a]) { g[a] += sh_box[a] / 2.f; if (a == bx.amax) g[a] += sh_box[3]; }
            }
            reinterpret_cast<float4*>(sh_e)[tid] = make_float4(raw.x, g[0], g[1], g[2]);
        }
        __syncthreads();
        // v_posed = v_template + coef . basis  (lbs.py:179,203): thread per (entry, coordinate, quarter row),
        // 14 independent 16-byte loads in flight per thread
#pragma unroll 1
        for (int it = tid; it < ne * 12; it += SDF_ADJ_NT) {
            const int e = it / 12, r6 = it - 12 * e, k = r6 >> 2, h = r6 & 3;
            const int v = sh_e[e].v;
            const float4* row = reinterpret_cast<const float4*>(M.bs_vm + ((size_t)v * 3 + k) * KROWS) + h * (KROWS / 16);
            float4 r[KROWS / 16];
#pragma unroll
            for (int p4 = 0; p4 < KROWS / 16; ++p4) r[p4] = row[p4];
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int p4 = 0; p4 < KROWS / 16; ++p4) {
                const float4 cf = reinterpret_cast<const float4*>(sh_coef)[h * (KROWS / 16) + p4];
                a0 += cf.x * r[p4].x; a1 += cf.y * r[p4].y; a2 += cf.z * r[p4].z; a3 += cf.w * r[p4].w;
            }
            sh_part[e][k][h] = (a0 + a1) + (a2 + a3);
        }
        // blended transform rows (lbs.py:209-213): thread per (entry, element)
#pragma unroll 1
        for (int it = tid; it < ne * 12; it += SDF_ADJ_NT) {
            const int e = it / 12, l = it - 12 * e;
            const float* wr = M.w_vm + (size_t)sh_e[e].v * NJ;
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) t += wr[j] * sh_A[j * 12 + l];
            sh_T[e][l] = t;
        }
        __syncthreads();
        for (int it = tid; it < ne * 3; it += SDF_ADJ_NT) {                  // g_vposed = Tr^T g
            const int e = it / 3, k = it - 3 * e;
            sh_vp[e][k] = M.vt_planes[(size_t)k * M.nv_pad + sh_e[e].v] + ((sh_part[e][k][0] + sh_part[e][k][1]) + (sh_part[e][k][2] + sh_part[e][k][3]));
            sh_gvp[e][k] = sh_T[e][k] * sh_e[e].g[0] + sh_T[e][4 + k] * sh_e[e].g[1] + sh_T[e][8 + k] * sh_e[e].g[2];
        }
        __syncthreads();
        // thread per output, entries in ascending order: g_coef[224] | g_A[24][12]
        if (tid < KROWS) {
#pragma unroll 16
            for (int e = 0; e < ne; ++e) {
                const float* bsv = M.bs_vm + (size_t)sh_e[e].v * 3 * KROWS + tid;
                acc += bsv[0] * sh_gvp[e][0] + bsv[KROWS] * sh_gvp[e][1] + bsv[2 * KROWS] * sh_gvp[e][2];
            }
            if (tid < 3) for (int e = 0; e < ne; ++e) gtv += sh_e[e].g[tid];
        } else {
#pragma unroll 8
            for (int e = 0; e < ne; ++e) {
                const float w = M.w_vm[(size_t)sh_e[e].v * NJ + oj];
                acc += w * (oe < 9 ? sh_e[e].g[oe / 3] * sh_vp[e][oe % 3] : sh_e[e].g[oe - 9]);
            }
        }
    }
    // Slice partial out with write-through stores, then the slice takes a ticket of its problem: the slice that arrives
    // LAST adds the eight partials in slice order (deterministic whoever it is) - the former third kernel, without its
    // launch.  Hand-off as the guide prescribes for other-CU data: agent-scope (sc1) stores, this wave's stores drained
    // before the workgroup barrier in front of the ticket, agent-scope loads on the reading side.
    // The result goes out with agent-scope (write-through) stores: in the service rounds of the single-launch fit its reader is
    // an optimiser workgroup that is RUNNING, on any XCD (closure_device.h: sdf_ld); behind them - drained, workgroup barrier -
    // the answer tag of the round (null in the chained rounds, where a launch boundary is the hand-off).
    auto publish_answer = [&]() {
        if (!answer_tag) return;                                     // uniform
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(answer_tag + b, answer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (small) {
        SdfAdj& R = out[b];
        __hip_atomic_store(tid < KROWS ? &R.gcoef[tid] : &R.gA[tid - KROWS], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 3) __hip_atomic_store(&R.gtau[tid], gtv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            double S = 0.0;
            for (int c = 0; c < SDF_NC; ++c) S += chunks[(size_t)b * SDF_NC + c].S;
            __hip_atomic_store(&R.S, (float)S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        publish_answer();
        return;
    }
    SdfAdj& O = part[(size_t)b * SDF_NS + y];
    __hip_atomic_store(tid < KROWS ? &O.gcoef[tid] : &O.gA[tid - KROWS], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 3) __hip_atomic_store(&O.gtau[tid], gtv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __shared__ int sh_last;
    __syncthreads();
    if (tid == 0) sh_last = __hip_atomic_fetch_add(tickets + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == SDF_NS - 1;
    __syncthreads();
    if (!sh_last) return;                                            // uniform
    const SdfAdj* p = part + (size_t)b * SDF_NS;
    float racc = 0.f, rgt = 0.f;
#pragma unroll
    for (int yy = 0; yy < SDF_NS; ++yy) {
        racc += __hip_atomic_load(tid < KROWS ? &p[yy].gcoef[tid] : &p[yy].gA[tid - KROWS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 3) rgt += __hip_atomic_load(&p[yy].gtau[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    SdfAdj& R = out[b];
    __hip_atomic_store(tid < KROWS ? &R.gcoef[tid] : &R.gA[tid - KROWS], racc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 3) __hip_atomic_store(&R.gtau[tid], rgt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
        double S = 0.0;
        for (int c = 0; c < SDF_NC; ++c) S += chunks[(size_t)b * SDF_NC + c].S;
        __hip_atomic_store(&R.S, (float)S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(tickets + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next round (a launch boundary away)
    }
    publish_answer();
}

// work area behind `entries`: [B][nv] SdfEntry | [B][SDF_NS] SdfAdj slice partials | [B][SDF_NC] SdfChunk | [B] tickets
// (the tickets must be zero before the first round: mvfit_api.hip clears them when it allocates the area; every round
// leaves them at zero)
static size_t sdf_part_offset(int B, int nv) { return ((size_t)B * nv * sizeof(SdfEntry) + 255) & ~(size_t)255; }
static size_t sdf_chunk_offset(int B, int nv) { return sdf_part_offset(B, nv) + (size_t)B * SDF_NS * sizeof(SdfAdj); }
size_t sdf_ticket_offset(int B, int nv) { return (sdf_chunk_offset(B, nv) + (size_t)B * SDF_NC * sizeof(SdfChunk) + 255) & ~(size_t)255; }
size_t sdf_work_bytes(int B, int nv) { return sdf_ticket_offset(B, nv) + (size_t)B * sizeof(int); }

hipError_t launch_sdf_term(const DevModel& M, const DevPose& P, const float* verts, int B, const int32_t* faces, int num_faces,
                           int G, const int* gate, SdfBox* box, float4* samp, void* entries, SdfAdj* adj, hipStream_t stream,
                           void* cull, unsigned* answer_tag, unsigned answer, const unsigned long long* box_parts) {
    static_assert(SDF_ADJ_NT == KROWS + NJ * 12, "thread per output of the pull-back");
    unsigned char* wk = reinterpret_cast<unsigned char*>(entries);
    SdfAdj* part = reinterpret_cast<SdfAdj*>(wk + sdf_part_offset(B, M.nv));
    SdfChunk* chunks = reinterpret_cast<SdfChunk*>(wk + sdf_chunk_offset(B, M.nv));
    int* tickets = reinterpret_cast<int*>(wk + sdf_ticket_offset(B, M.nv));
    if (M.nv > SDF_NC * 8 * SDF_NIT * 64) return hipErrorInvalidValue;
    const bool culled = cull && num_faces >= SDF_CULL_MIN_FACES;
    if (num_faces <= SDF_CH) {
        hipLaunchKernelGGL(sdf_entries_kernel<true>, dim3(SDF_NC, B), dim3(SDF_ADJ_NT), 0, stream, M.nv, verts, box, samp, gate,
                           reinterpret_cast<SdfEntry*>(entries), chunks, faces, num_faces, G, box_parts, M.ntiles);
    } else {
        hipLaunchKernelGGL(sdf_bbox_kernel, dim3(B), dim3(512), 0, stream, verts, M.nv, gate, box,
                           culled ? cull_views(cull, B, num_faces).flag : (int*)nullptr);
        if (culled) {
            const SdfCullWs W = cull_views(cull, B, num_faces);
            const dim3 gf((num_faces + 255) / 256, B);
            hipLaunchKernelGGL(sdf_cull_count_kernel, gf, dim3(256), 0, stream, verts, M.nv, (const SdfBox*)box, faces, num_faces, gate, W);
            hipLaunchKernelGGL(sdf_cull_scan_kernel, dim3(B), dim3(SDF_SCAN_NT), 0, stream, num_faces, gate, W);
            hipLaunchKernelGGL(sdf_cull_fill_kernel, gf, dim3(256), 0, stream, verts, M.nv, (const SdfBox*)box, faces, num_faces, gate, W);
            hipLaunchKernelGGL(sdf_sample_culled_kernel, dim3((M.nv + SDF_VPB - 1) / SDF_VPB, B), dim3(SDF_NT), 0, stream, verts,
                               M.nv, (const SdfBox*)box, num_faces, G, gate, W, samp);
        } else {
            hipLaunchKernelGGL(sdf_sample_kernel, dim3((M.nv + SDF_NT - 1) / SDF_NT, B), dim3(SDF_NT), 0, stream, verts, M.nv,
                               (const SdfBox*)box, faces, num_faces, G, gate, samp);
        }
        hipLaunchKernelGGL(sdf_entries_kernel<false>, dim3(SDF_NC, B), dim3(SDF_ADJ_NT), 0, stream, M.nv, verts, box, samp, gate,
                           reinterpret_cast<SdfEntry*>(entries), chunks, faces, num_faces, G, (const u64*)nullptr, 0);
    }
    hipLaunchKernelGGL(sdf_pullback_kernel, dim3(SDF_NS, B), dim3(SDF_ADJ_NT), 0, stream, M, P, gate, (const SdfBox*)box,
                       reinterpret_cast<const SdfEntry*>(entries), (const SdfChunk*)chunks, part, tickets, adj, answer_tag, answer);
    return hipGetLastError();
}

}  // namespace mvfit

#ifdef MVFIT_SDF_STATS
extern "C" __attribute__((visibility("default"))) int mvfit_debug_sdf_stats(unsigned long long* out, int reset) {
    hipDeviceSynchronize();
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(mvfit::g_sdf_stats), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = {}; hipMemcpyToSymbol(HIP_SYMBOL(mvfit::g_sdf_stats), z, sizeof(z)); }
    return 0;
}
#endif

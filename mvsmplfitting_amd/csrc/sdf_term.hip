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

// order-preserving map float -> uint32 (total order of the finite values)
__device__ __forceinline__ unsigned ord_bits(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_float(unsigned o) {
    const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __builtin_bit_cast(float, u);
}
typedef unsigned long long u64;
template <int CTRL>
__device__ __forceinline__ u64 dpp_u64(u64 v) { return __builtin_bit_cast(u64, dpp_mov<CTRL>(__builtin_bit_cast(double, v))); }
// whole-wave min / max of 64-bit keys on the DPP path (wave_ops.h), same result in every lane
template <bool MIN>
__device__ __forceinline__ u64 wave64_key(u64 v) {
    auto pick = [](u64 p, u64 q) { return MIN ? (p < q ? p : q) : (p > q ? p : q); };
    v = pick(v, dpp_u64<DPP_XOR1>(v));
    v = pick(v, dpp_u64<DPP_XOR2>(v));
    v = pick(v, dpp_u64<DPP_HALF_MIRROR>(v));
    v = pick(v, dpp_u64<DPP_MIRROR>(v));
    double p, q;
    swap_pair<false>(__builtin_bit_cast(double, v), p, q); v = pick(__builtin_bit_cast(u64, p), __builtin_bit_cast(u64, q));
    swap_pair<true>(__builtin_bit_cast(double, v), p, q);  v = pick(__builtin_bit_cast(u64, p), __builtin_bit_cast(u64, q));
    return v;
}

// fitting.py:282-288 + :356-359.  First-occurrence arg indices (ties: lowest vertex index): the reductions
// run on keys (ordered value << 32 | index) for the minima and (ordered value << 32 | ~index) for the maxima.
__global__ __launch_bounds__(512) void sdf_bbox_kernel(const float* __restrict__ verts, int nv, const int* __restrict__ gate,
                                                       SdfBox* __restrict__ box) {
    __shared__ u64 s_k[8][6];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (gate && !gate[b]) return;                       // the problem's current stage has no SDF term (uniform)
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

// entry list of one problem: the vertices that carry gradient, in ascending vertex order
struct SdfEntry { int v; float g[3]; };                                  // vertex, dS/dvertex
static_assert(sizeof(SdfEntry) == 16, "entry layout");

constexpr int SDF_NC = 8;            // vertex chunks per problem in the entry kernel (one workgroup each)
constexpr int SDF_NIT = 2;           // 64-vertex rows per wave of a chunk: nv <= SDF_NC * 8 waves * SDF_NIT * 64 = 8192

// per (problem, vertex chunk): partial sums of S and of the box adjoint, entries written (at the chunk's own offset)
struct SdfChunk { double S, gc0, gc1, gc2, gs; int cnt, pad; };
static_assert(sizeof(SdfChunk) == 48, "chunk record");
#ifndef SDF_NS_
#define SDF_NS_ 8                   // (-DSDF_NS_=1 reproduces the single-chain summation order bit for bit: the control experiment)
#endif
constexpr int SDF_NS = SDF_NS_;      // workgroups (entry slices) per problem in the pull-back

// Kernel 1 of the adjoint, grid (SDF_NC, B): a workgroup scans one eighth of a problem's vertices (the whole list
// through one CU took 9 us): partial S and box-adjoint sums in float64, and the chunk's entries - the vertices that carry
// gradient plus the box's arg-min / arg-max vertices - compacted in ascending order at the chunk's offset of the entry
// buffer, with the gradient of the sampling only (the box adjoint needs the sums of ALL chunks: sdf_pullback_kernel
// adds it when it loads an entry).
__global__ __launch_bounds__(SDF_ADJ_NT) void sdf_entries_kernel(int nv, const float* __restrict__ verts,
                                                                 const SdfBox* __restrict__ box, const float4* __restrict__ samp,
                                                                 const int* __restrict__ gate,
                                                                 SdfEntry* __restrict__ entries, SdfChunk* __restrict__ chunks) {
    __shared__ double sh_d[8][5];
    __shared__ int sh_cnt[8];
    const int b = blockIdx.y, y = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (gate && !gate[b]) return;
    const SdfBox bx = box[b];
    const float* vb = verts + (size_t)b * nv * 3;
    const float4* sb = samp + (size_t)b * nv;
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
            q[i] = sb[vc];
            px[i] = vb[3 * vc]; py[i] = vb[3 * vc + 1]; pz[i] = vb[3 * vc + 2];
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
    S = wave64_sum(S); gc0 = wave64_sum(gc0); gc1 = wave64_sum(gc1); gc2 = wave64_sum(gc2); gs = wave64_sum(gs);
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
                                                                  int* __restrict__ tickets, SdfAdj* __restrict__ out) {
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
    const int lo = (int)((long long)ntot * y / SDF_NS), n = (int)((long long)ntot * (y + 1) / SDF_NS);
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
    if (tid < KROWS) R.gcoef[tid] = racc; else R.gA[tid - KROWS] = racc;
    if (tid < 3) R.gtau[tid] = rgt;
    if (tid == 0) {
        double S = 0.0;
        for (int c = 0; c < SDF_NC; ++c) S += chunks[(size_t)b * SDF_NC + c].S;
        R.S = (float)S;
        __hip_atomic_store(tickets + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next round (a launch boundary away)
    }
}

// work area behind `entries`: [B][nv] SdfEntry | [B][SDF_NS] SdfAdj slice partials | [B][SDF_NC] SdfChunk | [B] tickets
// (the tickets must be zero before the first round: mvfit_api.hip clears them when it allocates the area; every round
// leaves them at zero)
static size_t sdf_part_offset(int B, int nv) { return ((size_t)B * nv * sizeof(SdfEntry) + 255) & ~(size_t)255; }
static size_t sdf_chunk_offset(int B, int nv) { return sdf_part_offset(B, nv) + (size_t)B * SDF_NS * sizeof(SdfAdj); }
size_t sdf_ticket_offset(int B, int nv) { return (sdf_chunk_offset(B, nv) + (size_t)B * SDF_NC * sizeof(SdfChunk) + 255) & ~(size_t)255; }
size_t sdf_work_bytes(int B, int nv) { return sdf_ticket_offset(B, nv) + (size_t)B * sizeof(int); }

hipError_t launch_sdf_term(const DevModel& M, const DevPose& P, const float* verts, int B, const int32_t* faces, int num_faces,
                           int G, const int* gate, SdfBox* box, float4* samp, void* entries, SdfAdj* adj, hipStream_t stream) {
    static_assert(SDF_ADJ_NT == KROWS + NJ * 12, "thread per output of the pull-back");
    unsigned char* wk = reinterpret_cast<unsigned char*>(entries);
    SdfAdj* part = reinterpret_cast<SdfAdj*>(wk + sdf_part_offset(B, M.nv));
    SdfChunk* chunks = reinterpret_cast<SdfChunk*>(wk + sdf_chunk_offset(B, M.nv));
    int* tickets = reinterpret_cast<int*>(wk + sdf_ticket_offset(B, M.nv));
    if (M.nv > SDF_NC * 8 * SDF_NIT * 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sdf_bbox_kernel, dim3(B), dim3(512), 0, stream, verts, M.nv, gate, box);
    hipLaunchKernelGGL(sdf_sample_kernel, dim3((M.nv + SDF_NT - 1) / SDF_NT, B), dim3(SDF_NT), 0, stream, verts, M.nv,
                       (const SdfBox*)box, faces, num_faces, G, gate, samp);
    hipLaunchKernelGGL(sdf_entries_kernel, dim3(SDF_NC, B), dim3(SDF_ADJ_NT), 0, stream, M.nv, verts, (const SdfBox*)box,
                       (const float4*)samp, gate, reinterpret_cast<SdfEntry*>(entries), chunks);
    hipLaunchKernelGGL(sdf_pullback_kernel, dim3(SDF_NS, B), dim3(SDF_ADJ_NT), 0, stream, M, P, gate, (const SdfBox*)box,
                       reinterpret_cast<const SdfEntry*>(entries), (const SdfChunk*)chunks, part, tickets, adj);
    return hipGetLastError();
}

}  // namespace mvfit

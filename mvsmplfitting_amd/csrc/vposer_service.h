// VPoser decoder service of the single-launch fit (code/model/VPoser.py:218-232: fc1 32->512, leaky_relu 0.2, fc2
// 512->512, leaky_relu, out 512->138; dropout is the identity in eval mode).
//
// One problem = one workgroup = one CU, and a matrix-vector product has no reuse: decoding in the problem's own
// workgroup streams the 1.4 MB of weights through that CU twice per closure (forward, transposed adjoint) - 37 us of a
// 62 us round, bound by what one CU can ingest.  The fit kernel leaves most of the chip idle (32 problems on 256 CUs), so
// the decoder runs on HELPER workgroups of the same launch instead, with the weights STATIONARY IN REGISTERS:
//
//   helper (set s, slice h), 512 threads, thread (wave w, lane l):
//     W1[o = tid][0..32)                       32 VGPRs   fc1 complete (every helper computes all of h1: 16 kFMA)
//     W2[64h + 8w + a][8l + c], a, c < 8       64 VGPRs   the slice's 64 fc2 units x all 512 inputs
//     W3[l + 64r][64h + 8w + a], r < 3, a < 8  24 VGPRs   the 138 outputs x the slice's 64 units
//
//   forward   z[32]  ->  helper h:  h1 = lrelu(W1 z + b1);  h2[slice] = lrelu(W2[slice] h1 + b2[slice]);
//                                   out_h = W3[:, slice] h2[slice]            (138 partial sums)
//             problem:  out = b3 + sum_h out_h   (h ascending)
//   adjoint   g_out[138] -> helper h:  g2 = (W3[:, slice]^T g_out) * lrelu'(pre2[slice]);  g1_h = W2[slice]^T g2  (512);
//                                      gz_h = W1^T (lrelu'(pre1) * g1_h)   (32 partial sums; W1^T and the mask are linear)
//             problem:  g_z += sum_h gz_h
//
// A set of 8 helpers (one per slice) serves the problems b with b % nsets == s (up to 24 of them); request and answers travel as
// data-tagged 8-byte granules {float, tag} written by single sc1 (write-through) stores and polled with agent-scope loads
// (guide: hand-off 0.8-1.5 us, placement-independent); tag = request number << 2 | kind, so a granule of an older request
// can never be taken for the current one.  The problem sends its next request only after it has consumed all eight
// answers of the previous one: neither side can overwrite words the other still reads.
// All reductions have a fixed association (ascending k per lane, a fixed exchange tree across lanes, waves and helpers in
// ascending order): the result does not depend on timing or placement.  It is NOT bit-identical to the in-workgroup
// decoder (closure_device.h, used by the chained rounds and the closure call): another summation order of the same fp32
// products (relative difference ~1e-7).
#pragma once
#include <hip/hip_runtime.h>

#include "wave_ops.h"

namespace mvfit {

constexpr int VPS_SLICES = 8;           // helpers per set = slices of the 512 fc2 units
constexpr int VPS_PMAX = 24;            // problems per set (wave w polls the slots w, w + 8, w + 16)
constexpr int VPS_MAX_SETS = 16;          // 16 for launches of <= 32 problems, else 8
constexpr int VPS_GRAN = 144;           // granules per request / per answer (138 used)
constexpr unsigned VPS_FWD = 1u, VPS_BWD = 2u, VPS_BYE = 3u;

struct VpService {
    unsigned long long* req;            // [nsets][VPS_PMAX][VPS_GRAN]
    unsigned long long* resp;           // [nsets][VPS_PMAX][VPS_SLICES][VPS_GRAN]
    unsigned* stat;                     // [0] answers that timed out (the problem then decodes locally), [1] helpers that gave up
    int nsets, nprob;                   // nsets == 0: no service in this launch
    int fault, pad_;                    // test hook (MVFIT_VP_FAULT=1): the helpers leave at once - every problem must time out and decode locally
};

__device__ __forceinline__ unsigned long long vps_pack(float v, unsigned tag) {
    return ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned vps_tag(unsigned long long g) { return (unsigned)(g >> 32); }
__device__ __forceinline__ float vps_val(unsigned long long g) { return __builtin_bit_cast(float, (unsigned)g); }
__device__ __forceinline__ void vps_store(unsigned long long* p, float v, unsigned tag) {
    __hip_atomic_store(p, vps_pack(v, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long vps_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// v[N] per lane -> the wave total of ONE element per lane: exchange steps over lane bits 5, 4, ... halve the vector
// (a lane keeps the half its bit selects and adds its partner's copy of that half), the remaining lane bits are plain
// butterflies.  Partners: the other 32-lane half / the other 16-lane row (v_permlane32_swap / v_permlane16_swap: the swap
// IS the exchange, one instruction per pair of elements), then the mirror lane of the row, of the 8-lane group, of the
// quad (DPP modifiers on the add) - every partner differs in the step's bit and agrees in all higher ones, which is all
// a reduce-scatter needs.  Returns element e(l) = the kept-half bits, most significant first: N = 8: 4 b5 + 2 b4 + b3;
// N = 32: 16 b5 + 8 b4 + 4 b3 + 2 b2 + b1.  Same bits in all lanes that share e(l); fixed association.
template <int M>
__device__ __forceinline__ float vps_partner(float x) {
    if constexpr (M == 8) return dpp_mov<DPP_MIRROR>(x);
    else if constexpr (M == 4) return dpp_mov<DPP_HALF_MIRROR>(x);
    else if constexpr (M == 2) return dpp_mov<DPP_XOR2>(x);
    else return dpp_mov<DPP_XOR1>(x);
}
template <int N, int M0 = 32>
__device__ __forceinline__ float wave_reduce_scatter(float (&v)[N], int lane) {
    if constexpr (N > 1) {
        float u[N / 2];
        if constexpr (M0 >= 16) {
#pragma unroll
            for (int j = 0; j < N / 2; ++j) {
                const unsigned a = __builtin_bit_cast(unsigned, v[j]), b = __builtin_bit_cast(unsigned, v[j + N / 2]);
                if constexpr (M0 == 32) {
                    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
                    u[j] = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
                } else {
                    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
                    u[j] = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
                }
            }
        } else {
            const bool up = (lane & M0) != 0;
#pragma unroll
            for (int j = 0; j < N / 2; ++j) {
                const float send = up ? v[j] : v[j + N / 2];
                const float keep = up ? v[j + N / 2] : v[j];
                u[j] = keep + vps_partner<M0>(send);
            }
        }
        return wave_reduce_scatter<N / 2, M0 / 2>(u, lane);
    } else {
        float t = v[0];
        if constexpr (M0 >= 32) { float a, b; swap_pair<true>(t, a, b); t = a + b; }
        if constexpr (M0 >= 16) { float a, b; swap_pair<false>(t, a, b); t = a + b; }
        if constexpr (M0 >= 8) t += vps_partner<8>(t);
        if constexpr (M0 >= 4) t += vps_partner<4>(t);
        if constexpr (M0 >= 2) t += vps_partner<2>(t);
        if constexpr (M0 >= 1) t += vps_partner<1>(t);
        return t;
    }
}
// the same for the 32 products w[i] * g, formed inside the first exchange step (16 live values instead of 32)
__device__ __forceinline__ float wave_reduce_scatter_scaled32(const float (&w)[32], float g, int lane) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, w[j] * g), __builtin_bit_cast(unsigned, w[j + 16] * g), false, false);
        v[j] = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
    return wave_reduce_scatter<16, 16>(v, lane);
}
// workgroup barrier that orders LDS accesses only: __syncthreads() also drains the vector-memory counter, i.e. waits for
// the write-through answer stores of the previous request (~1 us each)
__device__ __forceinline__ void vps_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ int vps_lane_of8(int a) { return ((a >> 2) & 1) << 5 | ((a >> 1) & 1) << 4 | (a & 1) << 3; }

struct VpHelperLds {
    __attribute__((aligned(16))) float req[VPS_PMAX][VPS_GRAN];
    __attribute__((aligned(16))) float h1[512];
    unsigned long long mask1[VPS_PMAX][8];     // fc1 units with a positive pre-activation: one ballot per wave
    float mask2[VPS_PMAX][64];                 // lrelu' of the slice's fc2 units
    __attribute__((aligned(16))) float part3[8][192];
    __attribute__((aligned(16))) float partk[8][512];
    float partz[8][32];
    unsigned ready[VPS_PMAX];           // tag of the request wave p found complete (0: none)
    int quit;
};

// Host-side layout of the register tiles (mvfit_create): float4 words, thread-minor, so that a helper's start-up loads
// are contiguous 8 KB rows.  tw2[h][j = 2a + half][tid] = W2[64h + 8w + a][8l + 4 half .. + 3];
// tw3[h][j = 2r + half][tid] = W3[l + 64r][64h + 8w + 4 half .. + 3] (zero rows for o >= 138).
struct VpTiles { const float4* tw2; const float4* tw3; const float* w1T; const float* b1; const float* b2; };

// The helper's main loop; returns when every problem of its set has said goodbye (or nothing arrived for 0.2 s).
__device__ __attribute__((noinline)) void vposer_helper(const VpTiles T, const VpService V, unsigned char* smem, int s, int h) {      // (by value: a reference would pin the kernel argument they are members of to the stack, and the problem workgroups read it from there every round)
    if (V.fault) { if (threadIdx.x == 0) atomicAdd(V.stat + 1, 1u); return; }
    VpHelperLds& S = *reinterpret_cast<VpHelperLds*>(smem);
    const int tid = threadIdx.x, w = tid >> 6;
    int l = tid & 63;
    const int nmine = (V.nprob - s + V.nsets - 1) / V.nsets;          // problems s, s + nsets, ...
    float w1[32], w2[8][8], w3[3][8];
#pragma unroll
    for (int i = 0; i < 32; ++i) w1[i] = T.w1T[i * 512 + tid];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float4 q = T.tw2[((size_t)h * 16 + j) * 512 + tid];
        w2[j >> 1][4 * (j & 1) + 0] = q.x; w2[j >> 1][4 * (j & 1) + 1] = q.y;
        w2[j >> 1][4 * (j & 1) + 2] = q.z; w2[j >> 1][4 * (j & 1) + 3] = q.w;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float4 q = T.tw3[((size_t)h * 6 + j) * 512 + tid];
        w3[j >> 1][4 * (j & 1) + 0] = q.x; w3[j >> 1][4 * (j & 1) + 1] = q.y;
        w3[j >> 1][4 * (j & 1) + 2] = q.z; w3[j >> 1][4 * (j & 1) + 3] = q.w;
    }
    const float b1o = T.b1[tid];
    const int e8 = ((l >> 5) & 1) * 4 + ((l >> 4) & 1) * 2 + ((l >> 3) & 1);
    const int e32 = ((l >> 5) & 1) * 16 + ((l >> 4) & 1) * 8 + ((l >> 3) & 1) * 4 + ((l >> 2) & 1) * 2 + ((l >> 1) & 1);
    const float b2e = T.b2[64 * h + 8 * w + e8];
    unsigned expect[3] = {1u, 1u, 1u};                     // wave w: number of the next request of the slots w, w + 8, w + 16
    unsigned alive = nmine >= 32 ? 0xffffffffu : ((1u << nmine) - 1u);
    if (tid == 0) S.quit = 0;
    long long t_last = wall_clock64();
    __syncthreads();

    while (alive) {
        asm volatile("" : "+v"(l));              // keeps the lane-derived addresses of the loop body out of loop-invariant registers
        // ---- poll: wave w watches the request slots of the problems w, w + 8, w + 16 of the set ----
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int p = w + 8 * k;
            if (p >= nmine) break;
            unsigned found = 0u;
            if ((alive >> p) & 1u) {
                const unsigned long long* rq = V.req + ((size_t)s * VPS_PMAX + p) * VPS_GRAN;
                const unsigned long long g0 = vps_load(rq + l), g1 = vps_load(rq + 64 + l);
                const unsigned long long g2 = l < VPS_GRAN - 128 ? vps_load(rq + 128 + l) : 0ull;
                const unsigned t0 = (unsigned)__builtin_amdgcn_readfirstlane((int)vps_tag(g0));
                // (a BYE may carry any number from the expected one on: a problem that gave up on a late answer numbers it
                // past the request it no longer waits for)
                if ((t0 >> 2) == expect[k] || ((t0 & 3u) == VPS_BYE && (t0 >> 2) > expect[k])) {
                    const unsigned kind = t0 & 3u;
                    bool ok = true;
                    if (kind == VPS_FWD) ok = l >= 32 || vps_tag(g0) == t0;
                    else if (kind == VPS_BWD) ok = vps_tag(g0) == t0 && vps_tag(g1) == t0 && (128 + l >= 138 || vps_tag(g2) == t0);
                    if (__all(ok)) {
                        found = t0;
                        S.req[p][l] = vps_val(g0);
                        S.req[p][64 + l] = vps_val(g1);
                        if (l < VPS_GRAN - 128) S.req[p][128 + l] = 128 + l < 138 ? vps_val(g2) : 0.f;
                        expect[k] += 1u;
                    }
                }
            }
            if (l == 0) S.ready[p] = found;
        }
        if (tid == 0 && wall_clock64() - t_last > 20000000) S.quit = 1;          // 0.2 s without a request
        __syncthreads();
        if (S.quit) { if (tid == 0) atomicAdd(V.stat + 1, 1u); break; }
        bool any = false;
        for (int p = 0; p < nmine; ++p) {
            const unsigned tag = S.ready[p];                // block-uniform
            if (!tag) continue;
            any = true;
            const unsigned kind = tag & 3u;
            if (kind == VPS_BYE) { alive &= ~(1u << p); continue; }
            unsigned long long* out = V.resp + (((size_t)s * VPS_PMAX + p) * VPS_SLICES + h) * VPS_GRAN;
#ifdef MVFIT_TIMING
            const long long tq0 = clock64();
#endif
            if (kind == VPS_FWD) {
                float c4[4] = {0.f, 0.f, 0.f, 0.f};             // four chains of eight: z[i], i = q (mod 4)
#pragma unroll
                for (int i4 = 0; i4 < 8; ++i4) {
                    const float4 z = *reinterpret_cast<const float4*>(&S.req[p][4 * i4]);
                    c4[0] = fmaf(w1[4 * i4], z.x, c4[0]); c4[1] = fmaf(w1[4 * i4 + 1], z.y, c4[1]);
                    c4[2] = fmaf(w1[4 * i4 + 2], z.z, c4[2]); c4[3] = fmaf(w1[4 * i4 + 3], z.w, c4[3]);
                }
                const float s1 = b1o + ((c4[0] + c4[1]) + (c4[2] + c4[3]));
                S.h1[tid] = s1 > 0.f ? s1 : 0.2f * s1;
                const unsigned long long pos = __ballot(s1 > 0.f);
                if (l == 0) S.mask1[p][w] = pos;
                vps_lds_barrier();
                const float4 ha = *reinterpret_cast<const float4*>(&S.h1[8 * l]), hb = *reinterpret_cast<const float4*>(&S.h1[8 * l + 4]);
                const float hk[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
                float pa[8];
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    float acc = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc = fmaf(w2[a][c], hk[c], acc);
                    pa[a] = acc;
                }
                const float pre2 = b2e + wave_reduce_scatter<8>(pa, l);
                const float h2 = pre2 > 0.f ? pre2 : 0.2f * pre2;
                if ((l & 7) == 0) S.mask2[p][8 * w + e8] = pre2 > 0.f ? 1.0f : 0.2f;
                float q[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const float ha_ = lane_read(h2, vps_lane_of8(a));
#pragma unroll
                    for (int r = 0; r < 3; ++r) q[r] = fmaf(w3[r][a], ha_, q[r]);
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) S.part3[w][l + 64 * r] = q[r];
                vps_lds_barrier();
                if (tid < 138) {
                    float acc = S.part3[0][tid];
#pragma unroll
                    for (int ww = 1; ww < 8; ++ww) acc += S.part3[ww][tid];
                    vps_store(out + tid, acc, tag);
                }
            } else {
                float pa[8];
                const float g0 = S.req[p][l], g1 = S.req[p][64 + l], g2 = l < 16 ? S.req[p][128 + l] : 0.f;
#pragma unroll
                for (int a = 0; a < 8; ++a) pa[a] = fmaf(w3[2][a], g2, fmaf(w3[1][a], g1, w3[0][a] * g0));
                const float gg = wave_reduce_scatter<8>(pa, l) * S.mask2[p][8 * w + e8];
                float pk[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const float ga = lane_read(gg, vps_lane_of8(a));
#pragma unroll
                    for (int c = 0; c < 8; ++c) pk[c] = fmaf(w2[a][c], ga, pk[c]);
                }
                *reinterpret_cast<float4*>(&S.partk[w][8 * l]) = make_float4(pk[0], pk[1], pk[2], pk[3]);
                *reinterpret_cast<float4*>(&S.partk[w][8 * l + 4]) = make_float4(pk[4], pk[5], pk[6], pk[7]);
                vps_lds_barrier();
                float gk = S.partk[0][tid];
#pragma unroll
                for (int ww = 1; ww < 8; ++ww) gk += S.partk[ww][tid];
                gk *= ((S.mask1[p][w] >> l) & 1ull) ? 1.0f : 0.2f;
                const float gz = wave_reduce_scatter_scaled32(w1, gk, l);
                if ((l & 1) == 0) S.partz[w][e32] = gz;
                vps_lds_barrier();
                if (tid < 32) {
                    float acc = S.partz[0][tid];
#pragma unroll
                    for (int ww = 1; ww < 8; ++ww) acc += S.partz[ww][tid];
                    vps_store(out + tid, acc, tag);
                }
            }
#ifdef MVFIT_TIMING
            if (s == 0 && h == 0 && tid == 0) { g_dbg[32 + (kind == VPS_FWD ? 0 : 1)] += clock64() - tq0; g_dbg[34 + (kind == VPS_FWD ? 0 : 1)] += 1; }
#endif
        }
#ifdef MVFIT_TIMING
        if (s == 0 && h == 0 && tid == 0) { g_dbg[36] += 1; if (any) g_dbg[37] += 1; }
#endif
        if (any) t_last = wall_clock64();
        else __builtin_amdgcn_s_sleep(2);
        __syncthreads();                                    // ready[] / the staging arrays are rewritten by the next poll
    }
}

}  // namespace mvfit

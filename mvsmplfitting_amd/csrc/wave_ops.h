// wave64 cross-lane reductions on the DPP path (no LDS, no ds_bpermute).
//
// A 16-lane DPP row is the reduction unit: four butterfly steps (quad_perm xor 1, quad_perm xor 2,
// row_half_mirror, row_mirror) leave the bit-identical row total in every lane of the row
// (fp add is commutative, so both partners of a step compute the same bits).  A whole-wave total
// continues with v_permlane16_swap / v_permlane32_swap (gfx950) across rows and halves.
#pragma once
#include <hip/hip_runtime.h>

namespace mvfit {

// developer build (-DMVFIT_TIMING): per-phase shader-clock accumulation of workgroup 0
#ifdef MVFIT_TIMING
static __device__ long long g_dbg[96];
static __device__ long long g_dbg_last;
#define PH_T(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long now_ = clock64(); g_dbg[k] += now_ - g_dbg_last; g_dbg_last = now_; } } while (0)
#define PH_T0() do { if (blockIdx.x == 0 && threadIdx.x == 0) g_dbg_last = clock64(); } while (0)
#define PH_ADD(k, v) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_dbg[k] += (v); } while (0)
// probe of another thread of workgroup 0: cycles since t0_ (a clock64() value taken by that thread) into slot k
#define PH_CLK() clock64()
#define PH_W(k, who, t0_) do { if (blockIdx.x == 0 && (int)threadIdx.x == (who)) g_dbg[k] += clock64() - (t0_); } while (0)
#else
#define PH_CLK() 0ll
#define PH_W(k, who, t0_) do { } while (0)
#define PH_T(k) do { } while (0)
#define PH_T0() do { } while (0)
#define PH_ADD(k, v) do { } while (0)
#endif


template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}

constexpr int DPP_XOR1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // lane i <-> 7 - i within each 8
constexpr int DPP_MIRROR = 0x140;       // lane i <-> 15 - i within each 16

// 64-bit keys (ordered value << 32 | index) of the SDF term's bounding box (sdf_term.hip: sdf_box_reduce; round 6: the vertex pass
// reduces its tile's keys itself, vertex_pass.hip: vp_box_parts).  Order-preserving map float -> uint32 (total order of the finite
// values) and its inverse; min / max of keys over a 16-lane DPP row (the same result in every lane of the row)
__device__ __forceinline__ unsigned ord_bits(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_float(unsigned o) {
    const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __builtin_bit_cast(float, u);
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
    return __builtin_bit_cast(unsigned long long, dpp_mov<CTRL>(__builtin_bit_cast(double, v)));
}
template <bool MIN>
__device__ __forceinline__ unsigned long long row16_key(unsigned long long v) {
    auto pick = [](unsigned long long p, unsigned long long q) { return MIN ? (p < q ? p : q) : (p > q ? p : q); };
    v = pick(v, dpp_u64<DPP_XOR1>(v));
    v = pick(v, dpp_u64<DPP_XOR2>(v));
    v = pick(v, dpp_u64<DPP_HALF_MIRROR>(v));
    v = pick(v, dpp_u64<DPP_MIRROR>(v));
    return v;
}

template <typename T>
__device__ __forceinline__ T row16_sum(T v) {      // every lane of a 16-lane row gets the row total
    v += dpp_mov<DPP_XOR1>(v);
    v += dpp_mov<DPP_XOR2>(v);
    v += dpp_mov<DPP_HALF_MIRROR>(v);
    v += dpp_mov<DPP_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<DPP_XOR1>(v));
    v = fmaxf(v, dpp_mov<DPP_XOR2>(v));
    v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_mov<DPP_MIRROR>(v));
    return v;
}
__device__ __forceinline__ double row16_max(double v) {
    v = fmax(v, dpp_mov<DPP_XOR1>(v));
    v = fmax(v, dpp_mov<DPP_XOR2>(v));
    v = fmax(v, dpp_mov<DPP_HALF_MIRROR>(v));
    v = fmax(v, dpp_mov<DPP_MIRROR>(v));
    return v;
}

__device__ __forceinline__ float lane_read(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double lane_read(double v, int lane) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)b, lane);
    const int hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}

// v_permlane16_swap(v, v): vdst keeps rows 0/2 and receives them in rows 1/3, vsrc the other way round,
// so r[0] = [row0 row0 row2 row2] and r[1] = [row1 row1 row3 row3]; v_permlane32_swap(v, v) likewise
// gives [lo lo] and [hi hi].  r[0] (op) r[1] is then the pairwise combination, same operand order and
// hence the same bits in both partner lanes.
__device__ __forceinline__ void swap16(unsigned v, unsigned& a, unsigned& b) {
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    a = r[0]; b = r[1];
}
__device__ __forceinline__ void swap32(unsigned v, unsigned& a, unsigned& b) {
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    a = r[0]; b = r[1];
}
template <bool HALF>   // HALF: combine the two 32-lane halves, else the two 16-lane rows of each half
__device__ __forceinline__ void swap_pair(float v, float& a, float& b) {
    unsigned ua, ub;
    if (HALF) swap32(__builtin_bit_cast(unsigned, v), ua, ub); else swap16(__builtin_bit_cast(unsigned, v), ua, ub);
    a = __builtin_bit_cast(float, ua); b = __builtin_bit_cast(float, ub);
}
template <bool HALF>
__device__ __forceinline__ void swap_pair(double v, double& a, double& b) {
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
    unsigned la, lb, ha, hb;
    if (HALF) { swap32((unsigned)bits, la, lb); swap32((unsigned)(bits >> 32), ha, hb); }
    else { swap16((unsigned)bits, la, lb); swap16((unsigned)(bits >> 32), ha, hb); }
    a = __builtin_bit_cast(double, ((unsigned long long)ha << 32) | la);
    b = __builtin_bit_cast(double, ((unsigned long long)hb << 32) | lb);
}

// A value that IS the same in every lane (a whole-wave reduction, a broadcast read) but that the compiler cannot prove
// uniform: through v_readfirstlane it becomes a scalar, and every branch that depends on it a scalar branch instead of
// an exec-masked region with its phi copies (the L-BFGS state machine is wave-uniform control flow on one wave).
__device__ __forceinline__ float wave_uniform(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double wave_uniform(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}

// whole-wave total, bit-identical in every lane: 4 DPP steps inside the rows, then rows, then halves
template <typename T>
__device__ __forceinline__ T wave64_sum(T v) {
    v = row16_sum(v);
    T a, b;
    swap_pair<false>(v, a, b); v = a + b;
    swap_pair<true>(v, a, b);  v = a + b;
    return v;
}
__device__ __forceinline__ float wave64_max(float v) {
    v = row16_max(v);
    float a, b;
    swap_pair<false>(v, a, b); v = fmaxf(a, b);
    swap_pair<true>(v, a, b);  v = fmaxf(a, b);
    return v;
}
__device__ __forceinline__ double wave64_max(double v) {
    v = row16_max(v);
    double a, b;
    swap_pair<false>(v, a, b); v = fmax(a, b);
    swap_pair<true>(v, a, b);  v = fmax(a, b);
    return v;
}

// LDS traffic between lanes of ONE wave needs no s_barrier: the wave's DS operations execute in
// order; this only keeps the compiler from caching / reordering across the hand-off.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace mvfit

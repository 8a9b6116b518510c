// Wave-level butterflies of the deterministic reduction trees, on the VALU instead of the LDS crossbar (gfx950 only).
//
// The trees are the ones the kernels have always used -- partner lane i ^ m for m = 32, 16, 8, 4, 2, 1 (a wave) or m = 4, 2, 1 (an
// 8-lane group) -- so every sum keeps its association and its bits; only the data movement changed: `__shfl_xor` is two
// ds_bpermute_b32 per double and step (address VGPR, LDS pipe, ~100 cycles of latency each, and the LDS pipe is what bounds the
// one-workgroup solver of small partitions: profiles/r3_notes.md), these are v_permlane32_swap / v_permlane16_swap / v_mov_dpp.
//   m = 32, 16: the swap instructions exchange half-waves / odd-even rows of two copies of x, giving (x_i, x_{i^m}) in the two
//               results up to operand order -- and a + b == b + a bit for bit;
//   m = 8:      row_ror:8 IS lane i ^ 8 inside a row of 16;
//   m = 4:      after the m = 8 step the value depends on lane % 8 only, and (lane + 4) % 8 == (lane % 8) ^ 4: row_ror:4;
//               for a general x (the 8-lane groups) two bank-masked row shifts build the true i ^ 4;
//   m = 2, 1:   quad_perm [2,3,0,1] and [1,0,3,2].
// All 64 lanes must be active (the callers' loops end converged; the old shuffles needed the same).
// tools/wave_selftest.hip compares every function with its __shfl_xor form bit for bit on the GPU (tests/test_gpu_parity.py runs it).
#pragma once
#include <hip/hip_runtime.h>

template <int CTRL, int BANK_MASK = 0xF>
__device__ __forceinline__ double mlx_dpp64(double old, double x)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, 0xF, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, 0xF, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}
// (x, x_{i^32}) / (x, x_{i^16}) up to the order of the pair
__device__ __forceinline__ void mlx_swap32(double x, double &a, double &b)
{
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    a = __hiloint2double((int)h[0], (int)l[0]);
    b = __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ void mlx_swap16(double x, double &a, double &b)
{
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    a = __hiloint2double((int)h[0], (int)l[0]);
    b = __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ double mlx_xor8(double x) { return mlx_dpp64<0x128>(x, x); }            // row_ror:8
__device__ __forceinline__ double mlx_ror4(double x) { return mlx_dpp64<0x124>(x, x); }            // row_ror:4 (== i ^ 4 for 8-periodic x)
__device__ __forceinline__ double mlx_xor4(double x)                                               // true i ^ 4
{
    const double t = mlx_dpp64<0x104, 0x5>(x, x);      // row_shl:4 into banks 0 and 2 (lanes 0-3, 8-11 take lane + 4)
    return mlx_dpp64<0x114, 0xA>(t, x);                // row_shr:4 into banks 1 and 3 (lanes 4-7, 12-15 take lane - 4)
}
__device__ __forceinline__ double mlx_xor2(double x) { return mlx_dpp64<0x4E>(x, x); }             // quad_perm [2,3,0,1]
__device__ __forceinline__ double mlx_xor1(double x) { return mlx_dpp64<0xB1>(x, x); }             // quad_perm [1,0,3,2]

// lane `src` (wave-uniform) of x to every lane
__device__ __forceinline__ double mlx_wave_bcast(double x, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src), hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double mlx_wave_allreduce_sum(double x)
{
    double a, b;
    mlx_swap32(x, a, b); x = a + b;
    mlx_swap16(x, a, b); x = a + b;
    x += mlx_xor8(x);
    x += mlx_ror4(x);
    x += mlx_xor2(x);
    x += mlx_xor1(x);
    return x;
}
__device__ __forceinline__ double mlx_wave_allreduce_max(double x)
{
    double a, b;
    mlx_swap32(x, a, b); x = fmax(a, b);
    mlx_swap16(x, a, b); x = fmax(a, b);
    x = fmax(x, mlx_xor8(x));
    x = fmax(x, mlx_ror4(x));
    x = fmax(x, mlx_xor2(x));
    x = fmax(x, mlx_xor1(x));
    return x;
}
// sum over the 8-lane group of the lane (m = 4, 2, 1)
__device__ __forceinline__ double mlx_group8_allreduce_sum(double x)
{
    x += mlx_xor4(x);
    x += mlx_xor2(x);
    x += mlx_xor1(x);
    return x;
}

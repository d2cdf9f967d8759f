// Wave-level (64-lane) helpers shared by the inverse and self-energy kernels.  gfx950.
#pragma once
#include "negf_common.h"

typedef double d4 __attribute__((ext_vector_type(4)));

// ---- wave-level arg-max of (v, key): larger v wins, ties -> smaller key -------------
// DPP lane moves inside each row of 16 lanes (xor 1, xor 2, half mirror, mirror), then
// the four row results are combined through v_readlane.  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ void dpp_step(double& v, int& key)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    const int okey = __builtin_amdgcn_update_dpp(key, key, CTRL, 0xF, 0xF, false);
    const double ov = __hiloint2double(ohi, olo);
    const bool take = (ov > v) | ((ov == v) & (okey < key));
    v = take ? ov : v; key = take ? okey : key;
}

__device__ __forceinline__ void wave_argmax(double& v, int& key)
{
    dpp_step<0xB1>(v, key);      // quad_perm [1,0,3,2]
    dpp_step<0x4E>(v, key);      // quad_perm [2,3,0,1]
    dpp_step<0x141>(v, key);     // row_half_mirror
    dpp_step<0x140>(v, key);     // row_mirror  -> every lane of a row holds the row result
    double bv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0),
                                 __builtin_amdgcn_readlane(__double2loint(v), 0));
    int bk = __builtin_amdgcn_readlane(key, 0);
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), r * 16);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), r * 16);
        const int k = __builtin_amdgcn_readlane(key, r * 16);
        const double ov = __hiloint2double(hi, lo);
        const bool take = (ov > bv) | ((ov == bv) & (k < bk));
        bv = take ? ov : bv; bk = take ? k : bk;
    }
    v = bv; key = bk;
}

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// complex 16x16 tile multiply-accumulate on the FP64 matrix cores:
//   (accr + i acci) += (pa) * (qb)  summed over the 4-deep k slice held by the wave
__device__ __forceinline__ void zmfma(d4& accr, d4& acci, cplx pa, cplx qb)
{
    accr = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.x, qb.x, accr, 0, 0, 0);
    accr = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.y, -qb.y, accr, 0, 0, 0);
    acci = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.x, qb.y, acci, 0, 0, 0);
    acci = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.y, qb.x, acci, 0, 0, 0);
}

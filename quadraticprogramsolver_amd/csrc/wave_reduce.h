// wave_reduce.h -- wave64 sum without the LDS pipe.
//
// `__shfl_xor` compiles to ds_bpermute: every step of a shuffle reduction is an LDS-pipe instruction (two for a double).  In the
// row-dot kernels one reduction per matrix row per wave, plus the read-back of the per-wave partials by every thread, made the
// narrow (KC = 1, batched n = 1024) sweep LDS-issue bound at ~2.9 TB/s.  The classic GCN DPP sequence (row_shr 1,2,3 / 4 / 8,
// row_bcast 15 / 31) does the same sum on the VALU; lane 63 ends up with the total and v_readlane hands it to the whole wave.
#pragma once
#include <hip/hip_runtime.h>

namespace qps {

template <int CTRL, int ROW_MASK, int BANK_MASK> __device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));   // lanes without a source: 0
}
template <int CTRL, int ROW_MASK, int BANK_MASK> __device__ __forceinline__ double dpp_get(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, BANK_MASK, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, BANK_MASK, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ float lane_get(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ double lane_get(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// sum over the 64 lanes of the wave, returned in every lane (wave-uniform); fixed association order
template <typename T> __device__ __forceinline__ T wave_sum_all(T a) {
    T t = a + dpp_get<0x111, 0xf, 0xf>(a);          // row_shr:1
    t = t + dpp_get<0x112, 0xf, 0xf>(a);            // row_shr:2
    t = t + dpp_get<0x113, 0xf, 0xf>(a);            // row_shr:3   -> t[i] = a[i-3 .. i] inside a row of 16
    t = t + dpp_get<0x114, 0xf, 0xe>(t);            // row_shr:4, lanes 4-15 of a row
    t = t + dpp_get<0x118, 0xf, 0xc>(t);            // row_shr:8, lanes 8-15: lane 15 holds the row sum
    t = t + dpp_get<0x142, 0xa, 0xf>(t);            // row_bcast:15 into rows 1 and 3
    t = t + dpp_get<0x143, 0xc, 0xf>(t);            // row_bcast:31 into rows 2 and 3: lane 63 holds the wave sum
    return lane_get(t, 63);
}
// sum of lanes 0..7 (other lanes ignored), returned in every lane
template <typename T> __device__ __forceinline__ T lanes8_sum_all(T a) {
    T t = a + dpp_get<0x111, 0xf, 0xf>(a);
    t = t + dpp_get<0x112, 0xf, 0xf>(a);
    t = t + dpp_get<0x113, 0xf, 0xf>(a);
    t = t + dpp_get<0x114, 0xf, 0xe>(t);            // lane 7 = a[0..7]
    return lane_get(t, 7);
}

// sum over each aligned row of 16 lanes; the total is valid in lane 15 of the row (other lanes hold partial prefixes)
template <typename T> __device__ __forceinline__ T row16_sum_last(T a) {
    T t = a + dpp_get<0x111, 0xf, 0xf>(a);
    t = t + dpp_get<0x112, 0xf, 0xf>(a);
    t = t + dpp_get<0x113, 0xf, 0xf>(a);
    t = t + dpp_get<0x114, 0xf, 0xe>(t);
    t = t + dpp_get<0x118, 0xf, 0xc>(t);
    return t;
}
// all-reduce sums over aligned groups of 4 / 8 lanes, every lane of the group gets the total (DPP quad_perm / row_half_mirror)
template <typename T> __device__ __forceinline__ T quad_sum_all(T v) {
    v = v + dpp_get<0xB1, 0xf, 0xf>(v);             // quad_perm [1,0,3,2]
    v = v + dpp_get<0x4E, 0xf, 0xf>(v);             // quad_perm [2,3,0,1]
    return v;
}
template <typename T> __device__ __forceinline__ T oct_sum_all(T v) {
    v = quad_sum_all(v);
    v = v + dpp_get<0x141, 0xf, 0xf>(v);            // row_half_mirror: lane i <-> 7 - i inside each group of 8
    return v;
}

}  // namespace qps

// bfloat16 storage helpers for the gfx950 kernels of libmpa_hip.so.
//
// The bf16 feature path (BASELINE configs 3 and 5) keeps FEATURES and their gradients in
// bf16 in HBM -- the path is HBM-bound, so the bytes are what is bought -- while every kernel computes
// in fp32 registers: coordinates, distances, indices, BatchNorm statistics, softmax, accumulators,
// parameters, parameter gradients and optimizer state stay fp32.  Kernels are written once over a
// storage type T in {float, bf16_t}; these helpers are the only place that knows the difference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// two fp32 -> one dword of two bf16 (round to nearest even; the cast compiles to v_cvt_pk_bf16_f32,
// which keeps NaNs NaN -- MI355X_MICROARCH.md "Correctness boundaries")
__device__ __forceinline__ unsigned mpa_pack_bf16x2(float lo, float hi)
{
    bf16x2_t v;
    v[0] = (bf16_t)lo;
    v[1] = (bf16_t)hi;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float mpa_bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float mpa_bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// ---- 4 consecutive elements of storage type T <-> float4 (16 B of fp32, 8 B of bf16)
template <typename T> __device__ __forceinline__ float4 mpa_ld4(const T *p);
template <> __device__ __forceinline__ float4 mpa_ld4<float>(const float *p) { return *reinterpret_cast<const float4 *>(p); }
template <> __device__ __forceinline__ float4 mpa_ld4<bf16_t>(const bf16_t *p)
{
    const uint2 u = *reinterpret_cast<const uint2 *>(p);
    return make_float4(mpa_bf16_lo(u.x), mpa_bf16_hi(u.x), mpa_bf16_lo(u.y), mpa_bf16_hi(u.y));
}
template <typename T> __device__ __forceinline__ void mpa_st4(T *p, float4 v);
template <> __device__ __forceinline__ void mpa_st4<float>(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
template <> __device__ __forceinline__ void mpa_st4<bf16_t>(bf16_t *p, float4 v)
{
    *reinterpret_cast<uint2 *>(p) = make_uint2(mpa_pack_bf16x2(v.x, v.y), mpa_pack_bf16x2(v.z, v.w));
}

// ---- single elements
template <typename T> __device__ __forceinline__ float mpa_ld1(const T *p);
template <> __device__ __forceinline__ float mpa_ld1<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float mpa_ld1<bf16_t>(const bf16_t *p)
{
    return __uint_as_float((unsigned)*reinterpret_cast<const unsigned short *>(p) << 16);
}
template <typename T> __device__ __forceinline__ void mpa_st1(T *p, float v);
template <> __device__ __forceinline__ void mpa_st1<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void mpa_st1<bf16_t>(bf16_t *p, float v) { *p = (bf16_t)v; }

// bytes a 4-element access of T must be aligned to
template <typename T> struct mpa_vec4_align { static constexpr uintptr_t mask = 15; };
template <> struct mpa_vec4_align<bf16_t> { static constexpr uintptr_t mask = 7; };

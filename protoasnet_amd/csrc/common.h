// Shared device helpers for the gfx950 kernels: dtype traits, 16-byte vector access, MFMA wrappers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/protoasnet_amd.h"

namespace pasn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// ---- host-side error plumbing -------------------------------------------------------------------
void set_error(const std::string& msg);
int check_launch(const char* what);
#define PASN_REQUIRE(cond, msg)                                        \
    do {                                                               \
        if (!(cond)) {                                                 \
            ::pasn::set_error(std::string(__func__) + ": " + (msg));   \
            return PASN_ERR_ARG;                                       \
        }                                                              \
    } while (0)

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// ---- dtype traits: one MFMA "k-chunk" is the 16 bytes a lane feeds to the matrix core ---------------
template <typename T>
struct Traits;
template <>
struct Traits<float> {
    static constexpr int CH = 4;      // elements per 16-byte lane chunk
    static constexpr int KSTEP = 8;   // K covered by the two lane halves of one fragment pair
    using frag = f32x4;
};
template <>
struct Traits<__bf16> {
    static constexpr int CH = 8;
    static constexpr int KSTEP = 16;
    using frag = bf16x8;
};

// D(32x32) += A(32xKSTEP) * B(KSTEPx32).  Lane l = (r = l & 31, h = l >> 5) supplies, for A row r and for
// B column r, the CH consecutive k values  k0 + h*CH .. k0 + h*CH + CH-1.
// bf16: one v_mfma_f32_32x32x16_bf16.  fp32: four v_mfma_f32_32x32x2_f32; MFMA j contracts the k pair
// {j, CH + j}; any k order is fine as long as A and B agree, which they do by construction.
__device__ __forceinline__ void mma32(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
}
// Accumulator element `reg` of lane l sits at column (l & 31), row acc_row(reg, l >> 5).
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

template <typename T>
__device__ __forceinline__ typename Traits<T>::frag zero_frag() {
    typename Traits<T>::frag z;
#pragma unroll
    for (int j = 0; j < Traits<T>::CH; ++j) z[j] = (T)0.0f;
    return z;
}
template <typename T>
__device__ __forceinline__ typename Traits<T>::frag load_frag(const T* p) {
    return *reinterpret_cast<const typename Traits<T>::frag*>(p);
}

// ---- 8-channel groups (the unit of the stencil / pooling kernels) ---------------------------------------
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = a[j];
        v[4 + j] = b[j];
    }
}
__device__ __forceinline__ void load8(const __bf16* p, float (&v)[8]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = v[j];
        b[j] = v[4 + j];
    }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}
__device__ __forceinline__ void store8(__bf16* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x8*>(p) = a;
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
    f32x4 a;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = v[j];
    *reinterpret_cast<f32x4*>(p) = a;
}
__device__ __forceinline__ void store4(__bf16* p, const float (&v)[4]) {
    bf16x4 a;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x4*>(p) = a;
}
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = a[j];
}
__device__ __forceinline__ void load4(const __bf16* p, float (&v)[4]) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)a[j];
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case PASN_ACT_RELU: return fmaxf(v, 0.0f);
        case PASN_ACT_SIGMOID: return sigmoidf_(v);
        case PASN_ACT_SWISH: return v * sigmoidf_(v);
        case PASN_ACT_ABS: return fabsf(v);
        default: return v;
    }
}

}  // namespace pasn

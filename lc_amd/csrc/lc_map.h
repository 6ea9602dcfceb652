// Element types of the network-output maps the f1 / f3 kernels read (dense front end, front end + selection, code decode, auxiliary
// losses): fp32 (the reference's precision) and the 16-bit types a mixed-precision backbone emits (BASELINE.json configs[2] bf16,
// configs[4] fp16; ptnet.py:68-82 hands the head's outputs over in the autocast type).  The maps are consumed in their own type --
// no up-cast copy in front of the kernel (at zlmo's shape, B = 64 x 21 x 128 x 128 code logits, that copy was a 132 MB pass in front
// of a 124 MB kernel) -- all arithmetic stays fp32, and gradients with respect to a map are written in the map's type (round to
// nearest even), which is what autograd hands a 16-bit leaf anyway.
// Same access pattern for every type: V = 4 consecutive elements per request are ONE 16-byte (fp32) or 8-byte (16-bit) access with the
// same element indexing, so every reduction adds the same values in the same order whatever the map type -- a 16-bit map gives bit
// for bit the result of the fp32 kernel on the up-cast values.
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

namespace lc {

enum MapDtype { kMapF32 = 0, kMapF16 = 1, kMapBF16 = 2 };  // include/lc_amd.h: LC_F32 / LC_F16 / LC_BF16

__host__ __device__ inline int map_elem_bytes(int dtype) { return dtype == kMapF32 ? 4 : 2; }

typedef float map_v4f_t __attribute__((ext_vector_type(4)));
typedef unsigned map_v2u_t __attribute__((ext_vector_type(2)));

template <typename T>
__device__ __forceinline__ float map_at(const T* q, size_t i) { return (float)q[i]; }

// V consecutive elements (V = 4: q aligned to four elements)
template <int V, typename T>
__device__ __forceinline__ void map_load(const T* q, float (&v)[V]) {
    static_assert(V == 1 || V == 4, "one element or four");
    if constexpr (V == 1) {
        v[0] = (float)q[0];
    } else if constexpr (sizeof(T) == 4) {
        const float4 t = *reinterpret_cast<const float4*>(q);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        const uint2 r = *reinterpret_cast<const uint2*>(q);
        T h[4];
        __builtin_memcpy(h, &r, 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (float)h[k];
    }
}
// The same four elements as they lie in memory, converted LATER: a kernel that keeps several requests in flight loads the raw words of all of
// them first -- with the conversion next to the load the compiler waits for each request before it issues the next (seen in the ISA of the
// code loss: s_waitcnt vmcnt after every request of an unrolled group of four, i.e. one request in flight).
template <typename T>
struct MapRaw4 {
    typedef uint2 type;
};
template <>
struct MapRaw4<float> {
    typedef float4 type;
};
template <typename T>
__device__ __forceinline__ typename MapRaw4<T>::type map_raw_load4(const T* q) {
    return *reinterpret_cast<const typename MapRaw4<T>::type*>(q);
}
template <typename T>
__device__ __forceinline__ float4 map_raw_cvt4(const typename MapRaw4<T>::type& r) {
    if constexpr (sizeof(T) == 4) {
        return r;
    } else {
        T h[4];
        __builtin_memcpy(h, &r, 8);
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    }
}
template <typename T>
__device__ __forceinline__ float4 map_load4(const T* q) {
    float v[4];
    map_load<4>(q, v);
    return make_float4(v[0], v[1], v[2], v[3]);
}

// fp32 -> the map's type, round to nearest even OF THE FP32 VALUE.  The value is made opaque first: left to itself the compiler folds the
// multiply that produced it into the conversion (v_fma_mixlo_f16: the exact product rounded once, straight to fp16), which differs from the
// fp32 kernel's result rounded to fp16 wherever the fp32 rounding lands on a tie (seen: 1 gradient of 70 000, in the subnormal range) --
// a better-than-specified answer that would make "the fp32 gradient, rounded" untrue.  The barrier costs no instruction.
template <typename T>
__device__ __forceinline__ T map_round(float v) {
    if constexpr (sizeof(T) == 2) asm volatile("" : "+v"(v));
    return (T)v;
}

// stream: a gradient that is written once and read by the next kernel of the backward pass (non-temporal store)
template <int V, typename T>
__device__ __forceinline__ void map_store(T* q, const float (&v)[V], bool stream = false) {
    static_assert(V == 1 || V == 4, "one element or four");
    if constexpr (V == 1) {
        q[0] = map_round<T>(v[0]);
    } else if constexpr (sizeof(T) == 4) {
        const map_v4f_t r = {v[0], v[1], v[2], v[3]};
        if (stream) __builtin_nontemporal_store(r, reinterpret_cast<map_v4f_t*>(q));
        else *reinterpret_cast<map_v4f_t*>(q) = r;
    } else {
        const T h[4] = {map_round<T>(v[0]), map_round<T>(v[1]), map_round<T>(v[2]), map_round<T>(v[3])};
        map_v2u_t r;
        __builtin_memcpy(&r, h, 8);
        if (stream) __builtin_nontemporal_store(r, reinterpret_cast<map_v2u_t*>(q));
        else *reinterpret_cast<map_v2u_t*>(q) = r;
    }
}

// one element of a small per-sample vector whose type is known at run time only (the weight scale: fp32 under autocast, the model's 16-bit
// type in a pure half-precision model)
__device__ __forceinline__ float map_scalar_at(const void* q, int dtype, size_t i) {
    return dtype == kMapF32 ? static_cast<const float*>(q)[i]
                            : (dtype == kMapF16 ? (float)static_cast<const _Float16*>(q)[i] : (float)static_cast<const __bf16*>(q)[i]);
}
__device__ __forceinline__ void map_scalar_put(void* q, int dtype, size_t i, float v) {
    if (dtype == kMapF32) static_cast<float*>(q)[i] = v;
    else if (dtype == kMapF16) static_cast<_Float16*>(q)[i] = map_round<_Float16>(v);
    else static_cast<__bf16*>(q)[i] = map_round<__bf16>(v);
}

// pointers that allow the four-element access of a map of `dtype` (null pointers pass)
inline bool map_aligned4(const void* q, int dtype) { return (reinterpret_cast<uintptr_t>(q) & (uintptr_t)(4 * map_elem_bytes(dtype) - 1)) == 0; }

}  // namespace lc

// Runs `...` with T bound to the element type of `dtype` (host side: picks the kernel instantiation).
#define LC_MAP_DISPATCH(dtype, ...)                                         \
    do {                                                                    \
        if ((dtype) == ::lc::kMapF16) { using T = _Float16; __VA_ARGS__; }  \
        else if ((dtype) == ::lc::kMapBF16) { using T = __bf16; __VA_ARGS__; } \
        else { using T = float; __VA_ARGS__; }                              \
    } while (0)

// ZebraPose-style binary surface codes -> normalised object coordinates.
//
// Replaces floatbits.py:130-160 (mod_logits2float_with_gt_bb_scripted: differentiable MSB-error decode used in training,
// ~15 elementwise ops + gather/scatter over (B,H,W,3,bits)) and floatbits.py:194-223 (mod_logits2float_bb: Gray-code
// decode used at inference), including the /(max/2)-1 normalisation of floatbits.py:108-118,162-180 and the strided
// sub-sampling of losses.py:163-184.  Works directly on the network layout (B,C,H,W), C = n0+n1+n2 code bits, one thread
// per (sampled) pixel: channel reads are coalesced across the pixels of a row; nothing is permuted or materialised.
#include "lc_common.h"
#include "lc_kernels.h"

namespace lc {
namespace {

constexpr int kThreads = 256;

struct AxisDecode {
    float val;     // decoded value in [0, 2^n - 1]
    int idx;       // channel (within the axis) that carries the gradient, or -1
    float dval;    // d val / d logit[idx]
};

// floatbits.py:130-160 for one axis of one pixel
__device__ __forceinline__ AxisDecode decode_with_gt(const float* lg, const unsigned char* gt, size_t stride, int n, bool in_msk,
                                                     int black_factor) {
    float out_val = 0.f, correct = 0.f;
    int idx = n - 1;
    bool found = false;
    float x_idx = 0.f, sgn_idx = 1.f;
    bool prev = false;
    for (int k = 0; k < n; ++k) {
        const bool b = gt[k * stride] != 0;
        float sgn = (k >= 1 && prev) ? -1.f : 1.f;   // logits_msk[1:] = -1 where the previous gt bit is set
        if (k < 2) sgn *= (float)black_factor;       // logits_msk[0:2] *= black_factor
        const float x = lg[k * stride] * sgn;
        const bool pred = x > 0.f;
        const float w = (float)(1 << (n - 1 - k));
        out_val += pred ? w : 0.f;
        const bool err = (pred != b) || (k == n - 1);
        if (err && !found) {  // first erroneous bit (torch.argmax returns the first maximal index)
            found = true;
            idx = k; x_idx = x; sgn_idx = sgn;
        } else {
            correct += b ? w : 0.f;  // gt bits with the MSB-error bit cleared
        }
        prev = b;
    }
    AxisDecode o;
    const float w = (float)(1 << (n - 1 - idx));
    const float s = 1.f / (1.f + __expf(-x_idx));
    if (in_msk) {
        o.val = correct + s * w;
        o.idx = idx;
        o.dval = w * s * (1.f - s) * sgn_idx;
    } else {
        o.val = out_val;
        o.idx = -1;
        o.dval = 0.f;
    }
    return o;
}

// floatbits.py:194-223 for one axis of one pixel
__device__ __forceinline__ float decode_gray(const float* lg, size_t stride, int n, bool black) {
    unsigned code = 0;
    float last = 0.f;
    for (int k = 0; k < n; ++k) {
        last = lg[k * stride];
        bool bit = last > 0.f;
        if (black && k < 2) bit = !bit;
        code = (code << 1) | (bit ? 1u : 0u);
    }
    unsigned v = code;  // inverse Gray code (the reference's lut[src] = dst with src = dst ^ (dst >> 1))
    for (unsigned sh = 1; sh < 32; sh <<= 1) v ^= v >> sh;
    const float lsb_factor = (v & 2u) ? -1.f : 1.f;
    return (float)(v & ~1u) + 1.f / (1.f + __expf(-last * lsb_factor));
}

__global__ __launch_bounds__(kThreads) void lc_bits_decode_gt_fwd_kernel(const BitsParams p) {
    const int Wn = (p.W - p.left + p.sample - 1) / p.sample;
    const size_t HW = (size_t)p.H * p.W;
    const size_t total = (size_t)p.B * p.N;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (size_t)gridDim.x * kThreads) {
        const int b = (int)(i / p.N), n = (int)(i % p.N);
        const int y = p.top + (n / Wn) * p.sample, x = p.left + (n % Wn) * p.sample;
        const size_t px = (size_t)y * p.W + x;
        const bool in_msk = p.gt_msk ? p.gt_msk[(size_t)b * HW + px] != 0 : true;
        int c0 = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int nb = p.bits[a];
            const size_t base = ((size_t)b * p.C + c0) * HW + px;
            const AxisDecode d = decode_with_gt(p.logits + base, p.gt_bits + base, HW, nb, in_msk, p.black_factor);
            p.out[i * 3 + a] = d.val / ((float)((1 << nb) - 1) * 0.5f) - 1.f;
            c0 += nb;
        }
    }
}

__global__ __launch_bounds__(kThreads) void lc_bits_decode_gt_bwd_kernel(const BitsParams p) {
    // one thread per pixel of the FULL map: writes every channel (zeros where no gradient arrives)
    const int Wn = (p.W - p.left + p.sample - 1) / p.sample;
    const size_t HW = (size_t)p.H * p.W;
    const size_t total = (size_t)p.B * HW;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (size_t)gridDim.x * kThreads) {
        const int b = (int)(i / HW);
        const size_t px = i % HW;
        const int y = (int)(px / p.W), x = (int)(px % p.W);
        const int dy = y - p.top, dx = x - p.left;
        const bool hit = dy >= 0 && dx >= 0 && (dy % p.sample) == 0 && (dx % p.sample) == 0;
        const bool in_msk = p.gt_msk ? p.gt_msk[(size_t)b * HW + px] != 0 : true;
        const size_t n = hit ? (size_t)(dy / p.sample) * Wn + dx / p.sample : 0;
        int c0 = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int nb = p.bits[a];
            const size_t base = ((size_t)b * p.C + c0) * HW + px;
            int idx = -1;
            float g = 0.f;
            if (hit && in_msk) {
                const AxisDecode d = decode_with_gt(p.logits + base, p.gt_bits + base, HW, nb, true, p.black_factor);
                idx = d.idx;
                g = p.g_out[((size_t)b * p.N + n) * 3 + a] * d.dval / ((float)((1 << nb) - 1) * 0.5f);
            }
            for (int k = 0; k < nb; ++k) p.d_logits[base + k * HW] = (k == idx) ? g : 0.f;
            c0 += nb;
        }
    }
}

__global__ __launch_bounds__(kThreads) void lc_bits_decode_kernel(const BitsParams p) {
    const size_t HW = (size_t)p.H * p.W;
    const size_t total = (size_t)p.B * HW;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (size_t)gridDim.x * kThreads) {
        const int b = (int)(i / HW);
        const size_t px = i % HW;
        int c0 = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int nb = p.bits[a];
            const float v = decode_gray(p.logits + ((size_t)b * p.C + c0) * HW + px, HW, nb, p.black_factor < 0);
            p.out[i * 3 + a] = v / ((float)((1 << nb) - 1) * 0.5f) - 1.f;
            c0 += nb;
        }
    }
}

int grid_for(size_t total) {
    size_t g = (total + kThreads - 1) / kThreads;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));  // cap + grid-stride (cdna_hip_programming.md Guideline 11)
}

}  // namespace

int launch_bits_decode_gt_fwd(const BitsParams& p, hipStream_t stream) {
    if (p.B <= 0 || p.N <= 0) return 0;
    hipLaunchKernelGGL(lc_bits_decode_gt_fwd_kernel, dim3(grid_for((size_t)p.B * p.N)), dim3(kThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int launch_bits_decode_gt_bwd(const BitsParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    hipLaunchKernelGGL(lc_bits_decode_gt_bwd_kernel, dim3(grid_for((size_t)p.B * p.H * p.W)), dim3(kThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int launch_bits_decode(const BitsParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    hipLaunchKernelGGL(lc_bits_decode_kernel, dim3(grid_for((size_t)p.B * p.H * p.W)), dim3(kThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc

// Pointwise conv for TINY position counts (fp32: exact-fp32 MFMA; bf16): the prototype heads of the image configs (BASELINE config 1: 8 images
// x 49 positions = 392 rows, 512 channels; Video_XProtoNet.py:27-62 / XProtoNet.py:17-41 add-on and occurrence-module convs).
//
// gemm_pw.hip gives such a layer 128 x 128 output tiles: 4 x 4 = 16 blocks for a 392 x 512 output, each walking K = 512 behind a
// double-buffered LDS pipeline -- ~45 us per conv, five convs per head (254 us for 0.87 MB: profiles/r02b_head_bench.jsonl).  With so few
// rows the only parallelism is across output tiles: here ONE WAVE owns a 32-channel x 32-position tile and walks K straight from global
// memory (both operands are K-contiguous rows: a lane's 16 bytes are 4 consecutive k of its row; eight k-steps of loads are requested
// before their MFMAs), 13 x 16 = 208 independent waves for the same layer.  Epilogue: lane swap -> 8 consecutive channels per lane ->
// scale / bias / residual / activation -> two 16-byte stores.
#include "common.h"

namespace pasn {

constexpr int PT_UNROLL = 8;  // k-steps (of 8) requested together

template <typename T>
__global__ __launch_bounds__(64) void pwconv_tiny_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ scale,
                                                         const float* __restrict__ bias, const T* __restrict__ res, T* __restrict__ y, int M,
                                                         int Cin_p, int kc, int Cout, int Cout_p, int w_rows, int act) {
    constexpr int CH = Traits<T>::CH, KSTEP = Traits<T>::KSTEP;  // 4 / 8 (fp32), 8 / 16 (bf16): a lane's 16 bytes, k per MFMA step
    using frag = typename Traits<T>::frag;
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int row = min(m0 + c, M - 1);                    // clamped: rows beyond M are computed and never stored
    const int wr = min(n0 + c, w_rows - 1);
    const T* xp = x + (long)row * Cin_p + CH * h;
    const T* wp = w + (long)wr * kc + CH * h;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const int nsteps = kc / KSTEP;                          // host: kc == Cin_p (no K padding to read past the row)
    int ks = 0;
    for (; ks + PT_UNROLL <= nsteps; ks += PT_UNROLL) {
        frag a[PT_UNROLL], b[PT_UNROLL];
#pragma unroll
        for (int u = 0; u < PT_UNROLL; ++u) {
            a[u] = load_frag<T>(wp + (ks + u) * KSTEP);
            b[u] = load_frag<T>(xp + (ks + u) * KSTEP);
        }
#pragma unroll
        for (int u = 0; u < PT_UNROLL; ++u) mma32(acc, a[u], b[u]);
    }
    for (; ks < nsteps; ++ks) mma32(acc, load_frag<T>(wp + ks * KSTEP), load_frag<T>(xp + ks * KSTEP));
    // accumulator: column = position (lane & 31), rows = channels; the half-wave exchange gives lanes < 32 channels 16 pr .. + 7 and lanes >= 32
    // channels 16 pr + 8 .. + 15 of their position
    const bool rowok = m0 + c < M;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const int ch = n0 + 16 * pr + 8 * h;
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * pr + q]), __float_as_uint(acc[8 * pr + 4 + q]), false, false);
            v[q] = __uint_as_float(sw[0]);
            v[4 + q] = __uint_as_float(sw[1]);
        }
        if (ch < Cout_p) {
            float sc[8], bs[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sc[e] = 1.0f;
                bs[e] = 0.0f;
            }
            if (scale) load8(scale + ch, sc);
            if (bias) load8(bias + ch, bs);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + bs[e];
            if (res && rowok) {
                float r8[8];
                load8(res + (long)(m0 + c) * Cout_p + ch, r8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r8[e];
            }
            act_vec(v, act);
            mask_tail(v, Cout - ch);
            if (rowok) store8(y + (long)(m0 + c) * Cout_p + ch, v);
        }
    }
}

// pointwise conv (stride 1, plain [rows][kc] weights, no input transform) on few positions
bool pw_tiny_applicable(const pasn_conv_desc& d, int dtype, bool has_gate) {
    if (const char* e = tune("PASN_NO_PWTINY"))
        if (e[0] == '1') return false;
    if ((dtype != PASN_F32 && dtype != PASN_BF16) || has_gate || d.in_swish || d.w_frag != 0) return false;
    if (d.kt != 1 || d.kh != 1 || d.kw != 1 || d.pt || d.ph || d.pw || d.st != 1 || d.sh != 1 || d.sw != 1) return false;
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    // few rows and a real K: below ~8 k rows the tiled kernels leave most CUs idle (fp32: gemm_pw's 128-row tiles; bf16: 64-row tiles x
    // 128 channels).  bf16 only where the head's widths make it matter (>= 256 input channels)
    return M <= 8192 && d.Cin_p % 8 == 0 && d.Cin_p >= (dtype == PASN_F32 ? 64 : 256) && d.w_kc == d.Cin_p && d.Cout_p % 8 == 0 &&
           d.w_rows >= ((d.Cout_p + 31) / 32) * 32;
}

int launch_pw_tiny(const void* x, const void* w, const float* scale, const float* bias, const void* res, void* y, const pasn_conv_desc& d,
                   int dtype, hipStream_t s) {
    const int M = (int)((long)d.N * d.To * d.Ho * d.Wo);
    const dim3 grid(ceil_div(M, 32), ceil_div(d.Cout_p, 32)), block(64);
    if (dtype == PASN_F32)
        hipLaunchKernelGGL(pwconv_tiny_kernel<float>, grid, block, 0, s, (const float*)x, (const float*)w, scale, bias, (const float*)res, (float*)y, M,
                           d.Cin_p, d.w_kc, d.Cout, d.Cout_p, d.w_rows, d.act);
    else
        hipLaunchKernelGGL(pwconv_tiny_kernel<__bf16>, grid, block, 0, s, (const __bf16*)x, (const __bf16*)w, scale, bias, (const __bf16*)res,
                           (__bf16*)y, M, d.Cin_p, d.w_kc, d.Cout, d.Cout_p, d.w_rows, d.act);
    return check_launch("pwconv_tiny_kernel");
}

}  // namespace pasn

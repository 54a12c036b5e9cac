// Pointwise conv for TINY position counts, fp32 (exact-fp32 MFMA): the prototype heads of the image configs (BASELINE config 1: 8 images
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

__global__ __launch_bounds__(64) void pwconv_tiny_f32_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                                                             const float* __restrict__ bias, const float* __restrict__ res, float* __restrict__ y,
                                                             int M, int Cin_p, int kc, int Cout, int Cout_p, int w_rows, int act) {
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int row = min(m0 + c, M - 1);                    // clamped: rows beyond M are computed and never stored
    const int wr = min(n0 + c, w_rows - 1);
    const float* xp = x + (long)row * Cin_p + 4 * h;
    const float* wp = w + (long)wr * kc + 4 * h;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const int nsteps = Cin_p / 8;                           // kc == Cin_p rounded up to 8 == Cin_p (channel strides are multiples of 8)
    int ks = 0;
    for (; ks + PT_UNROLL <= nsteps; ks += PT_UNROLL) {
        f32x4 a[PT_UNROLL], b[PT_UNROLL];
#pragma unroll
        for (int u = 0; u < PT_UNROLL; ++u) {
            a[u] = *reinterpret_cast<const f32x4*>(wp + (ks + u) * 8);
            b[u] = *reinterpret_cast<const f32x4*>(xp + (ks + u) * 8);
        }
#pragma unroll
        for (int u = 0; u < PT_UNROLL; ++u) mma32(acc, a[u], b[u]);
    }
    for (; ks < nsteps; ++ks) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(wp + ks * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(xp + ks * 8);
        mma32(acc, a, b);
    }
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
            const int chc = min(ch, w_rows - 8);
            if (scale) load8(scale + chc, sc);
            if (bias) load8(bias + chc, bs);
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

// fp32 pointwise conv (stride 1, plain [rows][kc] weights, no input transform) on few positions
bool pw_tiny_applicable(const pasn_conv_desc& d, int dtype, bool has_gate) {
    if (const char* e = getenv("PASN_NO_PWTINY"))
        if (e[0] == '1') return false;
    if (dtype != PASN_F32 || has_gate || d.in_swish || d.w_frag != 0) return false;
    if (d.kt != 1 || d.kh != 1 || d.kw != 1 || d.pt || d.ph || d.pw || d.st != 1 || d.sh != 1 || d.sw != 1) return false;
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    // few rows and a real K: below ~8 k rows gemm_pw's 128-row tiles leave most CUs idle
    return M <= 8192 && d.Cin_p % 8 == 0 && d.Cin_p >= 64 && d.w_kc == d.Cin_p && d.Cout_p % 8 == 0 && d.w_rows >= ((d.Cout_p + 31) / 32) * 32;
}

int launch_pw_tiny(const void* x, const void* w, const float* scale, const float* bias, const void* res, void* y, const pasn_conv_desc& d,
                   hipStream_t s) {
    const int M = (int)((long)d.N * d.To * d.Ho * d.Wo);
    const dim3 grid(ceil_div(M, 32), ceil_div(d.Cout_p, 32)), block(64);
    hipLaunchKernelGGL(pwconv_tiny_f32_kernel, grid, block, 0, s, (const float*)x, (const float*)w, scale, bias, (const float*)res, (float*)y, M,
                       d.Cin_p, d.w_kc, d.Cout, d.Cout_p, d.w_rows, d.act);
    return check_launch("pwconv_tiny_f32_kernel");
}

}  // namespace pasn

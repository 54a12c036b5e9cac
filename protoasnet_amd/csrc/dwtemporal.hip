// Depthwise TEMPORAL conv (kt,1,1), kt = 3 or 5, stride 1, "same" padding, + scale / bias + activation: the X3D stem's conv_t where it runs as a
// launch of its own (the train-mode forward, and its input gradient = the same conv with reversed taps: train.py) -- round 5.
//
// The generic strip kernel (conv.hip) moves this layer at 2.7 TB/s (462 / 433 us for 1.23 GB at 64 x 16 x 112 x 112 x 24).  There is nothing to
// tile: a thread owns (position, 8 channels) and MARCHES ALONG T with the last kt frames of its 8 channels in registers -- one 16-byte load, kt x 8
// FMAs and one 16-byte store per frame, consecutive threads on consecutive 16-byte pieces of a frame.  Frame t + pad is requested before frame
// t's outputs are formed.
#include "common.h"

namespace pasn {

template <typename T, int KT>
__global__ __launch_bounds__(256) void dwconv_t_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                                                       const float* __restrict__ bias, T* __restrict__ y, long per_clip, int N, int Tn, int Cp,
                                                       int C, int act) {
    constexpr int PAD = KT / 2;
    const int CG = Cp >> 3;
    const long total = (long)N * per_clip;  // per_clip = H W CG (16-byte pieces of a frame)
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long n = idx / per_clip, piece = idx - n * per_clip;
    const int cg = (int)(piece % CG);
    const long fstride = per_clip * 8;  // elements per frame
    const T* xp = x + (n * Tn) * fstride + piece * 8;
    T* yp = y + (n * Tn) * fstride + piece * 8;
    float wv[KT][8], sc[8], bs[8];
#pragma unroll
    for (int k = 0; k < KT; ++k) load8(w + (long)k * Cp + cg * 8, wv[k]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = 1.0f;
        bs[j] = 0.0f;
    }
    if (scale) load8(scale + cg * 8, sc);
    if (bias) load8(bias + cg * 8, bs);
    const int nvalid = C - cg * 8;  // channels of this piece that exist (the padded ones are stored as zeros)
    float r[KT][8];                 // r[k] = frame t + k - PAD
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) r[k][j] = 0.0f;
#pragma unroll
    for (int k = PAD; k < KT; ++k)
        if (k - PAD < Tn) load8(xp + (long)(k - PAD) * fstride, r[k]);
    for (int t = 0; t < Tn; ++t) {
        float nx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) nx[j] = 0.0f;
        if (t + PAD + 1 < Tn) load8(xp + (long)(t + PAD + 1) * fstride, nx);  // the frame that enters the window at the next step
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = 0.0f;
#pragma unroll
            for (int k = 0; k < KT; ++k) a = fmaf(wv[k][j], r[k][j], a);
            v[j] = a * sc[j] + bs[j];
        }
        act_vec(v, act);
        if (nvalid < 8) mask_tail(v, nvalid);
        store8(yp + (long)t * fstride, v);
#pragma unroll
        for (int k = 0; k + 1 < KT; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) r[k][j] = r[k + 1][j];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[KT - 1][j] = nx[j];
    }
}

bool dw_temporal_applicable(const pasn_conv_desc& d, int dtype) {
    if (dtype != PASN_BF16 && dtype != PASN_F32) return false;
    if (tune("PASN_DW_TEMPORAL") && tune("PASN_DW_TEMPORAL")[0] == '0') return false;
    return (d.kt == 3 || d.kt == 5) && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == d.kt / 2 && d.ph == 0 && d.pw == 0 &&
           d.To == d.Ti && d.Ho == d.Hi && d.Wo == d.Wi && d.Cin_p == d.Cout_p && d.Cin == d.Cout && d.Cout_p % 8 == 0;
}

int launch_dw_temporal(const void* x, const float* w, const float* scale, const float* bias, void* y, const pasn_conv_desc& d, int dtype, hipStream_t s) {
    const long per_clip = (long)d.Ho * d.Wo * (d.Cout_p / 8);
    const long total = (long)d.N * per_clip;
    const dim3 grid((unsigned)ceil_div(total, 256L)), block(256);
#define PASN_DT_(T_, KT_) \
    hipLaunchKernelGGL((dwconv_t_kernel<T_, KT_>), grid, block, 0, s, (const T_*)x, w, scale, bias, (T_*)y, per_clip, (int)d.N, (int)d.Ti, (int)d.Cout_p, (int)d.Cout, (int)d.act)
    if (dtype == PASN_BF16) {
        if (d.kt == 5) PASN_DT_(__bf16, 5);
        else PASN_DT_(__bf16, 3);
    } else {
        if (d.kt == 5) PASN_DT_(float, 5);
        else PASN_DT_(float, 3);
    }
#undef PASN_DT_
    return check_launch("dwconv_t_kernel");
}

}  // namespace pasn

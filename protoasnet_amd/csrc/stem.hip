// Fused X3D stem: (1,3,3) stride-(1,2,2) conv 3 -> C  ->  depthwise (5,1,1) temporal conv  ->  BN -> ReLU, one launch.
//
// Unfused, the C-channel tensor between the two convs (308 MB at the benchmark shape) is written once and then read
// by the temporal conv -- whose 5 taps are 5 different frames 600 KB apart, so L2 does not catch the reuse: PMC shows
// 1.26 GB of HBM traffic for 0.62 GB of algorithmic bytes (profiles/pmc_traffic.json, dwconv3d_strip_kernel<bf16,7,1,1>).
// Here a thread owns one output position (n, ho, wo) x all C channels and MARCHES ALONG T: it computes the spatial conv
// of frame t (3*3*3 planar taps, weights from the scalar cache), keeps the last five results in a register ring, and emits
// output frame t-2.  Traffic = clip in + stem out, nothing else.  The ring holds values rounded to the activation dtype,
// so the result is bit-identical to the two unfused launches.
#include "common.h"

namespace pasn {

// CIN = 3: the reference's clip.  CIN = 1: a grey clip (the three channels of an echo clip are identical, as_dataloader.py:168-170), w_xy
// holds the 9 taps summed over the input channels: a third of the input bytes and of the spatial FMAs.  x' = x * in_a + in_b is applied
// to every loaded value (device-side normalisation, as_dataloader.py:180-182; 1, 0 = none); padding pads the normalised tensor.
template <typename TIN, typename T, int COP, int CIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void x3d_stem_kernel(const TIN* __restrict__ x, const float* __restrict__ wxy,
                                                       const float* __restrict__ wt, const float* __restrict__ scale,
                                                       const float* __restrict__ bias, T* __restrict__ y, pasn_conv_desc d,
                                                       float in_a, float in_b) {
    // The 27 x COP spatial weights are wave-uniform with compile-time offsets: read with s_load through the scalar cache
    // and fed to v_pk_fma as SGPR operands -- no LDS traffic, no vector registers.  The temporal weights + scale + bias
    // (7 x COP floats) come from LDS as broadcast reads instead: as scalars they did not fit next to the spatial ones, and
    // hipcc spilled them through VGPR lanes (252 v_readlane/v_writelane per frame for 120 FMAs).
    __shared__ __attribute__((aligned(16))) float tl[7 * COP];  // [5 temporal taps | scale | bias][COP]
    for (int i = threadIdx.x; i < 7 * COP; i += 256) tl[i] = i < 5 * COP ? wt[i] : (i < 6 * COP ? scale[i - 5 * COP] : bias[i - 6 * COP]);
    __syncthreads();
    const long P = (long)d.N * d.Ho * d.Wo;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int wo = (int)(p % d.Wo);
    const int ho = (int)((p / d.Wo) % d.Ho);
    const int n = (int)(p / ((long)d.Wo * d.Ho));
    const int T_ = d.Ti, Hi = d.Hi, Wi = d.Wi;
    const long plane = (long)Hi * Wi;
    // the 9 spatial taps of this position: offsets inside a frame, -1 where the window leaves the image
    int off[9];
#pragma unroll
    for (int kr = 0; kr < 3; ++kr)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int hi = ho * 2 - 1 + kr, wi = wo * 2 - 1 + ks;
            off[kr * 3 + ks] = (hi >= 0 && hi < Hi && wi >= 0 && wi < Wi) ? hi * Wi + wi : -1;
        }
    // spatial-conv results of frames t-4 .. t, slot = frame % 5, held IN THE ACTIVATION DTYPE as 8-wide vectors
    // (bf16: 5 x 12 registers instead of 5 x 24) -- exactly the values the unfused path would have stored
    typedef T RV __attribute__((ext_vector_type(8)));
    RV ring[5][COP / 8];
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int c = 0; c < COP; ++c) ring[u][c >> 3][c & 7] = (T)0.0f;

#pragma unroll 1
    for (int t = 0; t < T_ + 2; ++t) {
        // ---- spatial conv of frame t (zeros beyond the clip: they flush the last two outputs) ----
        // an opaque zero in an SGPR, renewed per frame: keeps the 768 weight s_loads INSIDE the loop (hoisted, they
        // spill ~250 SGPRs)
        int zo = 0;
        asm volatile("" : "+s"(zo));
        const float* wq = wxy + zo;
        float acc[COP];
#pragma unroll
        for (int c = 0; c < COP; ++c) acc[c] = 0.0f;
        if (t < T_) {
            float xv[9 * CIN];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const TIN* xp = x + (((long)n * CIN + ci) * T_ + t) * plane;
#pragma unroll
                for (int q = 0; q < 9; ++q) {  // unconditional load (clamped), select after: no per-tap branches
                    const float v = fmaf((float)xp[off[q] >= 0 ? off[q] : 0], in_a, in_b);
                    xv[ci * 9 + q] = off[q] >= 0 ? v : 0.0f;
                }
            }
            // software pipeline over taps: the s_loads of tap q+1 are issued before the FMAs of tap q, and scheduling
            // barriers stop the scheduler from front-loading all 27 taps (which spills ~250 SGPRs)
            float wc[COP], wn[COP];
#pragma unroll
            for (int c = 0; c < COP; ++c) wc[c] = wq[c];
#pragma unroll
            for (int q = 0; q < 9 * CIN; ++q) {
                if (q + 1 < 9 * CIN) {
#pragma unroll
                    for (int c = 0; c < COP; ++c) wn[c] = wq[(q + 1) * COP + c];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < COP; ++c) acc[c] = fmaf(xv[q], wc[c], acc[c]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < COP; ++c) wc[c] = wn[c];
            }
        }
        // ---- rotate the ring (oldest frame out), newest in slot 4, rounded as the unfused path stores it ----
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < COP / 8; ++j) ring[u][j] = ring[u + 1][j];
#pragma unroll
        for (int c = 0; c < COP; ++c) ring[4][c >> 3][c & 7] = (T)acc[c];
        // ---- temporal conv: output frame to = t - 2 from frames t-4 .. t = slots 0 .. 4 ----
        const int to = t - 2;
        if (to >= 0) {
            int zv = 0;  // opaque per-frame zero in a VGPR: keeps the LDS reads inside the loop (hoisted: 168 registers)
            asm volatile("" : "+v"(zv));
            const float* tw = tl + zv;
            float o[COP];
#pragma unroll
            for (int c = 0; c < COP; ++c) o[c] = 0.0f;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
#pragma unroll
                for (int c = 0; c < COP; c += 4) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(tw + k * COP + c);  // wave-uniform address: a broadcast
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[c + j] = fmaf((float)ring[k][(c + j) >> 3][(c + j) & 7], wv[j], o[c + j]);
                }
            }
            const float* sc = tw + 5 * COP;
            const float* bi = tw + 6 * COP;
            T* yp = y + ((((long)n * d.To + to) * d.Ho + ho) * d.Wo + wo) * COP;
#pragma unroll
            for (int c = 0; c < COP; c += 8) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(o[c + j] * sc[c + j] + bi[c + j], 0.0f);
                mask_tail(v, d.Cout - c);
                store8(yp + c, v);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

}  // namespace pasn

using namespace pasn;

extern "C" int pasn_x3d_stem_supported(const pasn_conv_desc* d) {
    if (!d) return 0;
    if (const char* e = tune("PASN_NO_STEM"))
        if (e[0] == '1') return 0;
    return (d->Cin == 3 || d->Cin == 1) && d->kt == 1 && d->kh == 3 && d->kw == 3 && d->st == 1 && d->sh == 2 && d->sw == 2 && d->pt == 0 &&
           d->ph == 1 && d->pw == 1 && d->To == d->Ti && d->Cout_p == 24 && d->Cout <= 24;
}

static int stem_dispatch(const void* x, const float* w_xy, const float* w_t, const float* scale, const float* bias, void* y,
                         const pasn_conv_desc* d, int in_dtype, int out_dtype, float in_a, float in_b, void* stream) {
    PASN_REQUIRE(x && w_xy && w_t && scale && bias && y && d, "null pointer");
    PASN_REQUIRE(pasn_x3d_stem_supported(d), "geometry is not the X3D stem ((1,3,3) s(1,2,2) p(0,1,1), 24 channels)");
    const long P = (long)d->N * d->Ho * d->Wo;
    const dim3 grid(ceil_div(P, 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define PASN_ST(TIN, T, CIN) \
    hipLaunchKernelGGL((x3d_stem_kernel<TIN, T, 24, CIN>), grid, block, 0, s, (const TIN*)x, w_xy, w_t, scale, bias, (T*)y, *d, in_a, in_b)
#define PASN_ST2(TIN, T)         \
    do {                         \
        if (d->Cin == 3) PASN_ST(TIN, T, 3); \
        else PASN_ST(TIN, T, 1); \
    } while (0)
    if (in_dtype == PASN_F32 && out_dtype == PASN_F32) PASN_ST2(float, float);
    else if (in_dtype == PASN_F32 && out_dtype == PASN_BF16) PASN_ST2(float, __bf16);
    else if (in_dtype == PASN_BF16 && out_dtype == PASN_BF16) PASN_ST2(__bf16, __bf16);
    else if (in_dtype == PASN_BF16 && out_dtype == PASN_F32) PASN_ST2(__bf16, float);
    else if (in_dtype == PASN_U8 && out_dtype == PASN_BF16 && d->Cin == 1) PASN_ST(unsigned char, __bf16, 1);
    else if (in_dtype == PASN_U8 && out_dtype == PASN_F32 && d->Cin == 1) PASN_ST(unsigned char, float, 1);
    else {
        set_error("pasn_x3d_stem_fwd: unknown dtype");
        return PASN_ERR_ARG;
    }
#undef PASN_ST2
#undef PASN_ST
    return check_launch("x3d_stem_kernel");
}

extern "C" int pasn_x3d_stem_fwd(const void* x, const float* w_xy, const float* w_t, const float* scale, const float* bias,
                                 void* y, const pasn_conv_desc* d, int in_dtype, int out_dtype, void* stream) {
    PASN_REQUIRE(d && d->Cin == 3, "pasn_x3d_stem_fwd reads 3 planar channels (pasn_x3d_stem_gray_fwd reads one)");
    return stem_dispatch(x, w_xy, w_t, scale, bias, y, d, in_dtype, out_dtype, 1.0f, 0.0f, stream);
}

extern "C" int pasn_x3d_stem_gray_fwd(const void* x, const float* w_xy, const float* w_t, const float* scale, const float* bias,
                                      void* y, const pasn_conv_desc* d, int in_dtype, int out_dtype, float in_a, float in_b,
                                      void* stream) {
    PASN_REQUIRE(d && d->Cin == 1, "pasn_x3d_stem_gray_fwd reads ONE planar channel");
    return stem_dispatch(x, w_xy, w_t, scale, bias, y, d, in_dtype, out_dtype, in_a, in_b, stream);
}

// ---- the same stem on the matrix cores (stem_mfma.hip): bf16 out, W % 4 == 0 ----
extern "C" int pasn_x3d_stem_mfma_supported(const pasn_conv_desc* d, int in_dtype, int out_dtype) {
    if (!d || (in_dtype != PASN_F32 && in_dtype != PASN_BF16 && in_dtype != PASN_U8)) return 0;
    return pasn::x3d_stem_mfma_supported(*d, out_dtype);
}

extern "C" int pasn_x3d_stem_mfma_fwd(const void* x, const void* wq, const float* scale, const float* bias, void* y, const pasn_conv_desc* d,
                                      int in_dtype, float in_a, float in_b, void* stream) {
    using namespace pasn;
    PASN_REQUIRE(x && wq && scale && bias && y && d, "null argument");
    PASN_REQUIRE(x3d_stem_mfma_supported(*d, PASN_BF16), "pasn_x3d_stem_mfma_fwd: layer not covered (ask pasn_x3d_stem_mfma_supported)");
    return launch_x3d_stem_mfma(x, wq, scale, bias, y, *d, in_dtype, in_a, in_b, (hipStream_t)stream);
}

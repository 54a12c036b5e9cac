// Epilogue shared by the implicit-GEMM kernels (igemm.hip, igemm_halo.hip): accumulators (position on the lane) -> scale / bias -> a
// wave-private LDS image of 32 positions x BN channels -> residual + activation + whole-row 16-byte stores.
//
// Measured on the 64 -> 144 (1,3,3) layer (8 x 32 x 56 x 56): the first version of this epilogue was 108 of the launch's 278 us.  It read
// scale / bias with one dependent global load per 4 channels (80 L2 round trips per wave, one after the other) and ran the row loop as
// load -> add -> store chains.  Now scale / bias sit in LDS (staged once per block, before the main loop) and the row loop issues the
// image reads and residual loads of NT iterations before it touches any of them.
#pragma once
#include "common.h"

namespace pasn {

// [2][BN] floats: scale (1 where absent / beyond the weight rows), then bias (0 ...).  Call before a block barrier.
template <int BN>
__device__ __forceinline__ void igemm_stage_scale_bias(float* scb, const float* __restrict__ scale, const float* __restrict__ bias, int n0,
                                                       int w_rows, int tid) {
    for (int i = tid; i < BN; i += 256) {
        const int n = n0 + i;
        const bool ok = n < w_rows;
        scb[i] = (scale && ok) ? scale[n] : 1.0f;
        scb[BN + i] = (bias && ok) ? bias[n] : 0.0f;
    }
}

// Tile j of this wave: rows mbase .. mbase + nvalid - 1 of y (nvalid >= 1), channels n0 .. n0 + 8 * cgs - 1.
template <int NT, int MT>
__device__ __forceinline__ void igemm_epilogue_tile(const f32x16 (&acc)[NT][MT], int j, __bf16* img, const float* scb,
                                                    const __bf16* __restrict__ res, __bf16* __restrict__ y, long mbase, int nvalid, int n0,
                                                    int cgs, const pasn_conv_desc& d, int lane) {
    constexpr int BN = NT * 32, OROW = BN + 8;
    const int c = lane & 31, h = lane >> 5;
    const int Cout_p = d.Cout_p;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = i * 32 + 8 * g + 4 * h;  // channel inside the block tile
            const f32x4 sc = *reinterpret_cast<const f32x4*>(scb + col);
            const f32x4 bs = *reinterpret_cast<const f32x4*>(scb + BN + col);
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = acc[i][j][4 * g + r] * sc[r] + bs[r];
            store4(img + (size_t)c * OROW + col, o);
        }
    // wave-private image: no block barrier, only this wave's LDS writes must have landed (the compiler orders LDS ops of a wave)
    const float inv = 1.0f / (float)cgs;  // row = p / cgs through a float multiply: exact for p < 1024, cgs <= 20
    const int total = 32 * cgs;
#pragma unroll
    for (int b0 = 0; b0 < 2; ++b0) {  // 2 * NT row-loop iterations cover 32 rows x BN / 8 channel groups: two batches of NT
        bf16x8 vi[NT], vr[NT];
        long dst[NT];
        int nn[NT];
        bool ok[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int p = lane + 64 * (b0 * NT + u);
            int row = (int)(((float)p + 0.5f) * inv);
            int cg = p - row * cgs;
            ok[u] = p < total && row < nvalid;
            row = ok[u] ? row : 0;
            cg = ok[u] ? cg : 0;
            nn[u] = n0 + cg * 8;
            dst[u] = (mbase + row) * Cout_p + nn[u];
            vi[u] = *reinterpret_cast<const bf16x8*>(img + (size_t)row * OROW + cg * 8);
            if (res) vr[u] = *reinterpret_cast<const bf16x8*>(res + dst[u]);
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)vi[u][e];
            if (res) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)vr[u][e];
            }
            act_vec(v, d.act);
            mask_tail(v, d.Cout - nn[u]);
            if (ok[u]) store8(y + dst[u], v);
        }
    }
}

}  // namespace pasn

// Two chained 1x1x1 convs on the same 64-position tile, bf16: X3D block i's project conv (conv_c: BN + residual + ReLU,
// optional fused SE-gate / Swish input transform) followed by block i+1's expand conv (conv_a: BN + ReLU).
//
// Why: in stage 4 (14x14) these are 21 launches of ~21 us each for 60-80 MB -- small uniform blocks running 3-4 "rounds",
// each with its own load -> barrier -> MFMA -> store latency chain.  The block output Y1 has to reach HBM anyway (it is
// the next residual), but the expand conv can take it from LDS: one launch, one latency chain, Y1 never re-read.
//
// Structure = pwconv_xtile.hip (whole-K X tile in LDS, one 32-channel tile per wave with its weights in registers, all
// independent loads issued up front, branch-free K loops) plus:
//   * after stage 1's K loops a block barrier frees the X tile; the epilogue's finished rows (post residual + ReLU, as
//     bf16 -- exactly what goes to HBM) are ALSO written into the same LDS region as stage 2's input tile;
//   * stage 2: every wave computes up to two 32-channel tiles of the expand conv (weights requested right after stage
//     1's K loop, so they land under its epilogue), scale / bias from LDS, same coalescing epilogue.
// Requirements (checked on the host): stage-1 output channels fit one block (<= 4 tiles) and fill stage 2's K extent
// exactly (KS2 * 16 == tiles * 32), stage-2 output <= 8 tiles, both weights fragment-major.
#include "common.h"

namespace pasn {

constexpr int XP_BM = 64;
constexpr int XP_SROW = 36;

template <int KS1, int KS2, bool XF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KS1 <= 8 ? 3 : 2)))
void pwconv_xpair_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ w1, const float* __restrict__ s1,
                         const float* __restrict__ b1, const __bf16* __restrict__ res, const float* __restrict__ gate,
                         __bf16* __restrict__ y1, const __bf16* __restrict__ w2, const float* __restrict__ s2,
                         const float* __restrict__ b2, __bf16* __restrict__ y2, long M, int S, int Cin_p, int C1, int C1_p,
                         int w1_kc, int act1, int in_swish, int C2, int C2_p, int w2_kc, int act2) {
    constexpr int CH = 8, KSTEP = 16;
    constexpr int PPR = KS1 * 2, NP = PPR * XP_BM / 256;
    constexpr int KP1 = KS1 * KSTEP + CH, KP2 = KS2 * KSTEP + CH;
    static_assert(KS1 % 2 == 0 && KS2 % 2 == 0 && KP2 <= KP1, "tile geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* xs = reinterpret_cast<__bf16*>(smem);  // stage 1: [64][KP1]; stage 2: [64][KP2] in the same place
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    float* fbase = reinterpret_cast<float*>(smem + (size_t)XP_BM * KP1 * 2);
    float* scr = fbase + wave * 32 * XP_SROW;   // [4 waves][32][36] epilogue images
    float* gl = fbase + 4 * 32 * XP_SROW;       // [2 clips][w1_kc] gate rows (XF only)
    float* sb2 = gl + 2 * w1_kc;                // [2][256] scale | bias of stage 2
    const int ct1 = wave, ctiles1 = (C1_p + 31) / 32, ctiles2 = (C2_p + 31) / 32;
    const bool live1 = ct1 < ctiles1;
    const int co1 = ct1 * 32;
    const long m0 = (long)blockIdx.x * XP_BM;
    const long n0 = m0 / S;
    const int r0 = (int)(m0 - n0 * S);
    const int nks1 = w1_kc / KSTEP, nks2 = w2_kc / KSTEP;

    // ---- all independent loads of stage 1, in consumption order ----------------------------------------------------------
    uint4 xr[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int p = tid + 256 * u, prow = p / PPR, pcol = p - prow * PPR;
        const bool ok = m0 + prow < M && pcol * CH < Cin_p;
        xr[u] = *reinterpret_cast<const uint4*>(x + (ok ? (m0 + prow) * Cin_p + pcol * CH : 0));
    }
    float gv[XF ? 4 : 1];
    if (XF && gate) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u, q = i / w1_kc, k = i - q * w1_kc;
            const bool ok = i < 2 * w1_kc && k < Cin_p && (n0 + q) * (long)S < M;
            const float g = gate[ok ? (n0 + q) * Cin_p + k : 0];
            gv[u] = ok ? g : 0.0f;
        }
    }
    bf16x8 A1[KS1];
    {
        const int ctc = live1 ? ct1 : ctiles1 - 1;
        const __bf16* abase = w1 + ((long)ctc * nks1 * 64 + lane) * CH;  // fragment-major
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) A1[ks] = load_frag<__bf16>(abase + (size_t)(ks < nks1 ? ks : nks1 - 1) * 64 * CH);
    }
    uint4 rr[2][2][1];
    if (live1) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = lane + 64 * i;
                const long m = m0 + j * 32 + (q >> 2);
                const int n = co1 + (q & 3) * 8;
                const bool ok = m < M && n < C1_p;
                rr[j][i][0] = *reinterpret_cast<const uint4*>(res + (ok ? m * C1_p + n : 0));
            }
    }
    for (int i = tid; i < 256; i += 256) {  // stage-2 scale / bias -> LDS (used after two barriers)
        sb2[i] = (s2 && i < C2_p) ? s2[i] : 1.0f;
        sb2[256 + i] = (b2 && i < C2_p) ? b2[i] : 0.0f;
    }

    // ---- stage the gate rows, then the X tile -------------------------------------------------------------------------------
    if (XF) {  // (no gate: rows of 1.0 -- the transform below has no per-element condition)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (tid + 256 * u < 2 * w1_kc) gl[tid + 256 * u] = gate ? gv[u] : 1.0f;
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int p = tid + 256 * u, prow = p / PPR, pcol = p - prow * PPR;
        const bool ok = m0 + prow < M && pcol * CH < Cin_p;
        uint4 xv = xr[u];
        xv.x = ok ? xv.x : 0u;
        xv.y = ok ? xv.y : 0u;
        xv.z = ok ? xv.z : 0u;
        xv.w = ok ? xv.w : 0u;
        if (XF && pcol * CH < Cin_p) {
            __bf16* e = reinterpret_cast<__bf16*>(&xv);
            // the piece's gate values as whole 16-byte reads, all before the arithmetic: `if (gp) f *= gp[j]` compiled to one predicated
            // ds_read_b32 + s_waitcnt per ELEMENT (NP x CH serialised LDS round trips per thread: the gated instances ran 13-17 us longer
            // than the plain ones)
            const float* gp = gl + ((r0 + prow >= S) ? w1_kc : 0) + pcol * CH;
            float gq[CH];
#pragma unroll
            for (int j = 0; j < CH; j += 4) {
                const f32x4 q4 = *reinterpret_cast<const f32x4*>(gp + j);
#pragma unroll
                for (int i = 0; i < 4; ++i) gq[j + i] = q4[i];
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) gq[j] *= (float)e[j];
            if (in_swish) {  // one wave-uniform branch around the piece (inside the element loop it compiles to a select per element)
#pragma unroll
                for (int j = 0; j < CH; ++j) gq[j] = gq[j] * sigmoidf_(gq[j]);
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) e[j] = (__bf16)gq[j];
        }
        *reinterpret_cast<uint4*>(xs + (size_t)prow * KP1 + pcol * CH) = xv;
    }
    __syncthreads();

    // ---- stage 1: K loop (waves without a channel tile run it on clamped weights and discard the result) -----------------
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
        if (ks >= nks1) A1[ks] = zero_frag<__bf16>();
    float sc[4][4], bs[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = co1 + 8 * g + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc[g][q] = 1.0f;
            bs[g][q] = 0.0f;
        }
        if (live1 && n < C1_p) {
            if (s1) load4(s1 + n, sc[g]);
            if (b1) load4(b1 + n, bs[g]);
        }
    }
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
    {
        const __bf16* xb0 = xs + (size_t)c * KP1 + h * CH;
        const __bf16* xb1 = xb0 + (size_t)32 * KP1;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            const bf16x8 v0 = load_frag<__bf16>(xb0 + (size_t)ks * KSTEP);
            const bf16x8 v1 = load_frag<__bf16>(xb1 + (size_t)ks * KSTEP);
            mma32(acc[0], A1[ks], v0);
            mma32(acc[1], A1[ks], v1);
        }
    }
    // stage-2 weights: this wave's tiles wave and wave + 4, requested now (they land under stage 1's epilogue)
    bf16x8 A2[2][KS2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ct2 = wave + 4 * t;
        const int ctc = ct2 < ctiles2 ? ct2 : ctiles2 - 1;
        const __bf16* abase = w2 + ((long)ctc * nks2 * 64 + lane) * CH;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) A2[t][ks] = load_frag<__bf16>(abase + (size_t)(ks < nks2 ? ks : nks2 - 1) * 64 * CH);
    }
    __syncthreads();  // every wave is done READING the X tile: its LDS region becomes stage 2's input tile

    // ---- stage 1 epilogue: rows to HBM and, as bf16, into the stage-2 tile ---------------------------------------------------
    if (live1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = acc[j][4 * g + q] * sc[g][q] + bs[g][q];
                *reinterpret_cast<f32x4*>(scr + c * XP_SROW + 8 * g + 4 * h) = o;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = lane + 64 * i;
                const int row = q >> 2, cg = q & 3;
                const long m = m0 + j * 32 + row;
                const int n = co1 + cg * 8;
                float v[8];
                load8(scr + row * XP_SROW + cg * 8, v);
                float r[8];
                raw_to_f8<__bf16>(rr[j][i], r);
                const bool ok = m < M && n < C1_p;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ok ? v[e] + r[e] : 0.0f;
                act_vec(v, act1);
                mask_tail(v, C1 - n);
                if (ok) store8(y1 + m * C1_p + n, v);
                store8(xs + (size_t)(j * 32 + row) * KP2 + n, v);  // stage-2 tile (zeros for padded channels / rows beyond M)
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();

    // ---- stage 2: expand conv from the LDS tile -------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ct2 = wave + 4 * t;
        if (ct2 >= ctiles2) break;  // wave-uniform
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks)
            if (ks >= nks2) A2[t][ks] = zero_frag<__bf16>();
        f32x16 a2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) a2[j][i] = 0.0f;
        const __bf16* xb0 = xs + (size_t)c * KP2 + h * CH;
        const __bf16* xb1 = xb0 + (size_t)32 * KP2;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            const bf16x8 v0 = load_frag<__bf16>(xb0 + (size_t)ks * KSTEP);
            const bf16x8 v1 = load_frag<__bf16>(xb1 + (size_t)ks * KSTEP);
            mma32(a2[0], A2[t][ks], v0);
            mma32(a2[1], A2[t][ks], v1);
        }
        const int co2 = ct2 * 32;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = co2 + 8 * g + 4 * h;
                const f32x4 sv = *reinterpret_cast<const f32x4*>(sb2 + col);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(sb2 + 256 + col);
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = a2[j][4 * g + q] * sv[q] + bv[q];
                *reinterpret_cast<f32x4*>(scr + c * XP_SROW + 8 * g + 4 * h) = o;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = lane + 64 * i;
                const int row = q >> 2, cg = q & 3;
                const long m = m0 + j * 32 + row;
                const int n = co2 + cg * 8;
                if (m < M && n < C2_p) {
                    float v[8];
                    load8(scr + row * XP_SROW + cg * 8, v);
                    act_vec(v, act2);
                    mask_tail(v, C2 - n);
                    store8(y2 + m * C2_p + n, v);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

static bool xp_pointwise(const pasn_conv_desc& d) {
    return d.kt == 1 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 0 && d.ph == 0 && d.pw == 0;
}
static size_t xp_lds(int ks1, int w1_kc) {
    return (size_t)XP_BM * (ks1 * 16 + 8) * 2 + (size_t)(4 * 32 * XP_SROW + 2 * w1_kc + 512) * sizeof(float);
}

// (KS1, KS2) of the instance, 0 = not covered.  Instances: stage 3 (108 -> 48 -> 108) and stage 4 (216 -> 96 -> 216) shapes.
int pw_xpair_ks(const pasn_conv_desc& d1, const pasn_conv_desc& d2, int dtype, int* ks2_out) {
    if (dtype != PASN_BF16) return 0;
    {
        const char* e = tune("PASN_NO_XPAIR");
        if (e && e[0] == '1') return 0;
    }
    if (!xp_pointwise(d1) || !xp_pointwise(d2) || d2.in_swish) return 0;
    if (d1.N != d2.N || d1.To != d2.Ti || d1.Ho != d2.Hi || d1.Wo != d2.Wi || d1.Cout != d2.Cin || d1.Cout_p != d2.Cin_p) return 0;
    if ((long)d1.To * d1.Ho * d1.Wo < XP_BM) return 0;
    if (d1.w_kc % 16 || d2.w_kc % 16 || d1.w_kc < d1.Cin_p || d2.w_kc < d2.Cin_p || 2 * d1.w_kc > 1024) return 0;
    const int t1 = (d1.Cout_p + 31) / 32, t2 = (d2.Cout_p + 31) / 32;
    if (t1 > 4 || t2 > 8 || d2.Cout_p > 256) return 0;
    if (d1.w_rows < t1 * 32 || d2.w_rows < t2 * 32) return 0;
    const int nks1 = d1.w_kc / 16, nks2 = d2.w_kc / 16;
    // measured on X3D-S: the 216 -> 96 -> 216 pairs (stage 4) save ~5 us each (37 vs 21 + 21 us; gated 50 vs 34 + 21); the
    // 108 -> 48 -> 108 pairs (stage 3) are even with the separate launches, so only K1 > 128 takes the chained kernel
    // (PASN_XPAIR_ALL=1 enables the narrower instances, which the tests exercise)
    const bool all = tune("PASN_XPAIR_ALL") && tune("PASN_XPAIR_ALL")[0] == '1';
    const int ks1 = nks1 <= 8 ? (all ? 8 : 0) : nks1 <= 14 ? 14 : 0;
    const int ks2 = nks2 <= 4 ? 4 : nks2 <= 6 ? 6 : 0;
    if (!ks1 || !ks2 || ks2 * 16 != t1 * 32) return 0;  // stage 1's tiles must fill stage 2's K extent exactly
    if (ks2_out) *ks2_out = ks2;
    return ks1;
}

int launch_pw_xpair(const void* x, const void* w1, const float* s1, const float* b1, const void* res, const float* gate, void* y1,
                    const pasn_conv_desc& d1, const void* w2, const float* s2, const float* b2, void* y2,
                    const pasn_conv_desc& d2, hipStream_t s) {
    int ks2 = 0;
    const int ks1 = pw_xpair_ks(d1, d2, PASN_BF16, &ks2);
    PASN_REQUIRE(ks1 != 0, "geometry not covered by the chained pointwise kernel");
    PASN_REQUIRE(res != nullptr, "the chained kernel expects the project conv's residual");
    const long M = (long)d1.N * d1.To * d1.Ho * d1.Wo;
    const int S = d1.To * d1.Ho * d1.Wo;
    const size_t lds = xp_lds(ks1, d1.w_kc);
    const dim3 grid(ceil_div(M, XP_BM)), block(256);
    const bool xf = gate != nullptr || d1.in_swish != 0;
#define PASN_XP(KS1_, KS2_, XF_)                                                                                              \
    hipLaunchKernelGGL((pwconv_xpair_kernel<KS1_, KS2_, XF_>), grid, block, lds, s, (const __bf16*)x, (const __bf16*)w1, s1, b1, \
                       (const __bf16*)res, gate, (__bf16*)y1, (const __bf16*)w2, s2, b2, (__bf16*)y2, M, S, d1.Cin_p, d1.Cout,  \
                       d1.Cout_p, d1.w_kc, d1.act, d1.in_swish, d2.Cout, d2.Cout_p, d2.w_kc, d2.act)
#define PASN_XP_XF(KS1_, KS2_)          \
    do {                                \
        if (xf) PASN_XP(KS1_, KS2_, true); \
        else PASN_XP(KS1_, KS2_, false);   \
    } while (0)
    if (ks1 == 8 && ks2 == 4) PASN_XP_XF(8, 4);
    else if (ks1 == 8 && ks2 == 6) PASN_XP_XF(8, 6);
    else if (ks1 == 14 && ks2 == 4) PASN_XP_XF(14, 4);
    else PASN_XP_XF(14, 6);
#undef PASN_XP_XF
#undef PASN_XP
    return check_launch("pwconv_xpair_kernel");
}

}  // namespace pasn

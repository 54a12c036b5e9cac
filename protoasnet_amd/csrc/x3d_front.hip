// Fused X3D block front for SMALL spatial planes (7x7; stage 5): 1x1x1 expand conv + BN + ReLU -> depthwise 3x3x3
// (stride 1, pad 1) + BN [+ Swish] [+ SE partial sums], one launch, no cross-workgroup communication.
//
// Why: at 7x7 the two unfused launches move 31 + 43 MB in 19 + 35 us (1.4 TB/s): each is a handful of ~20 us launches'
// worth of latency chains, the stencil spends as many instructions converting bf16 -> fp32 and re-reading weights from
// LDS as on FMAs, and the inner tensor (2.25x the block width) makes a round trip through HBM in between.  Here
//   * a block owns (clip, chunk of Tc = 8 output frames, 32 inner channels).  Phase A computes the expand conv for its
//     32 channels over the Tc + 2 frames it needs (halo frames recomputed: MFMA work is free here) straight into an LDS
//     tile of fp32 PLANES [channel group][half][frame][H+2][W+2][4] with a zero halo -- the MFMA accumulator layout
//     (position on the lane, 4 consecutive channels per quad) is exactly one 16-byte plane element per quad;
//   * phase B is the T-marching stencil (three accumulator sets rotating by name) reading that tile: no conversions
//     (the tile is fp32), no bounds checks (zero halo), no global latency in the loop, and since a WAVE owns one 8-channel
//     group the 27 x 8 stencil weights are wave-uniform: scalar loads, SGPR operands of v_pk_fma (as in stem.hip);
//   * lanes = output positions (h, w); consecutive positions are consecutive 16-byte plane elements (conflict-free).
// SE partial sums: one row per (clip, T chunk): pool_blocks = number of T chunks.
//
// STATUS (round 1): correct (tests/test_gpu_kernels.py::test_x3d_expand_dw_fused) but OPT-IN (PASN_FRONT=1): 92 us per
// stage-5 launch against 54 us for the unfused pair.  In-kernel stamps: setup (operand round trip + tile zeroing) 27 %,
// phase A 8 %, phase B 62 %.  Phase B is LDS-BANDWIDTH bound -- 18 data + 54 weight ds_read_b128 per frame per wave, four
// waves per CU -- and the 104 KB tile leaves ONE block per CU, so the setup latency of a block is never hidden.  What a
// next version needs: channels on the lanes (a lane keeps its 27 x 2 weights in registers, 9 ds_read_b64 per frame) and
// a tile small enough for 2-3 blocks per CU.
#include "common.h"

namespace pasn {

constexpr int XF_TC = 8;  // output frames per block

template <int KS, int HW>
__global__ __launch_bounds__(256) void x3d_front_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ wa,
                                                        const float* __restrict__ sa, const float* __restrict__ ba,
                                                        const float* __restrict__ wb, const float* __restrict__ sb,
                                                        const float* __restrict__ bb, __bf16* __restrict__ y,
                                                        float* __restrict__ pool, pasn_conv_desc d, int nT, int ctiles) {
    constexpr int HH = HW + 2;            // plane edge with the zero halo
    constexpr int FS = HH * HH;           // plane elements per frame
    constexpr int F = XF_TC + 2;          // frames in the tile
    constexpr int NPOS = F * FS;          // elements per plane
    constexpr int P2 = HW * HW;           // output positions per frame
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [8 planes = 4 groups x 2 halves][NPOS][4]
    float* wl = tile + (size_t)8 * NPOS * 4;                      // [27 taps + scale + bias][4 groups][8] stencil weights
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the stencil weights of a wave are SGPR operands
    const int c = lane & 31, h = lane >> 5;
    const int T = d.Ti, K = d.Cin_p, Cp = d.Cout_p;
    const int ct = blockIdx.x % ctiles;
    const int tch = (blockIdx.x / ctiles) % nT;
    const int n = blockIdx.x / (ctiles * nT);
    const int t0 = tch * XF_TC, t1 = min(t0 + XF_TC, T);

    // ---- phase A operands first: ALL of this wave's position tiles (<= 4 x KS fragments: one wave per SIMD leaves the
    // registers) plus the expand weights, requested before anything else so one memory round trip covers them -------------
    const int nks = d.w_kc / 16;
    const int fa = max(t0 - 1, 0) - (t0 - 1);   // first / last tile frame that lies inside the clip
    const int fb = min(t1, T - 1) - (t0 - 1);
    const int rows = (fb - fa + 1) * P2;
    const int ntile = (rows + 31) / 32;          // <= (XF_TC + 2) * 49 / 32 = 16: at most 4 per wave
    constexpr int NTW = (F * P2 + 31) / 32 / 4 + (((F * P2 + 31) / 32) % 4 ? 1 : 0);
    bf16x8 Ball[NTW][KS];
    int lposA[NTW];
    bool validA[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int r = (wave + 4 * i) * 32 + c;
        validA[i] = r < rows;
        const int rr = validA[i] ? r : 0;
        const int f = fa + rr / P2, p2 = rr % P2, hh = p2 / HW, ww = p2 % HW;  // compile-time divisors
        lposA[i] = f * FS + (hh + 1) * HH + (ww + 1);
        const __bf16* xp = x + ((((long)n * T + (t0 - 1 + f)) * HW + hh) * HW + ww) * K + h * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) Ball[i][ks] = load_frag<__bf16>(xp + (ks < nks ? ks : nks - 1) * 16);
    }
    // ---- zero the tile (halo and frames outside the clip stay zero); stage the stencil weights of the 32 channels ---------
    for (int i = tid; i < 8 * NPOS; i += 256) *reinterpret_cast<f32x4*>(tile + (size_t)i * 4) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = tid; i < 29 * 8; i += 256) {  // 16-byte pieces: row (tap | scale | bias) x 8 pieces of 4 channels
        const int row = i >> 3, q4 = i & 7, ch = ct * 32 + q4 * 4;
        const float* src = row < 27 ? wb + (long)row * Cp : (row == 27 ? sb : bb);
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (ch < Cp) v = *reinterpret_cast<const f32x4*>(src + ch);
        *reinterpret_cast<f32x4*>(wl + i * 4) = v;
    }

    // ---- expand weights of this block's 32 channels (rows ct*32 + c), whole K, and their BN ------------------------------
    bf16x8 A[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) A[ks] = load_frag<__bf16>(wa + (long)(ct * 32 + c) * d.w_kc + (ks < nks ? ks : nks - 1) * 16 + h * 8);
    f32x4 sca[4], bia[4];  // channels ct*32 + 8q + 4h .. +3 of accumulator quad q
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sca[q] = *reinterpret_cast<const f32x4*>(sa + ct * 32 + 8 * q + 4 * h);
        bia[q] = *reinterpret_cast<const f32x4*>(ba + ct * 32 + 8 * q + 4 * h);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        if (ks >= nks) A[ks] = zero_frag<__bf16>();  // wave-uniform; only the template steps beyond w_kc

    // ---- phase A: expand conv over the frames t0-1 .. t1 that exist, into the fp32 planes --------------------------------
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        if (wave + 4 * i < ntile) {  // wave-uniform
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) mma32(acc, A[ks], Ball[i][ks]);
            if (validA[i]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaxf(acc[4 * q + j] * sca[q][j] + bia[q][j], 0.0f);
                    *reinterpret_cast<f32x4*>(tile + ((size_t)(q * 2 + h) * NPOS + lposA[i]) * 4) = o;
                }
            }
        }
    }
    __syncthreads();

    // ---- phase B: T-marching stencil on the planes; wave = one 8-channel group, lanes = positions ------------------------
    const int chg = ct * 4 + wave;  // scalar
    if (chg * 8 >= Cp) return;      // this block's last groups may not exist (wave-uniform)
    const float* wq = wl + wave * 8;  // tap t at + t * 32: wave-uniform LDS addresses -> broadcast reads
    float psum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) psum[j] = 0.0f;
    const float* plane0 = tile + (size_t)(wave * 2 + 0) * NPOS * 4;
    const float* plane1 = tile + (size_t)(wave * 2 + 1) * NPOS * 4;

    for (int pass = 0; pass * 64 < P2; ++pass) {
        const int p2 = pass * 64 + lane;
        const bool active = p2 < P2;
        const int pc = active ? p2 : 0;
        const int hh = pc / HW, ww = pc % HW;
        const int lp0 = hh * HH + ww;  // plane element of the (kh = 0, kw = 0) neighbour in frame 0
        float S0[8], S1[8], S2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) S0[j] = S1[j] = S2[j] = 0.0f;

        // one tile frame f (time t0-1+f): Pv = output f-2 (kt = 2), Cv = output f-1 (kt = 1), Nv = output f (kt = 0),
        // outputs counted from t0; after the frame, Pv is complete
        auto frame = [&](int f, float (&Pv)[8], float (&Cv)[8], float (&Nv)[8]) {
            int zo = 0;  // opaque zero, renewed per frame: keeps the 54 weight reads INSIDE the frame loop (216 registers)
            asm volatile("" : "+v"(zo));
            const float* wf = wq + zo;
            const float* q0 = plane0 + (size_t)(f * FS + lp0) * 4;
            const float* q1 = plane1 + (size_t)(f * FS + lp0) * 4;
            // all 9 neighbour vectors of the frame first (18 reads in flight), the weights one kh group (3 neighbours x 3
            // kt taps) ahead of their FMAs: with one wave per SIMD every LDS round trip left in the dependency chain is
            // paid in full (measured 2.7k cycles per frame for 150 instructions in the per-neighbour version)
            f32x4 v0[9], v1[9];
#pragma unroll
            for (int nb = 0; nb < 9; ++nb) {
                v0[nb] = *reinterpret_cast<const f32x4*>(q0 + ((nb / 3) * HH + nb % 3) * 4);  // immediate offsets
                v1[nb] = *reinterpret_cast<const f32x4*>(q1 + ((nb / 3) * HH + nb % 3) * 4);
            }
            f32x4 wc[3][3][2], wn[3][3][2];  // [kw][kt][half] of one kh group
            auto wload = [&](int kh, f32x4 (&wv)[3][3][2]) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int kt = 0; kt < 3; ++kt) {
                        wv[kw][kt][0] = *reinterpret_cast<const f32x4*>(wf + ((kt * 3 + kh) * 3 + kw) * 32);
                        wv[kw][kt][1] = *reinterpret_cast<const f32x4*>(wf + ((kt * 3 + kh) * 3 + kw) * 32 + 4);
                    }
            };
            wload(0, wc);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                if (kh < 2) wload(kh + 1, wn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int nb = kh * 3 + kw;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        Nv[j] = fmaf(v0[nb][j], wc[kw][0][0][j], Nv[j]);
                        Nv[4 + j] = fmaf(v1[nb][j], wc[kw][0][1][j], Nv[4 + j]);
                        Cv[j] = fmaf(v0[nb][j], wc[kw][1][0][j], Cv[j]);
                        Cv[4 + j] = fmaf(v1[nb][j], wc[kw][1][1][j], Cv[4 + j]);
                        Pv[j] = fmaf(v0[nb][j], wc[kw][2][0][j], Pv[j]);
                        Pv[4 + j] = fmaf(v1[nb][j], wc[kw][2][1][j], Pv[4 + j]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int kt = 0; kt < 3; ++kt) {
                        wc[kw][kt][0] = wn[kw][kt][0];
                        wc[kw][kt][1] = wn[kw][kt][1];
                    }
            }
            const int to = t0 + f - 2;
            if (to >= t0 && to < t1) {  // wave-uniform
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = Pv[j] * wf[27 * 32 + j] + wf[28 * 32 + j];
                    psum[j] += active ? v[j] : 0.0f;
                }
                act_vec(v, d.act);
                mask_tail(v, d.Cout - chg * 8);
                if (active) store8(y + ((((long)n * T + to) * HW + hh) * HW + ww) * Cp + chg * 8, v);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) Pv[j] = 0.0f;  // becomes the set of output f + 1
        };
#pragma unroll 1
        for (int f = 0; f < F; f += 3) {
            frame(f, S0, S1, S2);
            if (f + 1 < F) frame(f + 1, S1, S2, S0);
            if (f + 2 < F) frame(f + 2, S2, S0, S1);
        }
    }
    if (pool) {  // block-uniform; one partial row per (clip, T chunk), reduced over the lanes in a fixed order
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = psum[j];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
            if (lane == 0) pool[((long)n * nT + tch) * Cp + chg * 8 + j] = (chg * 8 + j < d.Cout) ? s : 0.0f;
        }
    }
}

// Geometry: ok = 0 -> not this kernel.
XfrontGeom x3d_front_geom(const pasn_conv_desc& d, int dtype) {
    XfrontGeom g = {0, 0, 0, 0, 0};
    if (dtype != PASN_BF16) return g;
    {  // opt-in until it beats the unfused pair (see STATUS above)
        const char* e = getenv("PASN_FRONT");
        if (!e || e[0] != '1') return g;
    }
    const bool shape = d.kt == 3 && d.kh == 3 && d.kw == 3 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 1 && d.ph == 1 &&
                       d.pw == 1 && d.To == d.Ti && d.Ho == d.Hi && d.Wo == d.Wi && d.Hi == d.Wi && d.Hi == 7;
    if (!shape || d.Cout_p % 8 != 0 || d.Cin_p % 8 != 0 || d.w_kc % 16 != 0 || d.w_kc < d.Cin_p) return g;
    const int nks = d.w_kc / 16;
    g.ks = nks <= 2 ? 2 : nks <= 4 ? 4 : nks <= 6 ? 6 : nks <= 8 ? 8 : nks <= 12 ? 12 : 0;
    if (!g.ks) return g;
    if (d.w_rows < ((d.Cout_p + 31) / 32) * 32) return g;
    g.nT = ceil_div(d.To, XF_TC);
    g.ctiles = ceil_div(d.Cout_p, 32);
    g.lds = 8 * (XF_TC + 2) * (d.Hi + 2) * (d.Wi + 2) * 16 + 29 * 32 * 4;
    g.ok = g.lds <= 160 * 1024;
    return g;
}

int launch_x3d_front(const void* x, const void* wa, const float* sa, const float* ba, const float* wb, const float* sb,
                     const float* bb, void* y, float* pool, const pasn_conv_desc& d, const XfrontGeom& g, hipStream_t s) {
    const dim3 grid((unsigned)d.N * g.nT * g.ctiles), block(256);
#define PASN_XF(KS_)                                                                                                        \
    do {                                                                                                                    \
        PASN_MAX_LDS(160 * 1024, x3d_front_kernel<KS_, 7>);                                                               \
        hipLaunchKernelGGL((x3d_front_kernel<KS_, 7>), grid, block, (size_t)g.lds, s, (const __bf16*)x, (const __bf16*)wa, sa, \
                           ba, wb, sb, bb, (__bf16*)y, pool, d, g.nT, g.ctiles);                                            \
    } while (0)
    switch (g.ks) {
        case 2: PASN_XF(2); break;
        case 4: PASN_XF(4); break;
        case 6: PASN_XF(6); break;
        case 8: PASN_XF(8); break;
        default: PASN_XF(12); break;
    }
#undef PASN_XF
    return check_launch("x3d_front_kernel");
}

}  // namespace pasn

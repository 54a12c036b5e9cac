// Temporal (3,1,1) conv, stride 1, pad (1,0,0), + BN (+ residual) + activation, bf16, weight-stationary and T-MARCHING (round 5): the second
// half of R(2+1)D's Conv2Plus1D (resnet_features.py's r2plus1d_18 trunk: Conv3d(mid, out, (3,1,1)) after every (1,3,3) conv) where the weights
// of a 32-channel tile fit a wave's registers for the whole K = 3 Cin (Cin <= 144: the 144 -> 64 layers of the 56 x 56 stage).
//
// The layer is a pointwise conv whose K axis is three FRAMES of the same position: y[t][p] = sum_dt W[dt] x[t + dt - 1][p].  The halo-tile
// implicit GEMM (igemm_halo.hip) treats it as a windowed conv: 112-131 us for 334-437 MB (3.0-3.3 TB/s) at 8 x 32 x 56 x 56.  Here:
//   * block = CT x 2 waves = one tile of 64 positions of one clip, ALL its frames: wave (ct, pt) keeps channel tile ct's weights for the whole
//     K (27 fragments at Cin = 144: 108 VGPRs) and owns 32 of the 64 positions; the block MARCHES ALONG T;
//   * every input frame's tile [64 positions][Cin] arrives ONCE by LDS-DMA into a ring of four slots (t - 1, t, t + 1 in use, t + 2 in
//     flight) and is the B operand of three output frames: k-steps 0 .. KSF - 1 read slot t - 1, the next KSF slot t, the last slot t + 1;
//     frames outside the clip are skipped (wave-uniform), not staged as zeros;
//   * ONE fence-free barrier per frame; what must have landed is waited for by count; epilogue as pwconv_ws.hip's (lane swap -> 8 channels of
//     a position per lane, scale / bias / residual / activation, one 16-byte store per piece); the residual rows come straight from memory
//     into registers, requested before the MFMA chain.
// 78 KB of LDS: two blocks = eight waves per CU.  Arithmetic: the MFMA k order is (dt, channel) -- igemm_halo's is the same per tap -- fp32
// accumulation; results agree with the implicit GEMM to fp32 summation order.
#include "common.h"

namespace pasn {

typedef __attribute__((address_space(3))) void* tc_lds_ptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned tc_u32x4;

constexpr unsigned TC_OOB = 0x80000000u;
constexpr int TC_BM = 64;   // positions per tile
constexpr int TC_NSL = 4;   // ring slots

__device__ __forceinline__ void tc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void tc_wait_all_but(int n) {  // n wave-uniform, 0 .. 8
    switch (n) {
#define TC_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        TC_W(0) TC_W(1) TC_W(2) TC_W(3) TC_W(4) TC_W(5) TC_W(6) TC_W(7) TC_W(8)
#undef TC_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// KSF: k-steps per frame (Cin_p / 16); CT: channel tiles of 32 (the block has 2 CT waves); RES: a residual tensor is added before the activation
template <int KSF, int CT, bool RES>
__global__ __launch_bounds__(128 * CT) void tconv_ws_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ w, const float* __restrict__ scale,
                                                       const float* __restrict__ bias, const __bf16* __restrict__ res, __bf16* __restrict__ y,
                                                       int N, int T, int HW, int Cin_p, int Cout, int Cout_p, int act, TcGeom g) {
    constexpr int KS = 3 * KSF;
    constexpr int PPRL = 2 * KSF + 1;  // 16-byte slots per staged row: Cin_p / 8 pieces + one pad slot (odd stride: conflict-free ds_read_b128)
    constexpr int SLOT = TC_BM * PPRL * 16;
    extern __shared__ __attribute__((aligned(1024))) char smem[];  // [TC_NSL][SLOT]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NW = CT * 2;
    const int ct = wave % CT, pt = wave / CT;
    const int c = lane & 31, h = lane >> 5;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int n = lb / g.ptiles, p0 = (lb - n * g.ptiles) * g.bm, bm = g.bm;  // bm <= 64 positions of the tile are real (host: the tile count that fills the card)
    const unsigned xrow = (unsigned)Cin_p * 2u, yrow = (unsigned)Cout_p * 2u;
    const unsigned xframe = (unsigned)HW * xrow, yframe = (unsigned)HW * yrow;
    // this clip (every frame): descriptors' ranges end with the clip
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x + (long)n * T * HW * Cin_p), 0, (unsigned)T * xframe, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(y + (long)n * T * HW * Cout_p, 0, (unsigned)T * yframe, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(RES ? res + (long)n * T * HW * Cout_p : x), 0, RES ? (unsigned)T * yframe : 0u, 0x00020000);

    // ---- stationary weights: fragment-major (tile ct, step ks) = 64 lanes x 16 bytes, K = (dt, channel) ----
    bf16x8 A[KS];
    {
        const __bf16* ab = w + ((long)ct * KS * 64 + lane) * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) A[ks] = load_frag<__bf16>(ab + (size_t)ks * 512);
    }
    // epilogue constants of this lane's two 8-channel pieces (after the lane swap: piece pr = channels 32 ct + 16 pr + 8 h ..)
    float sc[2][8], bs[2][8];
    unsigned yoff[2];
    const bool row_ok = pt * 32 + c < bm && p0 + pt * 32 + c < HW;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const int ch = ct * 32 + 16 * pr + 8 * h;
        const bool ok = ch < Cout_p;
        const int chc = ok ? ch : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[pr][j] = 1.0f;
            bs[pr][j] = 0.0f;
        }
        if (scale) load8(scale + chc, sc[pr]);
        if (bias) load8(bias + chc, bs[pr]);
        yoff[pr] = (ok && row_ok) ? (unsigned)(p0 + pt * 32 + c) * yrow + (unsigned)ch * 2u : TC_OOB;
    }
    const bool tail = ct * 32 + 32 > Cout;  // wave-uniform: this tile holds channels beyond the real count (stored as zeros)

    // ---- DMA roles: instruction j of a frame covers the slot's 16-byte cells 64 j .. 64 j + 63; wave `wave` issues j = wave + NW e, e < NE --
    // EVERY wave issues NE instructions per frame (those beyond the tile carry out-of-range offsets and land in a spare KB behind the ring):
    // the count is a compile-time constant, so the compiler's own wait for the residual registers names exactly the DMAs issued behind them
    constexpr int NIX = (TC_BM * PPRL + 63) >> 6;
    constexpr int NE = (NIX + NW - 1) / NW;
    unsigned xoff[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int s = (wave + NW * e) * 64 + lane;
        const int r = s / PPRL, p = s - r * PPRL;  // compile-time divisor
        xoff[e] = (wave + NW * e < NIX && r < bm && p < 2 * KSF && p0 + r < HW) ? (unsigned)(p0 + r) * xrow + (unsigned)p * 16u : TC_OOB;
    }
    auto issue = [&](int f) {  // frame f of the clip -> slot f % 4; a frame behind the clip's end: NE instructions that fetch nothing
        const bool in = f < T;   // (issued all the same, NOT branched around: a branch would make the count unknown to the compiler again)
        char* sb = smem + (f & (TC_NSL - 1)) * SLOT;
        const unsigned fo = in ? (unsigned)f * xframe : 0u;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int j = wave + NW * e;
            char* dst = (in && j < NIX) ? sb + j * 1024 : smem + TC_NSL * SLOT;  // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (tc_lds_ptr_t)dst, 16, (int)(in ? xoff[e] : TC_OOB), (int)fo, 0, 0);
        }
    };
    // the weights and constants are in their registers before the march starts (no wait of the compiler's for them inside the loop)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(A[ks]));
    issue(0);
    issue(1);
    constexpr int KST = 2;  // stores of this wave per frame

#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        // frame t + 1 (requested one step ago; t = 0: in the prologue, with frame 0) has landed: this wave has since issued only the stores
        // of frame t - 1; behind the barrier everyone's pieces have, and nobody still reads frame t - 2's slot
        tc_wait_all_but(t == 0 ? 0 : KST);
        tc_barrier();
        tc_u32x4 rq[2];
        if (RES) {  // (ahead of the DMA group: the wait for these registers then leaves the group in flight)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) rq[pr] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, (int)yoff[pr], (int)((unsigned)t * yframe), 0);
        }
        issue(t + 2);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int fp = 0; fp < 3; ++fp) {
            const int f = t + fp - 1;
            if (f < 0 || f >= T) continue;  // wave-uniform: the temporal padding contributes nothing
            const char* xb = smem + (f & (TC_NSL - 1)) * SLOT + ((pt * 32 + c) * PPRL + h) * 16;
#pragma unroll
            for (int kk = 0; kk < KSF; ++kk) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(xb + 2 * kk * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[fp * KSF + kk], b, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            // registers 8 pr + q (channels 32 ct + 16 pr + 4 h + q) and 8 pr + 4 + q (+ 8): after the half-wave exchange lanes < 32 hold channels
            // 16 pr .. + 7 and lanes >= 32 channels 16 pr + 8 .. + 15 of their position
            float v[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * pr + q]), __float_as_uint(acc[8 * pr + 4 + q]), false, false);
                v[q] = __uint_as_float(sw[0]);
                v[4 + q] = __uint_as_float(sw[1]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[pr][e] + bs[pr][e];
            if (RES) {
                float r8[8];
                const bf16x8 rv = __builtin_bit_cast(bf16x8, rq[pr]);
#pragma unroll
                for (int e = 0; e < 8; ++e) r8[e] = (float)rv[e];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r8[e];
            }
            act_vec(v, act);
            if (tail) mask_tail(v, Cout - (ct * 32 + 16 * pr + 8 * h));
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
            // (frame offset in the vector offset, soffset 0: profiles/README.md entry 144)
            const unsigned off = yoff[pr] == TC_OOB ? TC_OOB : yoff[pr] + (unsigned)t * yframe;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tc_u32x4, o), yrsrc, (int)off, 0, 0);
        }
    }
}

// ---- host -----------------------------------------------------------------------------------------------------------------------------
TcGeom tconv_geom(const pasn_conv_desc& d, int dtype, bool has_gate) {
    TcGeom g{};
    if (dtype != PASN_BF16 || has_gate || d.in_swish || d.w_frag != 1) return g;
    if (tune("PASN_TCONV") && tune("PASN_TCONV")[0] == '0') return g;
    const bool shape = d.kt == 3 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 1 && d.ph == 0 && d.pw == 0 &&
                       d.To == d.Ti && d.Ho == d.Hi && d.Wo == d.Wi;
    if (!shape || d.Cin_p % 16 != 0 || d.w_kc != d.Cin_p) return g;
    g.KSF = d.Cin_p / 16;
    if (g.KSF != 3 && g.KSF != 4 && g.KSF != 9) return TcGeom{};  // compiled instances: 48, 64 and 144 (padded) input channels (the weights of a channel tile stay in registers: 3 KSF <= 28 fragments)
    g.CT = ceil_div(d.Cout_p, 32);
    if (g.CT > 2) return TcGeom{};           // compiled instances: up to 64 output channels
    const long HW = (long)d.Ho * d.Wo;
    if ((long)d.Ti * HW * std::max(d.Cin_p, d.Cout_p) * 2 >= (1L << 31)) return TcGeom{};  // one clip per buffer descriptor
    // tiles per frame: at least HW / 64; more (shorter tiles, the MFMA tile's last rows idle -- the launch is bound by its bytes) where that
    // fills the card's 512 block slots better: 8 clips of 56 x 56 are 392 blocks of 64 positions or exactly 512 of 49
    {
        const int tmin = (int)ceil_div(HW, (long)TC_BM), tmax = (int)ceil_div(HW, 40L);
        double best = 1e30;
        for (int t = tmin; t <= tmax; ++t) {
            const double cost = (double)ceil_div((long)d.N * t, 512L) * (double)ceil_div(HW, (long)t);
            if (cost < best) {
                best = cost;
                g.ptiles = t;
            }
        }
        g.bm = (int)ceil_div(HW, (long)g.ptiles);
    }
    g.lds = TC_NSL * TC_BM * (2 * g.KSF + 1) * 16 + 1024;  // + the KB where the DMA instructions beyond the tile land
    g.ok = 1;
    return g;
}

int launch_tconv_ws(const void* x, const void* w, const float* scale, const float* bias, const void* res, void* y, const pasn_conv_desc& d,
                    const TcGeom& g, hipStream_t s) {
    const dim3 grid((unsigned)((long)d.N * g.ptiles)), block((unsigned)(g.CT * 2 * 64));
    const int HW = d.Ho * d.Wo;
#define PASN_TC(KSF_, CT_, RES_)                                                                                                         \
    do {                                                                                                                               \
        PASN_MAX_LDS(80 * 1024, tconv_ws_kernel<KSF_, CT_, RES_>);                                                                     \
        hipLaunchKernelGGL((tconv_ws_kernel<KSF_, CT_, RES_>), grid, block, (size_t)g.lds, s, (const __bf16*)x, (const __bf16*)w, scale, bias, \
                           (const __bf16*)res, (__bf16*)y, (int)d.N, (int)d.Ti, HW, (int)d.Cin_p, (int)d.Cout, (int)d.Cout_p, (int)d.act, g);  \
    } while (0)
#define PASN_TC2(KSF_, CT_)               \
    do {                                  \
        if (res) PASN_TC(KSF_, CT_, true); \
        else PASN_TC(KSF_, CT_, false);   \
    } while (0)
    if (g.KSF == 9) {
        if (g.CT == 2) PASN_TC2(9, 2);
        else PASN_TC2(9, 1);
    } else if (g.KSF == 3) {
        if (g.CT == 2) PASN_TC2(3, 2);
        else PASN_TC2(3, 1);
    } else {
        if (g.CT == 2) PASN_TC2(4, 2);
        else PASN_TC2(4, 1);
    }
#undef PASN_TC2
#undef PASN_TC
    return check_launch("tconv_ws_kernel");
}

}  // namespace pasn

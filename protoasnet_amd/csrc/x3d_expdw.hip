// First block of an X3D stage, front half, in ONE launch (bf16): 1x1x1 expand conv + BN + ReLU -> depthwise 3x3x3 conv, stride (1,2,2),
// pad 1 + BN (+ Swish, + squeeze-excite partial sums).  The expanded activation -- 2.25x the block width at the INPUT resolution, four
// times the positions of everything downstream -- never reaches memory.
//
// Why: the stride-2 blocks are where the expanded tensor costs most.  At the benchmark shape the first block of stage 2 writes 694 MB
// (54 channels at 112 x 112) and reads it back once: 1.39 GB of the step's 11 GB, for a K = 24 GEMM; unfused that is two launches at the
// fabric's rate (182 + 235 us).  Round 1's fused kernel (VALU stencil from an LDS ring) lost to the unfused pair; this one runs BOTH halves
// on the matrix cores:
//   * block = 4 waves = one 64-channel quad of the expanded channels of one clip; a unit = (T chunk, region of 3 x 14 outputs); the block
//     MARCHES ALONG T over the 7 x 29 input positions the region needs;
//   * per input frame: the x rows of the region arrive by LDS-DMA (out-of-image pieces zero-filled by the hardware); the expand conv is
//     v_mfma_f32_32x32x16_bf16 with a staged ROW of the region as the 32-position tile (A = this wave's 32 expand channels, whole K, in
//     registers for the launch); the accumulator goes through the lane swap (8 channels of one position per lane), BN + ReLU, is forced
//     to ZERO outside the image (the stencil pads the EXPANDED activation), and is written as one 16-byte piece per lane into the frame
//     image of a 2-frame ring -- [position][9 slots], the layout dwmfma.hip stages by DMA;
//   * the stencil is dwmfma.hip's: v_mfma_f32_16x16x32_bf16 with block-diagonal weight operands, one 16-byte LDS read per lane and tap
//     pair, three accumulator sets (outputs t-1, t, t+1) whose role rotation rides in the MFMAs, outputs stored straight from the
//     accumulators through a per-frame buffer descriptor;
//   * ONE fence-free barrier per frame: frame t+1 is expanded into the other ring slot before frame t's stencil, the x rows of frame
//     t+2 are requested in between; what must have landed is waited for by count.
// Rounding points are those of the two launches (the expanded activation is rounded to bf16 before the stencil; fp32 accumulation), the
// MFMA k order of the expand conv is pwconv's, the stencil's is dwmfma's.
#include "common.h"

namespace pasn {

typedef __attribute__((ext_vector_type(4))) unsigned xe_u32x4;
typedef __attribute__((ext_vector_type(2))) short xe_s16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 xe_bf16x2;
typedef __attribute__((ext_vector_type(2))) float xe_f32x2;
typedef __attribute__((ext_vector_type(2))) unsigned xe_u32x2;
typedef __attribute__((address_space(3))) void* xe_lds_ptr_t;

// Region geometry per stride.  Stride 2: 3 x 14 outputs from 7 x 29 staged positions, an expand MFMA tile = one staged row (29 of 32
// lanes), 9 slots per position.  Stride 1: 6 x 14 outputs from 8 x 16 staged positions = exactly four 32-position MFMA tiles (two staged
// rows each), 10 slots per position (the slot counts are dwmfma.hip's bank rule: stride x slots = 2 mod 4).
template <int SS>
struct XeR {
    static constexpr int NT = SS == 2 ? 3 : 6;                  // output rows of a region = 16-lane position tiles per wave
    static constexpr int BW = 14;                               // output columns of a region
    static constexpr int RW = (BW - 1) * SS + 3;                // staged columns: 29 / 16
    static constexpr int RH = (NT - 1) * SS + 3;                // staged rows: 7 / 8
    static constexpr int POS = RH * RW;                         // staged positions per frame: 203 / 128
    static constexpr int SLOTS = SS == 2 ? 9 : 10;              // 16-byte slots per staged position (8 used)
    static constexpr int FRB = (POS * SLOTS * 16 + 1023) / 1024 * 1024;  // bytes per frame image
    static constexpr int TP = SS == 2 ? RW : 32;                // staged positions per expand tile (stride 2: a row; stride 1: two rows)
    static constexpr int TILES = SS == 2 ? RH : POS / 32;       // expand tiles per frame: 7 / 4
};
constexpr int XE_NE = 6;  // x DMA instructions per wave and frame, at most (203 positions x <= 7 slots)
constexpr unsigned XE_OOB = 0x80000000u;

__device__ __forceinline__ unsigned xe_bf16_bits(float f) {
    const __bf16 b = (__bf16)f;
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ void xe_wait_all_but(int n) {  // n wave-uniform
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;  // n <= NT <= 6
    }
}
__device__ __forceinline__ void xe_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// KS: k-steps of the expand conv held in registers (K = block width <= 16 KS channels); ACT: the stencil's epilogue activation
template <int KS, int ACT, int SS, bool ABLB = false, bool FOLD = false>  // ABLB: timing-only ablation instance (PASN_EXPDW_ABL; results are wrong when set)
__global__ __launch_bounds__(256, 2) void x3d_expdw_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ wa,
                                                           const float* __restrict__ sa, const float* __restrict__ ba,
                                                           const float* __restrict__ w, const float* __restrict__ scale,
                                                           const float* __restrict__ bias, __bf16* __restrict__ y, float* __restrict__ pool,
                                                           pasn_conv_desc d, int Cin_p, int nks, XeGeom g) {
    using R = XeR<SS>;
    constexpr int XE_NT = R::NT, XE_BW = R::BW, XE_RW = R::RW, XE_RH = R::RH, XE_POS = R::POS, XE_SLOTS = R::SLOTS, XE_FRB = R::FRB;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const ring = smem;                                          // [2][XE_FRB] expanded frames
    char* const xt = smem + 2 * XE_FRB;                               // [2][g.xtb] x rows of the region: [position][XS slots]
    float* const scb = reinterpret_cast<float*>(xt + 2 * g.xtb);      // [2][64] expand scale | bias of this quad
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* const dummy = reinterpret_cast<char*>(scb) + 512 + lane * 16;  // [64 x 16 + 32] where lanes without a position put their stores (never read)
    const int m = lane & 15, q = lane >> 4;   // stencil roles
    const int c32 = lane & 31, h32 = lane >> 5;  // expand roles
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int n = lb / g.bpc, bx = lb - n * g.bpc;
    const int chunk = bx / g.CQ, cq = bx - chunk * g.CQ;
    const int Cp = d.Cout_p;
    const int c0 = (cq * 4 + wave) * 16;
    const bool wave_live = c0 < Cp;
    const bool wave_tail = c0 + 16 > d.Cout;
    const int XS = g.XS, pieces = Cin_p >> 3;
    const bool fuse = g.fuse != 0;
    const int abl = ABLB ? g.abl : 0;  // 1: no expand MFMAs, 2: no expand epilogue arithmetic, 4: no x DMA, 8: no stencil MFMAs, 16: no output epilogue / stores

    // ---- stencil: block-diagonal weight operands (dwmfma.hip) ----
    xe_u32x4 A[3][5];
    {
        const int c = c0 + m;
        const bool mine = ((m >> 3) == (q & 1)) && c < Cp;
        const int dwsel = (m & 7) >> 1, sh = (m & 1) * 16;
        float wv[3][5];
        const int cc = min(c, Cp - 1);
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int j = 0; j < 5; ++j) wv[kt][j] = w[(kt * 9 + min(2 * j + (q >> 1), 8)) * Cp + cc];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const bool live = mine && 2 * j + (q >> 1) < 9;
                const unsigned bits = live ? (xe_bf16_bits(wv[kt][j]) << sh) : 0u;
                A[kt][j] = xe_u32x4{dwsel == 0 ? bits : 0u, dwsel == 1 ? bits : 0u, dwsel == 2 ? bits : 0u, dwsel == 3 ? bits : 0u};
            }
    }
    const int ce = c0 + 4 * q;
    const bool cev = ce < Cp;
    // FOLD instances (the plan's form): the stencil's epilogue constants of this quad live in LDS and are read where they are used -- as eight
    // registers held across the march they were what pushed the squeeze-excite instance <2, none, 2, fold> over 256 (8 spilled VGPRs, scratch
    // traffic inside the pinned schedule: round-3 verdict).  They take the place of the expand conv's scale | bias table, which a FOLD instance
    // reads only once, into biasC, before the march (the launch sits at 79.6 KB of LDS: two blocks per CU leave no room for another table).
    // (zero for the padded channels: act(0 * P + 0) = 0 for none / ReLU / Swish -- the epilogue stores without a tail mask)
    float sc_r[4], bs_r[4];
    if (!FOLD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sc_r[i] = (cev && ce + i < d.Cout) ? scale[ce + i] : 0.0f;
            bs_r[i] = (cev && ce + i < d.Cout) ? bias[ce + i] : 0.0f;
        }
    }
    const float* const scl = scb + wave * 16 + 4 * q;  // FOLD: this lane's four channels of the [2][64] table
    float psum[4] = {0.0f, 0.0f, 0.0f, 0.0f};

    // ---- expand conv: this wave's 32-channel tile (ct = wave & 1 of the quad) for the expand tiles (wave >> 1) + 2 i of every frame ----
    const int ect = wave & 1;
    const int ectiles = (Cp + 31) >> 5;
    bf16x8 AE[KS];
    {
        const int ctg = min(cq * 2 + ect, ectiles - 1);
        const __bf16* ab = wa + ((long)ctg * nks * 64 + lane) * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) AE[ks] = load_frag<__bf16>(ab + (size_t)(ks < nks ? ks : nks - 1) * 512);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            if (ks >= nks || cq * 2 + ect >= ectiles) AE[ks] = zero_frag<__bf16>();  // wave-uniform
    }
    if (threadIdx.x < 64) {
        const int ch = cq * 64 + threadIdx.x;
        scb[threadIdx.x] = (ch < Cp && sa) ? sa[ch] : 1.0f;
        scb[64 + threadIdx.x] = ch < Cp ? ba[ch] : 0.0f;
    }
    __syncthreads();
    // FOLD (scale_a == NULL, the plan's form): norm_a's scale is folded into the expand weights (W * scale rounded to bf16 ONCE, by the host) and its
    // bias is the accumulator's initial value -- the epilogue is then lane swap, ReLU, rounding: a third fewer vector instructions per
    // expanded element on a kernel bound by vector issue.  biasC: this lane's 16 accumulator rows = channels 32 ect + acc_row(r, h32)
    constexpr bool folded = FOLD;  // (a compile-time property: the host launches the FOLD instance when scale_a == NULL)
    f32x16 biasC;
#pragma unroll
    for (int r = 0; r < 16; ++r) biasC[r] = folded ? scb[64 + 32 * ect + acc_row(r, h32)] : 0.0f;
    if (FOLD) {
        __syncthreads();  // everyone holds its biasC: the table's place takes the stencil's scale | bias
        if (threadIdx.x < 64) {
            const int ch = cq * 64 + threadIdx.x;
            scb[threadIdx.x] = ch < d.Cout ? scale[ch] : 0.0f;
            scb[64 + threadIdx.x] = ch < d.Cout ? bias[ch] : 0.0f;
        }
        __syncthreads();
    }
    const int Ti = d.Ti, Hi = d.Hi, Wi = d.Wi;
    const long fx = (long)Hi * Wi * Cin_p;  // elements per x frame
    const unsigned fx_bytes = (unsigned)(fx * 2);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x + (long)n * Ti * fx), 0, (unsigned)Ti * fx_bytes, 0x00020000);
    const int nix = (XE_POS * XS + 63) >> 6;   // 1-KiB x DMA instructions per frame
    int tapoff[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int tap9 = min(2 * j + (q >> 1), 8);
        tapoff[j] = ((tap9 / 3) * XE_RW + (tap9 % 3)) * (XE_SLOTS * 16);
    }
    const int regions = g.RTH * g.RTW;
    const int units = g.nT * regions;
    const int u_end = min(units, (chunk + 1) * g.upb);

#pragma unroll 1
    for (int u = chunk * g.upb; u < u_end; ++u) {
        const int tch = u / regions, reg = u - tch * regions;
        const int rth = reg / g.RTW, rtw = reg - rth * g.RTW;
        const int t0 = tch * g.Tc, t1 = min(t0 + g.Tc, d.To);
        const int h0 = rth * XE_NT, w0 = rtw * XE_BW;
        // ---- x DMA roles: instruction i = wave + 4 e covers slots 64 i .. of the x tile; slot -> (staged position, piece) ----
        unsigned goff[XE_NE];
#pragma unroll
        for (int e = 0; e < XE_NE; ++e) {
            const int slot = (wave + 4 * e) * 64 + lane;
            const int rp = slot / XS, p = slot - rp * XS;
            const int rr = rp / XE_RW, cc = rp - rr * XE_RW;
            const int hi = h0 * SS - 1 + rr, wi = w0 * SS - 1 + cc;
            const bool ok = wave + 4 * e < nix && rp < XE_POS && p < pieces && hi >= 0 && hi < Hi && wi >= 0 && wi < Wi;
            goff[e] = ok ? (unsigned)(((hi * Wi + wi) * Cin_p + p * 8) * 2) : XE_OOB;
        }
        const int kdma = max(0, (nix - wave + 3) >> 2);
        auto staged = [&](int ti) -> bool { return ti >= 0 && ti < Ti && ti >= t0 - 1 && ti <= t1; };
        auto slot_of = [&](int ti) -> int { return (ti - (t0 - 1)) & 1; };
        auto issue_x = [&](int ti) {
            if (!staged(ti) || (abl & 4)) return;
            const unsigned foff = (unsigned)ti * fx_bytes;
            char* dst = xt + slot_of(ti) * g.xtb;
#pragma unroll
            for (int e = 0; e < XE_NE; ++e)
                if (wave + 4 * e < nix)  // wave-uniform
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (xe_lds_ptr_t)(dst + (wave + 4 * e) * 1024), 16, (int)goff[e], (int)foff, 0, 0);
        };
        // this lane's staged position inside expand tile i of a frame: stride 2: (row tile, column c32 < 29); stride 1: (row 2 tile + c32 / 16,
        // column c32 % 16).  keepm bit i: the position lies inside the image (the stencil pads the EXPANDED activation with zeros)
        constexpr int NU = (R::TILES + 1) / 2;  // expand tiles per wave, at most
        const bool lane_used = c32 < R::TP;
        unsigned keepm = 0;
        bool all_in = true;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int tile = (wave >> 1) + 2 * i;
            const int rr = SS == 2 ? tile : 2 * tile + (c32 >> 4), cc = SS == 2 ? c32 : (c32 & 15);
            const bool in = (unsigned)(h0 * SS - 1 + rr) < (unsigned)Hi && (unsigned)(w0 * SS - 1 + cc) < (unsigned)Wi && cc < XE_RW;
            keepm |= in ? (1u << i) : 0u;
            const bool lane_counts = lane_used && tile < R::TILES;
            all_in = all_in && (__ballot(lane_counts && !in) == 0);
        }
        // expand frame ti from its x tile into its ring image.  (K columns beyond the block width carry zero WEIGHTS -- the packed rows
        // are zero-padded to w_kc -- so the lanes that supply them read a real piece instead of selecting a zero fragment.)
        // The three parts of expand tile i of this wave (tile (wave >> 1) + 2 i of the frame), shared by produce() and by the fused step:
        // p_read: the x fragments; p_mma: the chain of KS 32x32x16 MFMAs; p_epi: ReLU + rounding + lane swap + border mask + 2 ds_write_b128.
        // Lanes / tiles that hold nothing (lanes 29-31 at stride 2, the fourth tile of waves 2 and 3) run like the others on clamped
        // addresses and store into a dummy region instead of branching around the store (a branch ends the scheduling region).
        auto p_read = [&](int i, const char* xb, bf16x8 (&xf)[KS]) {
            const int tile = (wave >> 1) + 2 * i;
            const int tl = min(tile, R::TILES - 1);
            const int pos = tl * R::TP + min(c32, R::TP - 1);
            const char* xp = xb + pos * XS * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xp + min(2 * ks + h32, XS - 1) * 16);
        };
        auto p_mma = [&](const bf16x8 (&xf)[KS], f32x16& acc) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (abl & 1) acc = biasC;
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AE[ks], xf[ks], ks == 0 ? biasC : acc, 0, 0, 0);
            }
        };
        auto p_epi = [&](int i, char* rb, const f32x16& acc) {
            const int tile = (wave >> 1) + 2 * i;
            const int tl = min(tile, R::TILES - 1);
            const int pos = tl * R::TP + min(c32, R::TP - 1);
            const bool wr = lane_used && tile < R::TILES;
            char* const rp = wr ? rb + (pos * XE_SLOTS + 4 * ect + h32) * 16 : dummy;
            const bool keep = (keepm >> i) & 1u;
            // ReLU and the bf16 rounding BEFORE the lane swap, on packed pairs: v_cvt_pk_bf16_f32 + v_pk_max_i16 (a bf16 is
            // negative iff it is negative as an int16; rounding keeps the sign, so max(round(v), 0) == round(max(v, 0)) and
            // -0 becomes +0 either way) and half the swaps -- 14 vector instructions per 8 x 64 outputs where the float path
            // (swap, canonicalise, max, convert) took 34, on a kernel bound by vector issue.
            unsigned P[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xe_f32x2 a = {acc[2 * j], acc[2 * j + 1]};
                if (!folded) {  // separate scale / bias: rows acc_row(2j, h32), +1 of this lane's 16 (before the swap)
                    a.x = a.x * scb[32 * ect + acc_row(2 * j, h32)] + scb[64 + 32 * ect + acc_row(2 * j, h32)];
                    a.y = a.y * scb[32 * ect + acc_row(2 * j + 1, h32)] + scb[64 + 32 * ect + acc_row(2 * j + 1, h32)];
                }
                xe_s16x2 m = __builtin_bit_cast(xe_s16x2, __builtin_convertvector(a, xe_bf16x2));  // one v_cvt_pk_bf16_f32
                if (!(abl & 2)) m = __builtin_elementwise_max(m, xe_s16x2{0, 0});
                P[j] = __builtin_bit_cast(unsigned, m);
            }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(P[4 * pr + 0], P[4 * pr + 2], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(P[4 * pr + 1], P[4 * pr + 3], false, false);
                xe_u32x4 ou = {s0[0], s1[0], s0[1], s1[1]};  // this lane's 8 consecutive channels 32 ect + 16 pr + 8 h32 ..
                if (!all_in) {  // wave-uniform: only regions on the image border select (incl. whole tiles of the zero padding)
                    ou.x = keep ? ou.x : 0u;
                    ou.y = keep ? ou.y : 0u;
                    ou.z = keep ? ou.z : 0u;
                    ou.w = keep ? ou.w : 0u;
                }
                *reinterpret_cast<xe_u32x4*>(rp + pr * 32) = ou;
            }
        };
        // Straight-line, two tiles at a time: the fragment reads of BOTH tiles, then their MFMA chains, then the two epilogues -- with a
        // wave-uniform branch per tile (rows of the zero padding, a wave's missing fourth tile) every tile paid an LDS round trip and an
        // MFMA chain latency on its own (read -> wait -> MFMA -> read -> wait -> MFMA -> swap ...: ~450 cycles each, four per frame).
        auto produce = [&](int ti) {
            if (!staged(ti)) return;
            const char* xb = xt + slot_of(ti) * g.xtb;
            char* rb = ring + slot_of(ti) * XE_FRB;
#pragma unroll
            for (int i0 = 0; i0 < NU; i0 += 2) {
                bf16x8 xf0[KS], xf1[KS];
                f32x16 acc0, acc1;
                p_read(i0, xb, xf0);
                if (i0 + 1 < NU) p_read(i0 + 1, xb, xf1);
                p_mma(xf0, acc0);
                if (i0 + 1 < NU) p_mma(xf1, acc1);
                p_epi(i0, rb, acc0);
                if (i0 + 1 < NU) p_epi(i0 + 1, rb, acc1);
            }
        };

        // ---- stencil roles (dwmfma.hip, one output row per tile) ----
        const bool lane_ok = m < XE_BW && w0 + m < d.Wo && cev;
        const int rows_valid = min(XE_NT, d.Ho - h0);
        const int ntl = rows_valid;
        const int lbase0 = ((min(m, XE_BW - 1) * SS) * XE_SLOTS + 2 * wave + (q & 1)) * 16;
        constexpr int lstep = SS * XE_RW * XE_SLOTS * 16;
        const int ystep = d.Wo * Cp;
        __bf16* yclip = y + (long)n * d.To * d.Ho * d.Wo * Cp;
        const long ofs = (long)d.Ho * d.Wo * Cp;
        const unsigned yvoff = lane_ok ? (unsigned)(((h0 * d.Wo + w0 + m) * Cp + ce) * 2) : XE_OOB;
        const unsigned fr_bytes = (unsigned)(ofs * 2);
        const int kst = wave_live ? XE_NT : 0;  // stores per emitted frame: one per tile, whether or not its row exists
        auto stored = [&](int to) -> int { return (to >= t0 && to < t1 && !(abl & 16)) ? kst : 0; };

        f32x4 S0[XE_NT], S1[XE_NT], S2[XE_NT];
        const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int l = 0; l < XE_NT; ++l) S0[l] = S1[l] = S2[l] = zero4;

        auto frame = [&](int ti, f32x4 (&P)[XE_NT], f32x4 (&C)[XE_NT], f32x4 (&N)[XE_NT]) {
            if (wave_live && ti >= 0 && ti < Ti && !(abl & 8)) {  // wave-uniform
                int fbo = slot_of(ti) * XE_FRB + lbase0;
                asm volatile("" : "+v"(fbo));
                const char* ta[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) ta[j] = ring + fbo + tapoff[j];
                bf16x8 Bq[2][5];
#pragma unroll
                for (int j = 0; j < 5; ++j) Bq[0][j] = *reinterpret_cast<const bf16x8*>(ta[j]);
                __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
                for (int l = 0; l < XE_NT; ++l) {
                    if (l + 1 < XE_NT) {
#pragma unroll
                        for (int j = 0; j < 5; ++j) Bq[(l + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(ta[j] + (l + 1) * lstep);
                        __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
                    }
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const bf16x8 B = Bq[l & 1][j];
                        P[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[2][j]), B, j == 0 ? C[l] : P[l], 0, 0, 0);
                        C[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[1][j]), B, j == 0 ? N[l] : C[l], 0, 0, 0);
                        N[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[0][j]), B, j == 0 ? zero4 : N[l], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 15, 0);
                }
            } else {
#pragma unroll
                for (int l = 0; l < XE_NT; ++l) {
                    P[l] = C[l];
                    C[l] = N[l];
                    N[l] = zero4;
                }
            }
        };
        auto emit = [&](int ti, f32x4 (&P)[XE_NT]) {  // output frame ti - 1 has now seen frames ti - 2, ti - 1, ti
            const int to = ti - 1;
            if (wave_live && to >= t0 && to < t1 && !(abl & 16)) {
                const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(yclip + (long)to * ofs, 0, fr_bytes, 0x00020000);
                // straight-line over ALL tiles, as in dwmfma.hip: a row below the plane stores out of the descriptor's range and counts nothing,
                // the pool sums are formed whether or not the launch has a row for them, padded channels carry zero scale and bias
                f32x4 sc, bs;
                if (FOLD) {
                    sc = *reinterpret_cast<const f32x4*>(scl);
                    bs = *reinterpret_cast<const f32x4*>(scl + 64);
                } else {
                    sc = f32x4{sc_r[0], sc_r[1], sc_r[2], sc_r[3]};
                    bs = f32x4{bs_r[0], bs_r[1], bs_r[2], bs_r[3]};
                }
#pragma unroll
                for (int l = 0; l < XE_NT; ++l)
                    {
                        float v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = P[l][i] * sc[i] + bs[i];
                        {
                            const bool ok = lane_ok && l < ntl;
#pragma unroll
                            for (int i = 0; i < 4; ++i) psum[i] += ok ? v[i] : 0.0f;
                        }
                        if constexpr (ACT == PASN_ACT_SWISH) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = v[i] * sigmoidf_(v[i]);
                        } else if constexpr (ACT != PASN_ACT_NONE) {
                            act_vec(v, d.act);
                        }
                        if (ACT == -1 && wave_tail) mask_tail(v, d.Cout - ce);  // (run-time activation: sigmoid(0) is not 0)
                        bf16x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(xe_u32x2, o), yrsrc, (int)yvoff, l * ystep * 2, 0);
                    }
            }
        };

        // The steady-state step as ONE scheduling region: the expand conv of frame ti + 1 (LDS reads, two 32x32x16 chains and ~60 vector
        // instructions per pair of tiles) is issued between the stencil MFMAs of frame ti, which leave the vector ALU idle for 16 cycles
        // each -- back to back the two halves left each wave waiting ~60 % of its time with two waves per SIMD (SQ counters: MFMA pipe
        // 31 %, VALU 21 % busy).  Pair p rides under stencil tile p.
        auto fused = [&](int ti, f32x4 (&P)[XE_NT], f32x4 (&C)[XE_NT], f32x4 (&N)[XE_NT]) {
            static_assert(SS != 2 || (NU == 4 && XE_NT == 3), "the pinned schedule below is written for 4 expand tiles under 3 stencil tiles");
            const char* xb = xt + slot_of(ti + 1) * g.xtb;
            char* rb = ring + slot_of(ti + 1) * XE_FRB;
            int fbo = slot_of(ti) * XE_FRB + lbase0;
            asm volatile("" : "+v"(fbo));
            const char* ta[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) ta[j] = ring + fbo + tapoff[j];
            bf16x8 Bq[2][5];
            bf16x8 xf[KS];
            f32x16 acc;
            unsigned Pk[8];
            // the expand epilogue of tile i in ten steps: eight packed convert + ReLU pairs, two (lane swap, border mask, store) halves
            auto e_cvt = [&](int j) {
                xe_f32x2 a = {acc[2 * j], acc[2 * j + 1]};
                if (!folded) {
                    a.x = a.x * scb[32 * ect + acc_row(2 * j, h32)] + scb[64 + 32 * ect + acc_row(2 * j, h32)];
                    a.y = a.y * scb[32 * ect + acc_row(2 * j + 1, h32)] + scb[64 + 32 * ect + acc_row(2 * j + 1, h32)];
                }
                xe_s16x2 mm = __builtin_bit_cast(xe_s16x2, __builtin_convertvector(a, xe_bf16x2));
                mm = __builtin_elementwise_max(mm, xe_s16x2{0, 0});
                Pk[j] = __builtin_bit_cast(unsigned, mm);
            };
            auto e_out = [&](int i, int pr) {
                const int tile = (wave >> 1) + 2 * i;
                const int tl = min(tile, R::TILES - 1);
                const int pos = tl * R::TP + min(c32, R::TP - 1);
                const bool wr = lane_used && tile < R::TILES;
                char* const rp = wr ? rb + (pos * XE_SLOTS + 4 * ect + h32) * 16 : dummy;
                const bool keep = (keepm >> i) & 1u;
                const auto s0 = __builtin_amdgcn_permlane32_swap(Pk[4 * pr + 0], Pk[4 * pr + 2], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(Pk[4 * pr + 1], Pk[4 * pr + 3], false, false);
                xe_u32x4 ou = {s0[0], s1[0], s0[1], s1[1]};
                if (!all_in) {
                    ou.x = keep ? ou.x : 0u;
                    ou.y = keep ? ou.y : 0u;
                    ou.z = keep ? ou.z : 0u;
                    ou.w = keep ? ou.w : 0u;
                }
                *reinterpret_cast<xe_u32x4*>(rp + pr * 32) = ou;
            };
            p_read(0, xb, xf);
#pragma unroll
            for (int j = 0; j < 5; ++j) Bq[0][j] = *reinterpret_cast<const bf16x8*>(ta[j]);
            __builtin_amdgcn_sched_barrier(0);
            p_mma(xf, acc);
            __builtin_amdgcn_sched_barrier(0);
            // 45 stencil MFMAs (16 cycles of the matrix pipe each); behind MFMA g, fenced, the vector work of slot g: expand tile E (0..3)
            // converts in slots 9E+2 .. 9E+5 (its chain was issued 4 slots earlier), the next chain goes out in slot 9E+6 with E's first
            // store half, the second half and the next tile's fragment reads follow in slot 9E+7
#pragma unroll
            for (int gq = 0; gq < 15 * XE_NT; ++gq) {
                const int l = gq / 15, j = (gq % 15) / 3, role = gq % 3;
                const bf16x8 B = Bq[l & 1][j];
                if (role == 0) P[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[2][j]), B, j == 0 ? C[l] : P[l], 0, 0, 0);
                if (role == 1) C[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[1][j]), B, j == 0 ? N[l] : C[l], 0, 0, 0);
                if (role == 2) N[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[0][j]), B, j == 0 ? zero4 : N[l], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int E = gq / 9, r = gq % 9;
                if (gq % 15 == 0 && l + 1 < XE_NT) {  // the next stencil tile's operands (its buffer was last read one tile ago)
#pragma unroll
                    for (int jj = 0; jj < 5; ++jj) Bq[(l + 1) & 1][jj] = *reinterpret_cast<const bf16x8*>(ta[jj] + (l + 1) * lstep);
                }
                if (E < NU) {
                    if (r >= 2 && r <= 5) {
                        e_cvt(2 * (r - 2));
                        e_cvt(2 * (r - 2) + 1);
                    }
                    if (r == 6) {
                        if (E + 1 < NU) p_mma(xf, acc);
                        e_out(E, 0);
                    }
                    if (r == 7) {
                        e_out(E, 1);
                        if (E + 2 < NU) p_read(E + 2, xb, xf);
                    }
                    if (r == 0 && E == 0) p_read(1, xb, xf);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        // prologue: frame t0 - 1 expanded, x rows of frame t0 requested
        issue_x(t0 - 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        produce(t0 - 1);
        issue_x(t0);
#pragma unroll 1
        for (int ti = t0 - 1; ti <= t1; ++ti) {
            // the x rows of frame ti + 1 (requested in step ti - 1, before that step's stores) have landed; behind the barrier everyone's
            // have, frame ti's image (written in step ti - 1) is complete, and nobody still reads the image / x tile of frame ti - 1
            xe_wait_all_but(stored(ti - 2));
            xe_barrier();
            if (SS == 2 && staged(ti + 1) && wave_live && ti >= 0 && ti < Ti && !(abl & 8) && fuse) {  // wave-uniform: the steady state (stride 1: 6 stencil tiles' accumulators leave no room, 66 spilled VGPRs)
                issue_x(ti + 2);  // (its x tile held frame ti: everyone is past produce(ti))
                fused(ti, S0, S1, S2);
                emit(ti, S0);
            } else {
                produce(ti + 1);
                issue_x(ti + 2);
                frame(ti, S0, S1, S2);
                emit(ti, S0);
            }
        }
        __syncthreads();
    }

    if (pool && wave_live) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s = psum[i];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            s += __shfl_xor(s, 8);
            psum[i] = s;
        }
        if (m == 0 && cev) {
            float* pr = pool + ((long)n * g.chunks + chunk) * Cp + ce;
            *reinterpret_cast<f32x4*>(pr) = f32x4{psum[0], psum[1], psum[2], psum[3]};
        }
    }
}

// ---- host -----------------------------------------------------------------------------------------------------------------------------
// de = the expand conv (1x1x1, stride 1, + BN + ReLU), d = the depthwise conv (3x3x3, stride (1,s,s), s = 1 or 2, pad 1) on its output.
XeGeom xe_geom(const pasn_conv_desc& de, const pasn_conv_desc& d, int dtype) {
    XeGeom g{};
    const char* mode = tune("PASN_EXPDW");  // 0: off
    if (mode && mode[0] == '0') return g;
    if (dtype != PASN_BF16) return g;
    const int ss = d.sh;
    // Stride 2 only.  The stride-1 instance (XeR<1>: 6 x 14 outputs from 8 x 16 staged positions = four 32-position expand tiles) was built,
    // passed the same parity test and LOST: 169 / 148 us against 54 + 114 / 55 + 107 for the two launches at 56 x 56, 106 / 94 against
    // 28 + 60 / 28 + 52 at 28 x 28; 10.34 k vs 10.58 k clips/s end to end (profiles/README.md entry 74).  At stride 1 the expanded tensor is
    // no larger than the stencil's output, the unfused pair is not bound by its bytes, and the expand epilogue's ~3 vector instructions per
    // element land on a kernel that is already bound by vector issue.
    // One exception: the SE blocks of the widest stride-1 stage (no Swish in the stencil epilogue, planes >= 56 wide): 148 vs 55 + 107 us,
    // 10.56 -> 10.67 k clips/s end to end (entry 80).  PASN_EXPDW_S1=0: off.
    // With the ReLU-only expand epilogue (entry 81) the Swish (non-SE) blocks of those planes win too, narrowly (10 845 -> 10 875 clips/s).
    const char* s1m = tune("PASN_EXPDW_S1");  // 0: off; 1: the SE blocks only
    const bool s1 = ss == 1 && d.Wo >= 56 && !(s1m && s1m[0] == '0') && (d.act == PASN_ACT_NONE || (d.act == PASN_ACT_SWISH && !(s1m && s1m[0] == '1')));
    if (ss != 2 && ss != 1) return g;
    const bool dw = d.kt == 3 && d.kh == 3 && d.kw == 3 && d.st == 1 && d.sw == ss && d.pt == 1 && d.ph == 1 && d.pw == 1 &&
                    d.To == d.Ti && d.Ho == (d.Hi - 1) / ss + 1 && d.Wo == (d.Wi - 1) / ss + 1 && d.Cin_p == d.Cout_p && d.Cin == d.Cout &&
                    d.Cout_p % 8 == 0;
    const bool ex = de.kt == 1 && de.kh == 1 && de.kw == 1 && !de.pt && !de.ph && !de.pw && de.st == 1 && de.sh == 1 && de.sw == 1 &&
                    !de.in_swish && de.act == PASN_ACT_RELU && de.N == d.N && de.To == d.Ti && de.Ho == d.Hi && de.Wo == d.Wi &&
                    de.Cout == d.Cin && de.Cout_p == d.Cin_p && de.w_frag == 1 && de.w_kc % 16 == 0 && de.w_kc >= de.Cin_p &&
                    de.Cin_p % 8 == 0 && de.w_rows >= ((de.Cout_p + 31) / 32) * 32;
    if (!dw || !ex) return g;
    const int nks = de.w_kc / 16;
    if (nks > 2) return g;  // (the LDS check below admits block widths up to 24 channels: three 16-byte pieces per staged x position)
    if ((long)d.Ti * d.Hi * d.Wi * de.Cin_p * 2 >= (1L << 31) || (long)d.To * d.Ho * d.Wo * d.Cout_p * 2 >= (1L << 31)) return g;
    // round 5: at stride 1 the Toeplitz formulation (x3d_expdw_tz.hip) takes the pair wherever the planes are wide enough to fill its
    // 8 x 14 regions (same coverage rule as the stride-1 use of this kernel; PASN_EXPDW_TZ=0: off)
    if (ss == 1 && s1) {
        xe_geom_tz(g, de, d);
        if (g.tz) {
            g.SS = 1;
            g.chunks = g.tzChunks;
            g.lds = g.tzLds;
            g.ok = 1;
            return g;
        }
    }
    if (ss != 2 && !s1) return g;
    const int NT = ss == 2 ? XeR<2>::NT : XeR<1>::NT, POS = ss == 2 ? XeR<2>::POS : XeR<1>::POS, FRB = ss == 2 ? XeR<2>::FRB : XeR<1>::FRB;
    g.SS = ss;
    g.KS = 2;
    g.XS = (de.Cin_p / 8) | 1;
    g.xtb = ((POS + 3) * g.XS * 16 + 1023) / 1024 * 1024;  // + 3 positions: lanes 29 .. 31 of the last staged row read past it (stride 2)
    g.fuse = !(tune("PASN_EXPDW_FUSE") && tune("PASN_EXPDW_FUSE")[0] == '0');
    g.lds = 2 * FRB + 2 * g.xtb + 512 + 1088;
    // two blocks per CU: block width 24 takes 78.5 KB.  (48 channels -- stage 4's first block -- need 107 KB = one block per CU: built,
    // correct, and slower end to end, 10.37 k vs 10.58 k clips/s: not taken.)
    if (g.lds > 80 * 1024 || (POS * g.XS + 63) / 64 > 4 * XE_NE) return XeGeom{};
    g.CQ = ceil_div(ceil_div(d.Cout_p, 16), 4);
    g.RTH = ceil_div(d.Ho, NT);
    g.RTW = ceil_div(d.Wo, XeR<1>::BW);
    const int regions = g.RTH * g.RTW;
    const int force_tc = tune("PASN_EXPDW_TC") ? atoi(tune("PASN_EXPDW_TC")) : 0;
    const int force_upb = tune("PASN_EXPDW_UPB") ? atoi(tune("PASN_EXPDW_UPB")) : 0;
    double best = 1e30;
    for (int tc = d.To;; tc = (tc + 1) / 2) {
        const int tcu = force_tc ? std::min(force_tc, d.To) : tc;
        const int nT = ceil_div(d.To, tcu), units = nT * regions;
        for (int upb = 1; upb <= units; ++upb) {
            if (force_upb && upb != std::min(force_upb, units)) continue;
            const int chunks = ceil_div(units, upb);
            if (chunks > 64 && upb < units && !force_upb) continue;  // the chunk count is the number of SE partial rows per clip (128: 209 vs 195 us per launch)
            const long blocks = (long)d.N * g.CQ * chunks;
            const double t = (double)ceil_div(blocks, 512L) * (4.0 + upb * (tcu + 3.0));
            if (t < best) {
                best = t;
                g.Tc = tcu;
                g.nT = nT;
                g.upb = upb;
                g.chunks = chunks;
            }
        }
        if (force_tc || tc <= 4) break;
    }
    g.bpc = g.CQ * g.chunks;
    g.abl = tune_dev("PASN_EXPDW_ABL") ? atoi(tune_dev("PASN_EXPDW_ABL")) : 0;
    g.ok = 1;
    return g;
}

int launch_x3d_expdw(const void* x, const void* wa, const float* sa, const float* ba, const float* w, const float* scale, const float* bias,
                     void* y, float* pool, const pasn_conv_desc& de, const pasn_conv_desc& d, const XeGeom& g, hipStream_t s) {
    if (g.tz) {
        PASN_REQUIRE(sa == nullptr, "x3d_expdw (Toeplitz form): norm_a's scale folded into the expand weights (scale_a == NULL)");
        return launch_x3d_expdw_tz(x, wa, ba, w, scale, bias, y, pool, de, d, g, s);
    }
    const dim3 grid((unsigned)(g.bpc * d.N)), block(256);
#define PASN_XEF(KS_, ACT_, SS_, ABL_, FOLD_)                                                                                    \
    do {                                                                                                                         \
        PASN_MAX_LDS(96 * 1024, x3d_expdw_kernel<KS_, ACT_, SS_, ABL_, FOLD_>);                                                  \
        hipLaunchKernelGGL((x3d_expdw_kernel<KS_, ACT_, SS_, ABL_, FOLD_>), grid, block, (size_t)g.lds, s, (const __bf16*)x,     \
                           (const __bf16*)wa, sa, ba, w, scale, bias, (__bf16*)y, pool, d, de.Cin_p, de.w_kc / 16, g);           \
    } while (0)
#define PASN_XE(KS_, ACT_, SS_, ABL_)                          \
    do {                                                       \
        if (sa == nullptr) PASN_XEF(KS_, ACT_, SS_, ABL_, true); \
        else PASN_XEF(KS_, ACT_, SS_, ABL_, false);            \
    } while (0)
#define PASN_XEK(KS_, SS_)                                                          \
    do {                                                                            \
        if (d.act == PASN_ACT_NONE) PASN_XE(KS_, PASN_ACT_NONE, SS_, false);        \
        else if (d.act == PASN_ACT_SWISH) PASN_XE(KS_, PASN_ACT_SWISH, SS_, false); \
        else PASN_XE(KS_, -1, SS_, false);                                          \
    } while (0)
    if (g.SS == 1) {
        PASN_REQUIRE((d.act == PASN_ACT_NONE || d.act == PASN_ACT_SWISH) && !g.abl, "x3d_expdw: stride-1 instances: activation none / Swish");
        if (d.act == PASN_ACT_NONE) PASN_XE(2, PASN_ACT_NONE, 1, false);
        else PASN_XE(2, PASN_ACT_SWISH, 1, false);
        return check_launch("x3d_expdw_kernel (stride 1)");
    }
#ifdef PASN_TUNING
    if (g.abl) {  // timing-ablation instances (113 / 44 spilled VGPRs): -DPASN_TUNING builds only, never in the product library
        PASN_XE(2, -1, 2, true);
        return check_launch("x3d_expdw_kernel (ablation)");
    }
#endif
    PASN_XEK(2, 2);
#undef PASN_XEK
#undef PASN_XE
#undef PASN_XEF
    return check_launch("x3d_expdw_kernel");
}

}  // namespace pasn

// Front half of an X3D block in ONE launch, second formulation (round 5): 1x1x1 expand conv + BN + ReLU -> depthwise 3x3x3 conv, stride 1,
// pad 1, + BN (+ Swish, + squeeze-excite pool partial rows), with the stencil as PER-CHANNEL TOEPLITZ matrix products on a CHANNEL-PLANAR
// image of the expanded activation.  (pytorchvideo's BottleneckTransform conv_a / conv_b as the x3d trunks instantiate them; the reference
// itself ships no X3D -- SURVEY 8a row 5.)
//
// Why a second formulation.  x3d_expdw.hip runs the stencil as v_mfma_f32_16x16x32_bf16 with BLOCK-DIAGONAL weight operands (K = 2 taps x 16
// channels): 1/16 of every MFMA is useful, 15 MFMAs and 5 LDS operand reads per (16 channels x 16 positions) tile and frame, and each 16-cycle
// MFMA holds the SIMD's vector issue for 8 cycles -- on a kernel whose waves are bound by vector issue (profiles/README.md entries 84, 123,
// r04_fwd_pmc_pipes.txt).  The stencil's operand never exists in HBM: it is born in LDS from the expand conv's accumulators, so its layout is
// free.  Here:
//   * the expand conv is v_mfma_f32_16x16x32_bf16 with A = a staged ROW of 16 x positions (8 channels per lane, straight from the channels-last
//     x tile) and B = the block's 16 expand channels (whole K in registers): the accumulator holds, per lane, 4 CONSECUTIVE COLUMNS of ONE
//     channel -- bias as the initial value, ReLU + rounding on packed pairs, border zeroing as a bitwise AND, one ds_write_b64 into the frame
//     image [channel][staged row][16 columns] (no lane swap, no per-element select);
//   * the stencil of channel c is D[out column m][row n] += sum_k A_c[m][k] B[k][n] with K = (2 input rows) x (16 input columns): A_c = the
//     3-diagonal Toeplitz matrix of taps (dt, dh, :) (and (dt, dh + 1, :)) of channel c, B = 16 B per lane of the planar image (8 consecutive
//     columns of one input row).  N = 2 output frames x 8 output rows; 3 dt x 2 row groups = 6 MFMAs and 6 operand reads per 224 outputs of a
//     channel (block-diagonal: 15 + 15 per 224), ONE accumulator of 4 registers, no T-marching accumulator sets and no role rotation: a
//     wave's persistent state is the Toeplitz operands of its 2 channels (48 registers), so 16 waves fit a CU (4 per SIMD; x3d_expdw: 8);
//   * outputs leave through a planar LDS image and ds_read_b64_tr_b16 (gfx950's transposing read): lane = output column, 2 reads = the 8
//     channels of one position = one 16-byte channels-last store.
// Block = 8 waves = 16 expanded channels x (8 x 14 outputs) of one clip, marching along T two output frames per step over a ring of 6 frame
// images; x rows by LDS-DMA two steps ahead; two barriers per step (frame images ready / output image ready).
// Rounding points are those of the two separate launches (expanded activation rounded to bf16, fp32 accumulation, norm_a's scale folded into the
// bf16 expand weights by the host); the summation ORDER of the 27 taps differs from x3d_expdw.hip's, so results agree to fp32 rounding, not bit
// for bit.
#include "common.h"

namespace pasn {

typedef __attribute__((ext_vector_type(4))) unsigned tz_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned tz_u32x2;
typedef __attribute__((ext_vector_type(2))) short tz_s16x2;
typedef __attribute__((ext_vector_type(4))) short tz_s16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 tz_bf16x2;
typedef __attribute__((ext_vector_type(2))) float tz_f32x2;
typedef __attribute__((address_space(3))) void* tz_lds_ptr_t;
typedef __attribute__((address_space(3))) tz_s16x4* tz_lds_s16x4_t;

constexpr int TZ_RT = 8;                       // output rows of a region
constexpr int TZ_BW = 14;                      // output columns of a region
constexpr int TZ_RH = TZ_RT + 2;               // staged rows
constexpr int TZ_POS = TZ_RH * 16;             // staged positions per frame (16 staged columns = one expand MFMA tile)
constexpr int TZ_CHS = 328;                    // bytes per channel of a frame image: 10 rows x 32 B + 8 (82 dwords: 16 channels x ds_write_b64 hit 32 distinct banks)
constexpr int TZ_FS = 5376;                    // bytes per frame image: 16 x 328 rounded up to a multiple of 256 (the two frames a B operand read spans stay bank-disjoint)
constexpr int TZ_NF = 6;                       // frame images in the ring: pairs k, k + 1 (read by step k) and k + 2 (written in step k)
constexpr int TZ_ORS = 40;                     // bytes per row of the output image (10 dwords: 16 rows x ds_write_b64 hit 32 distinct banks; 8-byte aligned for the transposing read)
constexpr int TZ_OCS = 16 * TZ_ORS + 16;       // bytes per channel of the output image
constexpr int TZ_XTB = 8192;                   // bytes per x tile (8 DMA instructions of 1 KiB; 160 positions x XS slots used)
constexpr unsigned TZ_OOB = 0x80000000u;

__device__ __forceinline__ unsigned tz_bf16_bits(float f) {
    const __bf16 b = (__bf16)f;
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ void tz_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// KS32: 32-wide k-steps of the expand conv (1: block width <= 32 channels, 2: <= 64); ACT: the stencil's epilogue (PASN_ACT_NONE / PASN_ACT_SWISH);
// POOL: squeeze-excite partial sums
template <int KS32, int ACT, bool POOL>
__global__ __launch_bounds__(512, 2) void x3d_expdw_tz_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ wa, const float* __restrict__ ba,
                                                              const float* __restrict__ w, const float* __restrict__ scale,
                                                              const float* __restrict__ bias, __bf16* __restrict__ y, float* __restrict__ pool,
                                                              pasn_conv_desc d, int Cin_p, int nks, XeGeom g) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const ring = smem;                                  // [TZ_NF][TZ_FS]
    char* const xt = smem + TZ_NF * TZ_FS;                    // [2 buffers][2 frames][TZ_XTB]
    char* const outi = xt + 4 * TZ_XTB;                       // [16 channels][TZ_OCS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int cgi = lb % g.tzCG, r1 = lb / g.tzCG;
    const int regions = g.tzRTH * g.tzRTW, units = g.tznT * regions;
    const int u = r1 % units, n = r1 / units;
    const int tch = u / regions, reg = u - tch * regions;
    const int rth = reg / g.tzRTW, rtw = reg - rth * g.tzRTW;
    const int t0 = tch * g.tzTc, t1 = min(t0 + g.tzTc, d.To);
    const int h0 = rth * TZ_RT, w0 = rtw * TZ_BW;
    const int Cp = d.Cout_p, Ti = d.Ti, Hi = d.Hi, Wi = d.Wi;
    const int steps = (t1 - t0 + 1) >> 1;                     // output frames t0 + 2 k, t0 + 2 k + 1; input pairs 0 .. steps: frames (t0 - 1 + 2 p, t0 + 2 p)
    const int XS = g.tzXS, pieces = Cin_p >> 3;

    // ---- stencil roles: this wave's two channels; Toeplitz operands in registers for the launch ----
    const int cA = cgi * 16 + 2 * wave;
    const bool wave_live = cA < Cp;
    tz_u32x4 AT[2][3][2];
    float scv[2], bsv[2];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
        const int c = cA + c2;
        const bool chok = c < Cp;
        const int cc = min(c, Cp - 1);
        scv[c2] = (chok && c < d.Cout) ? scale[cc] : 0.0f;  // padded channels: act(0 * acc + 0) = 0 for none / Swish
        bsv[c2] = (chok && c < d.Cout) ? bias[cc] : 0.0f;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                // K group q of this lane: input row r8 + dh, columns 8 (q & 1) .. + 7; g2 = 0: dh = q >> 1 (0, 1); g2 = 1: dh = 2 for q >> 1 == 0, nothing for q >> 1 == 1
                const int dh = g2 == 0 ? (q >> 1) : 2;
                const bool on = chok && m < TZ_BW && (g2 == 0 || (q >> 1) == 0);
                unsigned b[3];
#pragma unroll
                for (int e = 0; e < 3; ++e) b[e] = tz_bf16_bits(w[((dt * 3 + dh) * 3 + e) * Cp + cc]);
                unsigned r[4];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) {
                    const int jl = 8 * (q & 1) + 2 * i2 - m, jh = jl + 1;  // tap index dw = input column - output column
                    const unsigned lo = jl == 0 ? b[0] : jl == 1 ? b[1] : jl == 2 ? b[2] : 0u;
                    const unsigned hi = jh == 0 ? b[0] : jh == 1 ? b[1] : jh == 2 ? b[2] : 0u;
                    r[i2] = on ? (lo | (hi << 16)) : 0u;
                }
                AT[c2][dt][g2] = tz_u32x4{r[0], r[1], r[2], r[3]};
            }
    }
    // output validity of this lane's 4 columns x its row (frame validity joins per step): pool weights
    float mk[4];
    {
        const int r8 = m & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) mk[i] = (4 * q + i < TZ_BW && w0 + 4 * q + i < d.Wo && h0 + r8 < d.Ho) ? 1.0f : 0.0f;
    }
    float psum[2] = {0.0f, 0.0f};
    // B operand base of this lane inside a frame image: channel 2 wave (+ c2), row r8 (+ dh), columns 8 (q & 1) ..
    const int bbase = (2 * wave) * TZ_CHS + (m & 7) * 32 + (q & 1) * 16;
    const int brow0 = (q >> 1) * 32;  // g2 = 0: + dh rows; g2 = 1: + 2 rows
    const int f2 = m >> 3;

    // ---- expand roles: this lane = expand channel m of the block for 4 consecutive staged columns 4 q .. ----
    const int ce = cgi * 16 + m;
    bf16x8 WB[KS32];
    {
        const int ectiles = (Cp + 31) >> 5;
        const int ctile = min(ce >> 5, ectiles - 1), c32 = ce & 31;
#pragma unroll
        for (int k2 = 0; k2 < KS32; ++k2) {
            const int ks16 = 2 * k2 + (q >> 1);
            const bool ok = ks16 < nks && (ce >> 5) < ectiles;
            const bf16x8 v = load_frag<__bf16>(wa + (((long)ctile * nks + min(ks16, nks - 1)) * 64 + (q & 1) * 32 + c32) * 8);
            WB[k2] = ok ? v : zero_frag<__bf16>();
        }
    }
    const float biasE = ce < Cp ? ba[ce] : 0.0f;
    // border zeroing of the expanded activation (the stencil pads the EXPANDED tensor with zeros): columns as AND masks on packed pairs
    unsigned cm01, cm23;
    {
        const bool k0 = (unsigned)(w0 - 1 + 4 * q + 0) < (unsigned)Wi, k1 = (unsigned)(w0 - 1 + 4 * q + 1) < (unsigned)Wi;
        const bool k2 = (unsigned)(w0 - 1 + 4 * q + 2) < (unsigned)Wi, k3 = (unsigned)(w0 - 1 + 4 * q + 3) < (unsigned)Wi;
        cm01 = (k0 ? 0xffffu : 0u) | (k1 ? 0xffff0000u : 0u);
        cm23 = (k2 ? 0xffffu : 0u) | (k3 ? 0xffff0000u : 0u);
    }

    // ---- x DMA role: 16-byte slot s = 64 wave + lane of a frame's x tile -> (staged position, piece) ----
    const long fx = (long)Hi * Wi * Cin_p;
    const unsigned fx_bytes = (unsigned)(fx * 2);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x + (long)n * Ti * fx), 0, (unsigned)Ti * fx_bytes, 0x00020000);
    unsigned goff;
    {
        const int s = wave * 64 + lane;
        const int pos = s / XS, p = s - pos * XS;
        const int rr = pos >> 4, cc = pos & 15;
        const int hi = h0 - 1 + rr, wi = w0 - 1 + cc;
        const bool ok = pos < TZ_POS && p < pieces && (unsigned)hi < (unsigned)Hi && (unsigned)wi < (unsigned)Wi;
        goff = ok ? (unsigned)(((hi * Wi + wi) * Cin_p + p * 8) * 2) : TZ_OOB;
    }
    // pair p = frames (t0 - 1 + 2 p, t0 + 2 p); x buffer p & 1; ring slots (2 p) % 6, + 1
    auto issue_pair = [&](int p) {
        char* dst = xt + (p & 1) * 2 * TZ_XTB + wave * 1024;
#pragma unroll
        for (int fs = 0; fs < 2; ++fs) {
            const int f = t0 - 1 + 2 * p + fs;
            if (f >= 0 && f < Ti)  // wave-uniform (frames outside the clip: zero images, made by the row mask)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (tz_lds_ptr_t)(dst + fs * TZ_XTB), 16, (int)goff, (int)((unsigned)f * fx_bytes), 0, 0);
        }
    };
    auto expand_tile = [&](int p, int e) {  // row tile e of pair p: frame e / 10, staged row e % 10
        const int fs = e >= TZ_RH ? 1 : 0, rr = e - fs * TZ_RH;
        const int f = t0 - 1 + 2 * p + fs;
        const char* xb = xt + ((p & 1) * 2 + fs) * TZ_XTB + (rr * 16 + m) * XS * 16;
        bf16x8 xf[KS32];
#pragma unroll
        for (int k2 = 0; k2 < KS32; ++k2) xf[k2] = *reinterpret_cast<const bf16x8*>(xb + min(4 * k2 + q, pieces - 1) * 16);
        f32x4 acc = {biasE, biasE, biasE, biasE};
#pragma unroll
        for (int k2 = 0; k2 < KS32; ++k2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[k2], WB[k2], acc, 0, 0, 0);
        const bool rowok = f >= 0 && f < Ti && (unsigned)(h0 - 1 + rr) < (unsigned)Hi;  // wave-uniform
        const unsigned rmask = rowok ? 0xffffffffu : 0u;
        tz_s16x2 p0 = __builtin_bit_cast(tz_s16x2, __builtin_convertvector(tz_f32x2{acc[0], acc[1]}, tz_bf16x2));
        tz_s16x2 p1 = __builtin_bit_cast(tz_s16x2, __builtin_convertvector(tz_f32x2{acc[2], acc[3]}, tz_bf16x2));
        p0 = __builtin_elementwise_max(p0, tz_s16x2{0, 0});  // ReLU on the rounded pair (a bf16 is negative iff it is negative as an int16)
        p1 = __builtin_elementwise_max(p1, tz_s16x2{0, 0});
        const tz_u32x2 o = {__builtin_bit_cast(unsigned, p0) & (cm01 & rmask), __builtin_bit_cast(unsigned, p1) & (cm23 & rmask)};
        const int slot = (2 * p) % TZ_NF + fs;
        *reinterpret_cast<tz_u32x2*>(ring + slot * TZ_FS + m * TZ_CHS + rr * 32 + q * 8) = o;
    };
    auto expand_pair = [&](int p) {
        expand_tile(p, wave);
        expand_tile(p, wave + 8);
        if (wave < 2 * TZ_RH - 16) expand_tile(p, wave + 16);  // wave-uniform
    };

    // ---- output roles: 16-lane group G = (row 2 wave + (G >> 1) & 1 ..., channel half): see the store phase ----
    const long oframe = (long)d.Ho * d.Wo * Cp;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(y + (long)n * d.To * oframe, 0, (unsigned)(d.To * oframe * 2), 0x00020000);
    const int G = tid >> 4, l16 = tid & 15;
    const int on16 = G >> 1, og = G & 1;          // row of the 16 (frame on16 >> 3, row on16 & 7), channels 8 og .. of the block
    const int tr_off = (8 * og + (l16 >> 2)) * TZ_OCS + on16 * TZ_ORS + (l16 & 3) * 8;
    const bool ost = l16 < TZ_BW && w0 + l16 < d.Wo && h0 + (on16 & 7) < d.Ho && cgi * 16 + 8 * og < Cp;
    const unsigned ooff = (unsigned)((((h0 + (on16 & 7)) * d.Wo + w0 + l16) * Cp + cgi * 16 + 8 * og) * 2);

    // ---- prologue: pairs 0 and 1 expanded, pair 2 requested ----
    issue_pair(0);
    if (steps >= 1) issue_pair(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    expand_pair(0);
    if (steps >= 1) expand_pair(1);
    __syncthreads();
    if (steps >= 2) issue_pair(2);

#pragma unroll 1
    for (int k = 0; k < steps; ++k) {
        const int t = t0 + 2 * k;
        // the x rows of pair k + 2 (requested one step ago, before that step's store) have landed; behind the barrier everyone's have, the
        // frame images of pairs k, k + 1 are complete, and nobody still reads the output image or the x tiles of pair k + 1
        if (k == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        tz_barrier();
        if (k + 3 <= steps) issue_pair(k + 3);
        if (k + 2 <= steps) expand_pair(k + 2);
        if (wave_live) {
            const int sb = (2 * k) % TZ_NF;
            int so[3];
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                int s = sb + dt + f2;
                s = s >= TZ_NF ? s - TZ_NF : s;
                so[dt] = s * TZ_FS + bbase;
            }
            const float fv = (t + f2 < t1) ? 1.0f : 0.0f;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                bf16x8 B[3][2];
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    const char* bp = ring + so[dt] + c2 * TZ_CHS;
                    B[dt][0] = *reinterpret_cast<const bf16x8*>(bp + brow0);
                    B[dt][1] = *reinterpret_cast<const bf16x8*>(bp + 64);
                }
                f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AT[c2][dt][0]), B[dt][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AT[c2][dt][1]), B[dt][1], acc, 0, 0, 0);
                }
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[i] * scv[c2] + bsv[c2];
                if (POOL) psum[c2] += fv * (v[0] * mk[0] + v[1] * mk[1] + v[2] * mk[2] + v[3] * mk[3]);
                if constexpr (ACT == PASN_ACT_SWISH) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = v[i] * sigmoidf_(v[i]);
                }
                const tz_bf16x2 o0 = __builtin_convertvector(tz_f32x2{v[0], v[1]}, tz_bf16x2);
                const tz_bf16x2 o1 = __builtin_convertvector(tz_f32x2{v[2], v[3]}, tz_bf16x2);
                *reinterpret_cast<tz_u32x2*>(outi + (2 * wave + c2) * TZ_OCS + m * TZ_ORS + q * 8) =
                    tz_u32x2{__builtin_bit_cast(unsigned, o0), __builtin_bit_cast(unsigned, o1)};
            }
        }
        tz_barrier();
        // store phase: 16-lane group = (output row of the 16, 8 channels); lane = output column.  Two transposing reads deliver channels
        // 8 og .. + 3 and + 4 .. + 7 of this lane's column: one 16-byte channels-last store.  (EXEC is all ones here, as the instruction needs.)
        {
            const tz_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tz_lds_s16x4_t)(outi + tr_off));
            const tz_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tz_lds_s16x4_t)(outi + tr_off + 4 * TZ_OCS));
            const tz_u32x2 ua = __builtin_bit_cast(tz_u32x2, a), ub = __builtin_bit_cast(tz_u32x2, b);
            const int to = t + (on16 >> 3);
            const unsigned off = (ost && to < t1) ? ooff : TZ_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(tz_u32x4{ua.x, ua.y, ub.x, ub.y}, yrsrc, (int)off, (int)((unsigned)to * (unsigned)(oframe * 2)), 0);
        }
    }

    if (POOL && pool && wave_live) {
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            float s = psum[c2];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
            psum[c2] = s;
        }
        if (lane == 0 && cA < Cp) {
            float* pr = pool + ((long)n * g.tzChunks + u) * Cp + cA;
            pr[0] = psum[0];
            pr[1] = psum[1];
        }
    }
}

// ---- host -----------------------------------------------------------------------------------------------------------------------------
// Fills the tz* fields of g (g.tz = 1) when the Toeplitz kernel covers the pair; called by xe_geom after its own checks passed.
void xe_geom_tz(XeGeom& g, const pasn_conv_desc& de, const pasn_conv_desc& d) {
    g.tz = 0;
    const char* mode = tune("PASN_EXPDW_TZ");
    if (mode && mode[0] == '0') return;
    const char* fold = tune("PASN_EXPDW_FOLD");
    if (fold && fold[0] == '0') return;                       // the kernel takes norm_a folded into the expand weights (scale_a == NULL)
    if (d.sh != 1 || d.sw != 1) return;                       // stride 1 only
    if (d.act != PASN_ACT_NONE && d.act != PASN_ACT_SWISH) return;
    const int nks = de.w_kc / 16;
    if (nks > 2) return;                                      // block width <= 32 channels in this round's instances (KS32 = 1)
    const int pieces = de.Cin_p / 8;
    g.tzXS = pieces | 1;
    if (TZ_POS * g.tzXS > 512) return;                        // one DMA instruction per wave and frame covers the x tile
    g.tzCG = ceil_div(d.Cout_p, 16);
    g.tzRTH = ceil_div(d.Ho, TZ_RT);
    g.tzRTW = ceil_div(d.Wo, TZ_BW);
    const int force_tc = tune("PASN_EXPDW_TC") ? atoi(tune("PASN_EXPDW_TC")) : 0;
    g.tzTc = force_tc > 0 ? std::min(force_tc, (int)d.To) : d.To;
    g.tznT = ceil_div(d.To, g.tzTc);
    g.tzChunks = g.tznT * g.tzRTH * g.tzRTW;
    if (g.tzChunks > 64 && !force_tc) return;                 // SE partial rows per clip the consumers sum (see x3d_expdw.hip)
    g.tzLds = TZ_NF * TZ_FS + 4 * TZ_XTB + 16 * TZ_OCS;
    g.tz = 1;
}

int launch_x3d_expdw_tz(const void* x, const void* wa, const float* ba, const float* w, const float* scale, const float* bias, void* y, float* pool,
                        const pasn_conv_desc& de, const pasn_conv_desc& d, const XeGeom& g, hipStream_t s) {
    const dim3 grid((unsigned)((long)d.N * g.tzCG * g.tznT * g.tzRTH * g.tzRTW)), block(512);
#define PASN_TZ(ACT_, POOL_)                                                                                                       \
    do {                                                                                                                         \
        PASN_MAX_LDS(80 * 1024, x3d_expdw_tz_kernel<1, ACT_, POOL_>);                                                            \
        hipLaunchKernelGGL((x3d_expdw_tz_kernel<1, ACT_, POOL_>), grid, block, (size_t)g.tzLds, s, (const __bf16*)x,            \
                           (const __bf16*)wa, ba, w, scale, bias, (__bf16*)y, pool, d, de.Cin_p, de.w_kc / 16, g);               \
    } while (0)
    if (d.act == PASN_ACT_SWISH) {
        if (pool) PASN_TZ(PASN_ACT_SWISH, true);
        else PASN_TZ(PASN_ACT_SWISH, false);
    } else {
        if (pool) PASN_TZ(PASN_ACT_NONE, true);
        else PASN_TZ(PASN_ACT_NONE, false);
    }
#undef PASN_TZ
    return check_launch("x3d_expdw_tz_kernel");
}

}  // namespace pasn

// Front half of an X3D block in ONE launch, second formulation (round 5): 1x1x1 expand conv + BN + ReLU -> depthwise 3x3x3 conv, stride 1,
// pad 1, + BN (+ Swish, + squeeze-excite pool partial rows), with the stencil as PER-CHANNEL TOEPLITZ matrix products on a CHANNEL-PLANAR
// image of the expanded activation.  (pytorchvideo's BottleneckTransform conv_a / conv_b as the x3d trunks instantiate them; the reference
// itself ships no X3D -- SURVEY 8a row 5.)
//
// Why a second formulation.  x3d_expdw.hip runs the stencil as v_mfma_f32_16x16x32_bf16 with BLOCK-DIAGONAL weight operands (K = 2 taps x 16
// channels): 1/16 of every MFMA is useful, 15 MFMAs and 5 LDS operand reads per (16 channels x 16 positions) tile and frame, each 16-cycle
// MFMA holds the SIMD's vector issue for 8 cycles, and the three T-marching accumulator sets plus 60 registers of operands keep it at two
// waves per SIMD -- on a kernel whose waves are bound by vector issue and wait 60 % of the time (profiles/README.md entries 84, 123,
// r04_fwd_pmc_pipes.txt).  The stencil's operand never exists in HBM: it is born in LDS from the expand conv's accumulators, so its layout
// is free.  Here:
//   * the expand conv is v_mfma_f32_16x16x32_bf16 with A = a staged ROW of 16 x positions (8 channels per lane, loaded straight from global
//     memory one step ahead: no x tile in LDS, no DMA) and B = the block's 16 expand channels (whole K in registers): the accumulator holds,
//     per lane, 4 CONSECUTIVE COLUMNS of ONE channel -- bias as the initial value, ReLU + rounding on packed pairs, border zeroing as a
//     bitwise AND, one ds_write_b64 into the frame image [channel][column tile][staged row][16 columns] (no lane swap, no select per element);
//   * the stencil of channel c is D[out column m][row n] += sum_k A_c[m][k] B[k][n] with K = two (frame, row) shifts x 16 input columns:
//     A_c = 3-diagonal Toeplitz matrices of tap rows (dt, dh, :) of channel c with norm_b's scale folded in, B = 16 B per lane of the planar
//     image (8 consecutive columns of one input row); N = 2 output frames x 8 output rows.  The 9 (dt, dh) rows pair up into 5 MFMAs and 5
//     operand reads per 224 outputs of a channel (block-diagonal: 15 + 15 per 224 when three frames are counted), ONE accumulator of 4
//     registers initialised with norm_b's bias: no T-marching accumulator sets, no role rotation, no scale / bias arithmetic.  A wave's
//     persistent state is the Toeplitz operands of its 2 channels (40 registers), so 16 waves fit a CU (4 per SIMD; x3d_expdw: 8);
//   * outputs leave through a planar LDS image and ds_read_b64_tr_b16 (gfx950's transposing read): lane = output column, 2 reads = the 8
//     channels of one position = one 16-byte channels-last store.
// Block = 8 waves = 16 expanded channels x (8 x 28 outputs) of one clip, marching along T two output frames per step over a ring of 4 frame
// images (2 pairs); two barriers per step: [stencil of pairs k, k + 1 -> output image] | [store, expand pair k + 2 over pair k, request the x
// rows of pair k + 3].
// Rounding points: the expanded activation is rounded to bf16 (as by the two separate launches), norm_a's scale meets the expand weights and
// norm_b's scale the stencil weights before THEIR rounding to bf16 (the separate launches apply the scales in fp32 after the MFMAs), fp32
// accumulation in a different order from x3d_expdw.hip's: results agree with both to one bf16 ulp of the output, not bit for bit.
#include "common.h"

namespace pasn {

typedef __attribute__((ext_vector_type(4))) unsigned tz_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned tz_u32x2;
typedef __attribute__((ext_vector_type(2))) short tz_s16x2;
typedef __attribute__((ext_vector_type(4))) short tz_s16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 tz_bf16x2;
typedef __attribute__((ext_vector_type(2))) float tz_f32x2;
typedef __attribute__((address_space(3))) tz_s16x4* tz_lds_s16x4_t;

constexpr int TZ_RT = 8;                       // output rows of a region
constexpr int TZ_BW = 14;                      // output columns of a column tile
constexpr int TZ_CT = 2;                       // column tiles of a region (28 output columns)
constexpr int TZ_RH = TZ_RT + 2;               // staged rows
constexpr int TZ_TS = TZ_RH * 32;              // bytes per (channel, column tile) of a frame image: 10 rows x 16 columns
constexpr int TZ_CHS = TZ_CT * TZ_TS + 16;     // bytes per channel of a frame image: a multiple of 16 -- the B operand reads are ds_read_b128, and a 16-byte LDS access
                                               // off its alignment is replayed at 64 cycles (the first version, at + 8, spent 78 % of its time in the LDS: SQ_LDS_IDX_ACTIVE
                                               // 21 per LDS instruction); 164 dwords: the 16 channels of the expand's ds_write_b64 fall on 8 bank pairs, 2-way
constexpr int TZ_FS = 16 * TZ_CHS;             // bytes per frame image: 10496, a multiple of 256 (the two frames a B operand read spans stay bank-disjoint)
static_assert(TZ_FS % 256 == 0 && TZ_CHS % 16 == 0, "frame images: 16-byte aligned channel planes, 256-byte aligned frames");
constexpr int TZ_NF = 4;                       // frame images in the ring: pairs k, k + 1
constexpr int TZ_ORS = 40;                     // bytes per row of the output image (10 dwords: the 16 rows of a ds_write_b64 hit 32 distinct banks; 8-byte aligned for the transposing read)
constexpr int TZ_OTS = 16 * TZ_ORS;            // bytes per (channel, column tile) of the output image
constexpr int TZ_OCS = TZ_CT * TZ_OTS + 16;    // bytes per channel of the output image
constexpr int TZ_ET = 2 * TZ_RH * TZ_CT;       // expand tiles (staged row x column tile) per pair of frames: 40
constexpr int TZ_EW = TZ_ET / 8;               // ... per wave: 5
constexpr unsigned TZ_OOB = 0x80000000u;

__device__ __forceinline__ unsigned tz_bf16_bits(float f) {
    const __bf16 b = (__bf16)f;
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ void tz_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef PASN_TUNING
// shader-clock stamps of block 0 / wave 0 (tuning builds, PASN_TZ_STAMPS=1; tools/tz_bench.py prints them): [0] start, [1] operands built, [2] prologue
// done, then per step 6: stencil done, barrier passed, stores issued, expand done, loads issued, barrier passed
__device__ long long tz_stamps[2 + 6 * 10];
#define TZ_STAMP(i) do { if (g.abl && (int)blockIdx.x == g.abl - 1 && threadIdx.x == 0 && (i) < 62) tz_stamps[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define TZ_STAMP(i) do { } while (0)
#endif

// The 9 (dt, dh) tap rows of a channel in 5 MFMAs: K half h of MFMA j carries tap row 2 j + h (row 9 = none)
__device__ __forceinline__ constexpr int tz_row(int j, int h) { return 2 * j + h; }

// KS32: 32-wide k-steps of the expand conv (1: block width <= 32 channels); ACT: the stencil's epilogue (PASN_ACT_NONE / PASN_ACT_SWISH);
// POOL: squeeze-excite partial sums
template <int KS32, int ACT, bool POOL>
__global__ __launch_bounds__(512, 4) void x3d_expdw_tz_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ wa, const float* __restrict__ ba,
                                                              const float* __restrict__ w, const float* __restrict__ scale,
                                                              const float* __restrict__ bias, __bf16* __restrict__ y, float* __restrict__ pool,
                                                              pasn_conv_desc d, int Cin_p, int nks, XeGeom g) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const ring = smem;                                  // [TZ_NF][TZ_FS]
    char* const outi = smem + TZ_NF * TZ_FS;                  // [16 channels][TZ_OCS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int cgi = lb % g.tzCG, r1 = lb / g.tzCG;
    const int regions = g.tzRTH * g.tzRTW, units = g.tznT * regions;
    const int u = r1 % units, n = r1 / units;
    const int tch = u / regions, reg = u - tch * regions;
    const int rth = reg / g.tzRTW, rtw = reg - rth * g.tzRTW;
    const int t0 = tch * g.tzTc, t1 = min(t0 + g.tzTc, d.To);
    const int h0 = rth * TZ_RT, w0 = rtw * (TZ_CT * TZ_BW);
    const int Cp = d.Cout_p, Ti = d.Ti, Hi = d.Hi, Wi = d.Wi;
    const int steps = (t1 - t0 + 1) >> 1;                     // output frames t0 + 2 k, t0 + 2 k + 1; input pairs 0 .. steps: frames (t0 - 1 + 2 p, t0 + 2 p)
    const int pieces = Cin_p >> 3;
    TZ_STAMP(0);

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
    // (scalars, not arrays: the column tile of a wave's expand tile is wave-uniform but not a compile-time constant, and an array indexed by it goes to scratch)
    auto colmask = [&](int wb) -> unsigned { return ((unsigned)wb < (unsigned)Wi ? 0xffffu : 0u) | ((unsigned)(wb + 1) < (unsigned)Wi ? 0xffff0000u : 0u); };
    const unsigned cm00 = colmask(w0 - 1 + 4 * q), cm01 = colmask(w0 - 1 + 4 * q + 2);
    const unsigned cm10 = colmask(w0 - 1 + TZ_BW + 4 * q), cm11 = colmask(w0 - 1 + TZ_BW + 4 * q + 2);
    // this lane's x row piece for expand tile (row 0, column tile ct): staged column m, channels 8 q ..
    auto xo = [&](int wi) -> unsigned { return (unsigned)wi < (unsigned)Wi ? (unsigned)((wi * Cin_p + min(q, pieces - 1) * 8) * 2) : TZ_OOB; };
    const unsigned xoff0 = xo(w0 - 1 + m), xoff1 = xo(w0 - 1 + TZ_BW + m);
    const long fx = (long)Hi * Wi * Cin_p;
    const unsigned fx_bytes = (unsigned)(fx * 2), rx_bytes = (unsigned)(Wi * Cin_p * 2);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x + (long)n * Ti * fx), 0, (unsigned)Ti * fx_bytes, 0x00020000);
    // expand tile i of this wave in pair p: e = 5 wave + i of the pair's 40 (frame, staged row, column tile)
    tz_u32x4 xq[TZ_EW];
#pragma unroll
    for (int i = 0; i < TZ_EW; ++i) xq[i] = tz_u32x4{0u, 0u, 0u, 0u};
    auto load_tile = [&](int p, int i) {
        const int e = wave * TZ_EW + i;
        const int fs = e / (TZ_RH * TZ_CT), rem = e - fs * (TZ_RH * TZ_CT);
        const int rr = rem >> 1, ct = rem & 1;
        const int f = t0 - 1 + 2 * p + fs, hi = h0 - 1 + rr;
        if (f >= 0 && f < Ti && (unsigned)hi < (unsigned)Hi)  // wave-uniform (rows / frames outside: zeroed by the row mask whatever the registers hold)
            xq[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(ct ? xoff1 : xoff0), (int)((unsigned)f * fx_bytes + (unsigned)hi * rx_bytes), 0);
    };
    auto expand_tile = [&](int p, int i) {
        const int e = wave * TZ_EW + i;
        const int fs = e / (TZ_RH * TZ_CT), rem = e - fs * (TZ_RH * TZ_CT);
        const int rr = rem >> 1, ct = rem & 1;
        const int f = t0 - 1 + 2 * p + fs;
        f32x4 acc = {biasE, biasE, biasE, biasE};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xq[i]), WB[0], acc, 0, 0, 0);
        const bool rowok = f >= 0 && f < Ti && (unsigned)(h0 - 1 + rr) < (unsigned)Hi;  // wave-uniform
        const unsigned rmask = rowok ? 0xffffffffu : 0u;
        tz_s16x2 p0 = __builtin_bit_cast(tz_s16x2, __builtin_convertvector(tz_f32x2{acc[0], acc[1]}, tz_bf16x2));
        tz_s16x2 p1 = __builtin_bit_cast(tz_s16x2, __builtin_convertvector(tz_f32x2{acc[2], acc[3]}, tz_bf16x2));
        p0 = __builtin_elementwise_max(p0, tz_s16x2{0, 0});  // ReLU on the rounded pair (a bf16 is negative iff it is negative as an int16)
        p1 = __builtin_elementwise_max(p1, tz_s16x2{0, 0});
        const unsigned c0m = ct ? cm10 : cm00, c1m = ct ? cm11 : cm01;
        const tz_u32x2 o = {__builtin_bit_cast(unsigned, p0) & (c0m & rmask), __builtin_bit_cast(unsigned, p1) & (c1m & rmask)};
        const int slot = ((2 * p) & (TZ_NF - 1)) + fs;
        *reinterpret_cast<tz_u32x2*>(ring + slot * TZ_FS + m * TZ_CHS + ct * TZ_TS + rr * 32 + q * 8) = o;
    };
    auto load_pair = [&](int p) {
#pragma unroll
        for (int i = 0; i < TZ_EW; ++i) load_tile(p, i);
    };
    auto expand_pair = [&](int p) {
#pragma unroll
        for (int i = 0; i < TZ_EW; ++i) expand_tile(p, i);
    };
    // the x rows of pairs 0 AND 1 are requested before the stencil operands are built (the second set of registers is free until they are)
    tz_u32x4 xq1[TZ_EW];
#pragma unroll
    for (int i = 0; i < TZ_EW; ++i) xq1[i] = tz_u32x4{0u, 0u, 0u, 0u};
    load_pair(0);
    {
#pragma unroll
        for (int i = 0; i < TZ_EW; ++i) {
            const int e = wave * TZ_EW + i;
            const int fs = e / (TZ_RH * TZ_CT), rem = e - fs * (TZ_RH * TZ_CT);
            const int rr = rem >> 1, ct = rem & 1;
            const int f = t0 + 1 + fs, hi = h0 - 1 + rr;
            if (f >= 0 && f < Ti && (unsigned)hi < (unsigned)Hi)
                xq1[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(ct ? xoff1 : xoff0), (int)((unsigned)f * fx_bytes + (unsigned)hi * rx_bytes), 0);
        }
    }

    // ---- stencil roles: this wave's two channels; Toeplitz operands in registers for the launch ----
    // FOLDB (the instances without pool sums): norm_b's scale meets the stencil weights BEFORE their rounding to bf16 and its bias is the
    // accumulator's initial value -- no scale / bias arithmetic in the epilogue.  The squeeze-excite instances keep the scale in fp32 behind the
    // MFMAs: a weight rounded after scaling shifts a channel's outputs by up to one bf16 ulp of each tap SYSTEMATICALLY, which the pool sum over
    // 50 k positions does not average away (and they have no Swish epilogue to make room for).
    constexpr bool FOLDB = !POOL;
    const int cA = cgi * 16 + 2 * wave;
    const bool wave_live = cA < Cp;
    tz_u32x4 AT[2][5];
    float bsv[2], scv[2];
    {
        // operand of lane (m, q), K group q: tap row (dt, dh) = 2 j + (q >> 1), input columns 8 (q & 1) .. + 7; output column m takes taps
        // (w0, w1, w2) at input columns m, m + 1, m + 2: the 48-bit string w0 | w1 | w2 shifted to slot m - 8 (q & 1) of the lane's eight
        // (all 54 weights requested before the first is used -- one scalar-load round trip, not one per operand -- and the 128-bit shift
        // branch-free: the first version waited for six scalar loads and took a divergent branch per operand, 9-12 k cycles per block)
        const int sh = 16 * (m - 8 * (q & 1));                // bit position of the string's first tap in the lane's 128 bits: -128 .. 240
        // (vector loads on purpose -- an opaque zero joins the wave-uniform index: as 54 scalar loads the weights sat in 160 spilled SGPRs)
        int vz = 0;
        asm volatile("" : "+v"(vz));
        float wv[2][27];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int cc = min(cA + c2, d.Cout - 1) + vz;
#pragma unroll
            for (int e = 0; e < 27; ++e) wv[c2][e] = w[e * Cp + cc];
        }
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int c = cA + c2;
            const bool chok = c < d.Cout;                     // padded channels: zero operands and zero bias -> act(0) = 0 for none / Swish
            const int cc = min(c, d.Cout - 1);
            const float sc = chok ? scale[cc] : 0.0f;
            scv[c2] = sc;
            bsv[c2] = chok ? bias[cc] : 0.0f;
            const float sw = FOLDB ? sc : 1.0f;
            const bool on = chok && m < TZ_BW;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                unsigned long long T[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = min(tz_row(j, h), 8);
                    const unsigned long long b0 = tz_bf16_bits(wv[c2][row * 3 + 0] * sw), b1 = tz_bf16_bits(wv[c2][row * 3 + 1] * sw),
                                             b2 = tz_bf16_bits(wv[c2][row * 3 + 2] * sw);
                    T[h] = tz_row(j, h) < 9 ? (b0 | (b1 << 16) | (b2 << 32)) : 0ull;
                }
                const unsigned long long Tl = on ? ((q >> 1) ? T[1] : T[0]) : 0ull;
                // (Tl << sh) as two 64-bit halves, shift amounts clamped into range and the out-of-range cases selected away
                const unsigned long long lo = (sh >= 0 && sh < 64) ? Tl << (sh & 63) : (sh < 0 && sh > -64) ? Tl >> ((-sh) & 63) : 0ull;
                const unsigned long long hi = (sh >= 64 && sh < 128) ? Tl << ((sh - 64) & 63) : (sh > 0 && sh < 64) ? Tl >> ((64 - sh) & 63) : 0ull;
                AT[c2][j] = tz_u32x4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
            }
        }
    }
    // B operand of MFMA j for this lane: frame t + f2 + dt - 1, row r8 + dh, columns 8 (q & 1) ..: offset inside the image + the frame's
    // number relative to the step's first frame
    const int f2 = m >> 3;
    int bpk[5];  // bits 0 .. 19: the offset inside the image, bits 20 ..: the frame's number relative to the step's first frame (dt + f2)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int row = min(tz_row(j, q >> 1), 8);            // (the empty half of the last MFMA reads tap row 8's operand: finite values times zero)
        bpk[j] = ((row / 3 + f2) << 20) | ((2 * wave) * TZ_CHS + ((m & 7) + row % 3) * 32 + (q & 1) * 16);
    }
    // Pool sums are taken from the ROUNDED outputs (v_dot2c_f32_bf16 of the packed pairs the store needs anyway with 1 / 0 pairs: 2 instructions
    // per tile where fp32 masks cost 5 and 8 registers): the rounding errors are unbiased and the squeeze-excite mean runs over 50 k positions
    // per clip -- 1e-5 of the mean's scale.  (Swish + pool, which no X3D block has, pools the pre-activation in fp32.)  On a region that lies
    // inside the plane the weights are ones except for output columns 14, 15 of a tile (lanes q = 3, second pair): ONE register; a region cut by
    // the plane's border computes its column / row weights where it uses them.
    const bool ragged = w0 + TZ_CT * TZ_BW > d.Wo || h0 + TZ_RT > d.Ho;  // wave-uniform
    unsigned* const ptab = reinterpret_cast<unsigned*>(outi + 16 * TZ_OCS);  // [column tile][pair][64 lanes]: the weights of a cut region
    if (POOL && wave == 0) {
#pragma unroll
        for (int ct = 0; ct < TZ_CT; ++ct)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                unsigned v = 0;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int col = 4 * q + 2 * h + i;
                    if (col < TZ_BW && w0 + ct * TZ_BW + col < d.Wo && h0 + (m & 7) < d.Ho) v |= 0x3f80u << (16 * i);
                }
                ptab[(ct * 2 + h) * 64 + lane] = v;
            }
    }
    const unsigned mk23 = q < 3 ? 0x3f803f80u : 0u;
    float psum[2] = {0.0f, 0.0f};

    // ---- output roles: 16-lane group = (output row of the 16, column tile, 8 channels), lane = output column; two passes of 32 groups ----
    const long oframe = (long)d.Ho * d.Wo * Cp;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(y + (long)n * d.To * oframe, 0, (unsigned)(d.To * oframe * 2), 0x00020000);
    // (item I = G + 32 pass: channel half I & 1, column tile (I >> 1) & 1, row of the 16 I >> 2 -- pass 1 is pass 0's row in the second frame)
    const int G = tid >> 4, l16 = tid & 15;
    int tr_off;
    unsigned ooff;
    {
        const int og = G & 1, ct = (G >> 1) & 1, n8 = G >> 2;
        tr_off = (8 * og + (l16 >> 2)) * TZ_OCS + ct * TZ_OTS + n8 * TZ_ORS + (l16 & 3) * 8;
        const int ho = h0 + n8, wo = w0 + ct * TZ_BW + l16;
        const bool ok = l16 < TZ_BW && wo < d.Wo && ho < d.Ho && cgi * 16 + 8 * og < Cp;
        ooff = ok ? (unsigned)(((ho * d.Wo + wo) * Cp + cgi * 16 + 8 * og) * 2) : TZ_OOB;
    }

    // ---- prologue: pairs 0 and 1 expanded, the x rows of pair 2 requested ----
    TZ_STAMP(1);
    expand_pair(0);
#pragma unroll
    for (int i = 0; i < TZ_EW; ++i) xq[i] = xq1[i];
    expand_pair(1);
    if (steps >= 2) load_pair(2);
    tz_barrier();
    TZ_STAMP(2);

#pragma unroll 1
    for (int k = 0; k < steps; ++k) {
        const int t = t0 + 2 * k;
        // ---- phase 1: the stencil of output frames t, t + 1 from pairs k, k + 1 -> output image ----
        if (wave_live) {
            int so[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) so[j] = ((2 * k + (bpk[j] >> 20)) & (TZ_NF - 1)) * TZ_FS + (bpk[j] & 0xfffff);
            const bool tailf = t + 1 >= t1;                   // wave-uniform: the step's second output frame does not exist (odd chunk)
            const unsigned fm = (tailf && f2) ? 0u : 0xffffffffu;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int ct = 0; ct < TZ_CT; ++ct) {
                    const char* bp = ring + c2 * TZ_CHS + ct * TZ_TS;
                    bf16x8 B[5];
#pragma unroll
                    for (int j = 0; j < 5; ++j) B[j] = *reinterpret_cast<const bf16x8*>(bp + so[j]);
                    const float a0 = FOLDB ? bsv[c2] : 0.0f;
                    f32x4 acc = {a0, a0, a0, a0};
#pragma unroll
                    for (int j = 0; j < 5; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AT[c2][j]), B[j], acc, 0, 0, 0);
                    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
                    if (!FOLDB) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = v[i] * scv[c2] + bsv[c2];
                    }
                    if constexpr (ACT == PASN_ACT_SWISH) {
                        if (POOL) {
                            const unsigned w01 = ptab[(ct * 2) * 64 + lane] & fm, w23 = ptab[(ct * 2 + 1) * 64 + lane] & fm;
#pragma unroll
                            for (int i = 0; i < 4; ++i) psum[c2] += (((i < 2 ? w01 : w23) >> (16 * (i & 1))) & 0xffffu) ? v[i] : 0.0f;
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = v[i] * sigmoidf_(v[i]);
                    }
                    const tz_bf16x2 o0 = __builtin_convertvector(tz_f32x2{v[0], v[1]}, tz_bf16x2);
                    const tz_bf16x2 o1 = __builtin_convertvector(tz_f32x2{v[2], v[3]}, tz_bf16x2);
                    if (POOL && ACT != PASN_ACT_SWISH) {
                        unsigned w01 = 0x3f803f80u, w23 = mk23;
                        if (ragged || tailf) {  // wave-uniform
                            w01 = ptab[(ct * 2) * 64 + lane] & fm;
                            w23 = ptab[(ct * 2 + 1) * 64 + lane] & fm;
                        }
                        psum[c2] = __builtin_amdgcn_fdot2_f32_bf16(o0, __builtin_bit_cast(tz_bf16x2, w01), psum[c2], false);
                        psum[c2] = __builtin_amdgcn_fdot2_f32_bf16(o1, __builtin_bit_cast(tz_bf16x2, w23), psum[c2], false);
                    }
                    *reinterpret_cast<tz_u32x2*>(outi + (2 * wave + c2) * TZ_OCS + ct * TZ_OTS + m * TZ_ORS + q * 8) =
                        tz_u32x2{__builtin_bit_cast(unsigned, o0), __builtin_bit_cast(unsigned, o1)};
                }
        }
        TZ_STAMP(3 + 6 * k);
        tz_barrier();  // the output image is complete; nobody reads pair k's frame images any more
        TZ_STAMP(4 + 6 * k);
        // ---- phase 2: store (two transposing reads deliver channels 8 og .. + 3 and + 4 .. + 7 of this lane's column: one 16-byte
        // channels-last store; EXEC is all ones here, as the instruction needs), expand pair k + 2 over pair k, request pair k + 3 ----
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const tz_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tz_lds_s16x4_t)(outi + tr_off + ps * 8 * TZ_ORS));
            const tz_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tz_lds_s16x4_t)(outi + tr_off + ps * 8 * TZ_ORS + 4 * TZ_OCS));
            const tz_u32x2 ua = __builtin_bit_cast(tz_u32x2, a), ub = __builtin_bit_cast(tz_u32x2, b);
            const int to = t + ps;
            // (the frame's offset rides in the VECTOR offset, soffset = 0: behind a 16-byte buffer store with an SGPR soffset the compiler puts no
            // wait state before a VALU write to the store's data registers, and gfx950 needs one -- dw_tz.hip met it; tools/store_hazard_scan.py)
            const unsigned off = to < t1 ? ooff + (unsigned)to * (unsigned)(oframe * 2) : TZ_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(tz_u32x4{ua.x, ua.y, ub.x, ub.y}, yrsrc, (int)off, 0, 0);
        }
        TZ_STAMP(5 + 6 * k);
        // (each tile's x row for the NEXT pair is requested as soon as the tile's registers are free)
        if (k + 2 <= steps) {
#pragma unroll
            for (int i = 0; i < TZ_EW; ++i) {
                expand_tile(k + 2, i);
                if (k + 3 <= steps) load_tile(k + 3, i);
            }
        }
        TZ_STAMP(6 + 6 * k);
        TZ_STAMP(7 + 6 * k);
        tz_barrier();  // pair k + 2's frame images are complete; everyone is done with the output image
        TZ_STAMP(8 + 6 * k);
    }

    if (POOL && pool && wave_live) {
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            float s = psum[c2];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
            psum[c2] = s;
        }
        if (lane == 0) {
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
    g.tzXS = 0;
    g.tzCG = ceil_div(d.Cout_p, 16);
    g.tzRTH = ceil_div(d.Ho, TZ_RT);
    g.tzRTW = ceil_div(d.Wo, TZ_CT * TZ_BW);
    const int force_tc = tune("PASN_EXPDW_TC") ? atoi(tune("PASN_EXPDW_TC")) : 0;
    g.tzTc = force_tc > 0 ? std::min(force_tc, (int)d.To) : d.To;
    g.tznT = ceil_div(d.To, g.tzTc);
    g.tzChunks = g.tznT * g.tzRTH * g.tzRTW;
    if (g.tzChunks > 64 && !force_tc) return;                 // SE partial rows per clip the consumers sum (see x3d_expdw.hip)
    g.tzLds = TZ_NF * TZ_FS + 16 * TZ_OCS + 1024;           // + the pool-weight table of a region cut by the plane's border
    g.abl = tune_dev("PASN_TZ_STAMPS") ? std::max(1, atoi(tune_dev("PASN_TZ_STAMPS"))) : 0;  // 1 + the block that leaves stamps
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

#ifdef PASN_TUNING
extern "C" int pasn_debug_tz_stamps(long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pasn::tz_stamps), sizeof(long long) * 62); }
#endif

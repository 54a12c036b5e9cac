// Depthwise 3x3x3 conv, stride 1, pad 1, + BN (+ Swish, + squeeze-excite pool partial rows) on planes at most 14 x 14 as PER-CHANNEL TOEPLITZ
// matrix products on a channel-planar LDS image (round 5): x3d_expdw_tz.hip's stencil for layers whose input arrives CHANNELS-LAST from HBM --
// the stride-1 blocks of the 14 x 14 stage, where the expand conv is a launch of its own (pytorchvideo's BottleneckTransform conv_b as the x3d
// trunks instantiate it).
//
// dwmfma.hip runs the same layers with block-diagonal operands at two waves per SIMD (15 MFMAs + 5 operand reads per 16-channel tile and frame,
// three T-marching accumulator sets, 226-253 registers).  The Toeplitz form needs 8 consecutive COLUMNS of one channel per lane; round 4 priced the
// transposition of a channels-last tensor at two passes per element and left it.  gfx950's ds_read_b64_tr_b16 does it in one: the rows of the
// plane arrive by LDS-DMA as they lie ([position][16 channels], 32 bytes per position), and a 16-lane group's transposing read of 4 positions x
// 16 channels hands lane i the 4 consecutive columns of channel i -- one ds_write_b64 into the planar frame image.  One read + one write per
// 4 positions x 16 channels, no vector arithmetic; zeros outside the image come from the DMA's range check.
//
// Block = 8 waves = 16 channels x one clip (a T chunk of it), the whole plane: the block's two 16 x 14-output tiles are ROW BANDS (tile ct =
// output rows 8 ct .. 8 ct + 7, every column).  It marches along T two output frames per step over a ring of 4 frame images, two barriers per
// step: [stencil of pairs k, k + 1 -> output image] | [store, transpose pair k + 2 over pair k, request the rows of pair k + 3].
// Stencil, operands, output path: x3d_expdw_tz.hip's (5 MFMAs per 224 outputs of a channel, ONE accumulator of 4 registers, 40 registers of
// Toeplitz operands per wave, ds_read_b64_tr_b16 on the way out, four waves per SIMD).  norm's scale is folded into the operands where no pool
// sums are taken (a weight rounded after scaling: results agree with dwmfma.hip to one bf16 ulp of the output, not bit for bit).
// Wider planes in regions of 8 x 28 outputs (two column tiles) were built and measured: 28 x 28 x 108 53 / 50 us against dwmfma.hip's 52 / 47,
// 56 x 56 x 54 117 / 114 against 103 / 86 -- a block's start-up (two memory round trips before its first MFMA) is a third of its time and
// 896 blocks on 512 slots run two rounds; that form is gone (profiles/README.md, round 5).
#include "common.h"

namespace pasn {

typedef __attribute__((ext_vector_type(4))) unsigned tz_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned tz_u32x2;
typedef __attribute__((ext_vector_type(4))) short tz_s16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 tz_bf16x2;
typedef __attribute__((ext_vector_type(2))) float tz_f32x2;
typedef __attribute__((address_space(3))) tz_s16x4* tz_lds_s16x4_t;
typedef __attribute__((address_space(3))) void* tz_lds_ptr_t;

constexpr int TZ_RT = 8;                       // output rows of a band
constexpr int TZ_BW = 14;                      // output columns of a band (the plane's width, at most)
constexpr int TZ_CT = 2;                       // bands of a plane
constexpr int TZ_RH = TZ_RT + 2;               // staged rows of a band
constexpr int TZ_TS = TZ_RH * 32;              // bytes per (channel, band) of a frame image: 10 rows x 16 columns
constexpr int TZ_CHS = TZ_CT * TZ_TS + 16;     // bytes per channel of a frame image (16-byte aligned planes: x3d_expdw_tz.hip)
constexpr int TZ_FS = 16 * TZ_CHS;             // bytes per frame image, a multiple of 256
static_assert(TZ_FS % 256 == 0 && TZ_CHS % 16 == 0, "frame images: 16-byte aligned channel planes, 256-byte aligned frames");
constexpr int TZ_NF = 4;                       // frame images in the ring: pairs k, k + 1
constexpr int TZ_ORS = 40;                     // bytes per row of the output image (32 used: conflict-free ds_write_b64, x3d_expdw_tz.hip)
constexpr int TZ_OTS = 16 * TZ_ORS;            // bytes per (channel, band) of the output image
constexpr int TZ_OCS = TZ_CT * TZ_OTS + 16;    // bytes per channel of the output image
constexpr int TZ_PR = 16;                      // staged rows of the plane: -1 .. 14
constexpr int TZ_RAWROW = 16 * 32;             // bytes per raw row: 16 positions (columns -1 .. 14) x 16 channels; one LDS-DMA instruction = two rows
constexpr int TZ_RAW = 2 * TZ_PR * TZ_RAWROW;  // raw rows of one pair of frames
constexpr int TZ_NR = 2;                       // LDS-DMA instructions per wave and pair: rows 2 wave, 2 wave + 1 of either frame
static_assert(TZ_NF * TZ_FS + 16 * TZ_OCS + TZ_RAW + 1024 <= 80 * 1024, "two blocks per CU");
constexpr unsigned TZ_OOB = 0x80000000u;

__device__ __forceinline__ unsigned tz_bf16_bits(float f) {
    const __bf16 b = (__bf16)f;
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ void tz_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifdef PASN_TUNING
// shader-clock stamps of one block's wave 0 (tuning builds, PASN_TZ_STAMPS = 1 + block; tools/dwtz_bench.py prints them): [0] start, [1] operands
// built, [2] prologue done, then five per step: stencil done, rows landed, barrier passed + stores issued, pair transposed + next requested,
// barrier passed
__device__ long long dt_stamps[62];
#define TZ_STAMP(i) do { if (g.abl && (int)blockIdx.x == g.abl - 1 && threadIdx.x == 0 && (i) < 62) dt_stamps[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define TZ_STAMP(i) do { } while (0)
#endif
// ds_read_b64_tr_b16 as inline assembly: through the builtin the compiler cannot tell the read from the cells a pending `buffer_load ... lds`
// writes and puts s_waitcnt vmcnt(0) in front of EVERY transposing read -- the rows requested for the next step awaited on the spot (first
// version: 60.7 us where the block-diagonal kernel takes 55.4).  The waits for the reads' own results are part of the statement.
template <int O0, int O1>
__device__ __forceinline__ void tz_read_tr2(unsigned addr, tz_u32x2& a, tz_u32x2& b) {
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b)
                 : "v"(addr), "n"(O0), "n"(O1)
                 : "memory");
}
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void tz_read_tr4(unsigned addr, tz_u32x2& a, tz_u32x2& b, tz_u32x2& c, tz_u32x2& e) {
    asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%5\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\tds_read_b64_tr_b16 %2, %4 offset:%7\n\t"
                 "ds_read_b64_tr_b16 %3, %4 offset:%8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(e)
                 : "v"(addr), "n"(O0), "n"(O1), "n"(O2), "n"(O3)
                 : "memory");
}
__device__ __forceinline__ unsigned tz_lds_addr(const void* p) { return (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)p; }

// The 9 (dt, dh) tap rows of a channel in 5 MFMAs: K half h of MFMA j carries tap row 2 j + h (row 9 = none)
__device__ __forceinline__ constexpr int tz_row(int j, int h) { return 2 * j + h; }

// ACT: the epilogue (PASN_ACT_NONE / PASN_ACT_SWISH); POOL: squeeze-excite partial sums
template <int ACT, bool POOL>
__global__ __launch_bounds__(512, 4) void dwconv3d_tz_kernel(const __bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                                                             const float* __restrict__ bias, __bf16* __restrict__ y, float* __restrict__ pool,
                                                             pasn_conv_desc d, DtGeom g) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const ring = smem;                                  // [TZ_NF][TZ_FS]
    char* const outi = smem + TZ_NF * TZ_FS;                  // [16 channels][TZ_OCS]
    char* const raw = outi + 16 * TZ_OCS;                     // the rows of ONE pair of frames as they lie in memory: [2 frames][TZ_PR][TZ_RAWROW]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int cgi = lb % g.CG, r1 = lb / g.CG;
    const int u = r1 % g.nT, n = r1 / g.nT;                   // T chunk, clip
    const int t0 = u * g.Tc, t1 = min(t0 + g.Tc, d.To);
    const int Cp = d.Cout_p, Ti = d.Ti, Hi = d.Hi, Wi = d.Wi;
    TZ_STAMP(0);
    const int steps = (t1 - t0 + 1) >> 1;                     // output frames t0 + 2 k, t0 + 2 k + 1; input pairs 0 .. steps: frames (t0 - 1 + 2 p, t0 + 2 p)

    // ---- staging roles: two LDS-DMA instructions per wave and pair of frames, 1 KB each: instruction i = rows 2 wave, 2 wave + 1 of the plane's
    // 16 staged rows (-1 .. 14) of frame i, lane -> (row lane >> 5, staged column (lane & 31) >> 1, channel half lane & 1), 16 bytes each;
    // positions outside the image and channels beyond the tensor carry an out-of-range offset: the hardware writes zeros.  The wave that
    // requested a row transposes it.
    const long fx = (long)Hi * Wi * Cp;
    const unsigned fx_bytes = (unsigned)(fx * 2);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x + (long)n * Ti * fx), 0, (unsigned)Ti * fx_bytes, 0x00020000);
    unsigned xoff;
    {
        const int ch = cgi * 16 + (lane & 1) * 8;
        const int hi = 2 * wave + (lane >> 5) - 1, wi = ((lane & 31) >> 1) - 1;
        xoff = ((unsigned)hi < (unsigned)Hi && (unsigned)wi < (unsigned)Wi && ch < Cp) ? (unsigned)(((hi * Wi + wi) * Cp + ch) * 2) : TZ_OOB;
    }
    auto frame_ok = [&](int p, int fs) -> bool {  // wave-uniform: frame fs of pair p exists
        const int f = t0 - 1 + 2 * p + fs;
        return f >= 0 && f < Ti;
    };
    auto dma_rows = [&](int p, char* base) {
#pragma unroll
        for (int fs = 0; fs < TZ_NR; ++fs)
            if (frame_ok(p, fs))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (tz_lds_ptr_t)(base + (fs * TZ_PR + 2 * wave) * TZ_RAWROW), 16, (int)xoff,
                                                         (int)((unsigned)(t0 - 1 + 2 * p + fs) * fx_bytes), 0, 0);
    };
    // transposing read of a raw row's 16 positions x 16 channels: lane 4 q' + p of a 16-lane group supplies the address of position 4 group + q',
    // channels 4 p ..; lane i receives channel i's values at the group's 4 positions = 4 consecutive columns of channel i -> one ds_write_b64
    // into the planar frame image.
    const int tr_in = (4 * q + (m >> 2)) * 32 + (m & 3) * 8;
    const unsigned raw_addr = tz_lds_addr(raw), outi_addr = tz_lds_addr(outi), tmp_addr = tz_lds_addr(ring + 2 * TZ_FS);
    // One pair of frames: the four transposing reads of this wave's rows (one statement, one wait: read by read, each with its own wait, the
    // pass was a chain of LDS round trips), the requests that refill the rows with pair pd (pd < 0: none) -- they are in registers -- and the
    // rows into the frame images: band 0 holds staged rows 0 .. 9 of the 16, band 1 rows 8 .. 17 as its rows 0 .. 9.  Frames outside the
    // clip: zeros whatever the raw rows hold.  (EXEC is all ones at every transposing read: the branches around them are wave-uniform.)
    auto stage_pair = [&](int p, unsigned base_addr, int pd) {
        tz_u32x2 uv[TZ_NR][2];
        tz_read_tr4<0, TZ_RAWROW, TZ_PR * TZ_RAWROW, (TZ_PR + 1) * TZ_RAWROW>(base_addr + 2 * wave * TZ_RAWROW + tr_in, uv[0][0], uv[0][1], uv[1][0], uv[1][1]);
        if (pd >= 0) dma_rows(pd, raw);
#pragma unroll
        for (int fs = 0; fs < TZ_NR; ++fs) {
            const unsigned rmask = frame_ok(p, fs) ? 0xffffffffu : 0u;
            char* img = ring + (((2 * p) & (TZ_NF - 1)) + fs) * TZ_FS + m * TZ_CHS + q * 8;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = 2 * wave + j;  // staged row of the plane
                const tz_u32x2 v = tz_u32x2{uv[fs][j].x & rmask, uv[fs][j].y & rmask};
                if (r < TZ_RH) *reinterpret_cast<tz_u32x2*>(img + r * 32) = v;
                if (r >= TZ_RT) *reinterpret_cast<tz_u32x2*>(img + TZ_TS + (r - TZ_RT) * 32) = v;
            }
        }
    };
    // Both pairs of the prologue are requested at once: pair 1 into the raw rows, pair 0 into the (still empty) frame images 2, 3 of the ring
    dma_rows(0, ring + 2 * TZ_FS);
    dma_rows(1, raw);

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
    // per tile where fp32 masks cost 5 and 8 registers): the rounding errors are unbiased and the squeeze-excite mean runs over 3 k positions
    // per clip and channel.  (Swish + pool, which no X3D block has, pools the pre-activation in fp32.)  The 1 / 0 weights of a lane's four
    // outputs (columns 14, 15 of a band, rows below the plane, the missing second frame of an odd chunk's last step) come from a 1 KB table.
    unsigned* const ptab = reinterpret_cast<unsigned*>(raw + TZ_RAW);  // [band][pair][64 lanes]
    if (POOL && wave == 0) {
#pragma unroll
        for (int ct = 0; ct < TZ_CT; ++ct)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                unsigned v = 0;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int col = 4 * q + 2 * h + i;
                    if (col < d.Wo && ct * TZ_RT + (m & 7) < d.Ho) v |= 0x3f80u << (16 * i);
                }
                ptab[(ct * 2 + h) * 64 + lane] = v;
            }
    }
    float psum[2] = {0.0f, 0.0f};

    // ---- output roles: 16-lane group = (output row of a band, band, 8 channels), lane = output column; the two frames of a step in turn ----
    const long oframe = (long)d.Ho * d.Wo * Cp;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(y + (long)n * d.To * oframe, 0, (unsigned)(d.To * oframe * 2), 0x00020000);
    const int G = tid >> 4, l16 = tid & 15;
    int tr_off;
    unsigned ooff;
    {
        const int og = G & 1, ct = (G >> 1) & 1, n8 = G >> 2;
        tr_off = (8 * og + (l16 >> 2)) * TZ_OCS + ct * TZ_OTS + n8 * TZ_ORS + (l16 & 3) * 8;
        const int ho = ct * TZ_RT + n8, wo = l16;
        const bool ok = wo < d.Wo && ho < d.Ho && cgi * 16 + 8 * og < Cp;
        ooff = ok ? (unsigned)(((ho * d.Wo + wo) * Cp + cgi * 16 + 8 * og) * 2) : TZ_OOB;
    }

    // (the operands are built while the prologue's rows are in flight: without this the compiler sinks the build into the first step)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int j = 0; j < 5; ++j) asm volatile("" : "+v"(AT[c2][j]));
    TZ_STAMP(1);
    // ---- prologue: pair 0 transposed out of the ring's far half, then pair 1 over it; the rows of pair 2 requested ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    auto zero_band_tail = [&](int sl0) {  // band 1's staged rows 8, 9 lie below every plane and are never staged: zeros once (finite values
                                          // under the pool sums' zero weights)
#pragma unroll
        for (int sl = sl0; sl < sl0 + 2; ++sl)
            if (tid < 16 * 8) *reinterpret_cast<tz_u32x2*>(ring + sl * TZ_FS + (tid >> 3) * TZ_CHS + TZ_TS + TZ_RT * 32 + (tid & 7) * 8) = tz_u32x2{0u, 0u};
    };
    zero_band_tail(0);
    tz_barrier();  // (the rows are read a barrier behind their wait, as in the loop)
    stage_pair(0, tmp_addr, -1);
    tz_barrier();  // everyone has read its rows of pair 0: frame images 2, 3 may be written
    zero_band_tail(2);
    stage_pair(1, raw_addr, steps >= 2 ? 2 : -1);
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
                        const unsigned w01 = ptab[(ct * 2) * 64 + lane] & fm, w23 = ptab[(ct * 2 + 1) * 64 + lane] & fm;
                        psum[c2] = __builtin_amdgcn_fdot2_f32_bf16(o0, __builtin_bit_cast(tz_bf16x2, w01), psum[c2], false);
                        psum[c2] = __builtin_amdgcn_fdot2_f32_bf16(o1, __builtin_bit_cast(tz_bf16x2, w23), psum[c2], false);
                    }
                    *reinterpret_cast<tz_u32x2*>(outi + (2 * wave + c2) * TZ_OCS + ct * TZ_OTS + m * TZ_ORS + q * 8) =
                        tz_u32x2{__builtin_bit_cast(unsigned, o0), __builtin_bit_cast(unsigned, o1)};
                }
        }
        TZ_STAMP(3 + 5 * k);
        // The rows of pair k + 2 were requested one step ago.  The wait stands BEFORE the barrier and the rows are read behind it
        // (cdna_hip_programming.md: read a staged buffer one phase after the wait that retires it).
        if (k + 2 <= steps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TZ_STAMP(4 + 5 * k);
        tz_barrier();  // the output image is complete; nobody reads pair k's frame images any more
        // ---- phase 2: the store -- two transposing reads deliver channels 8 og .. + 3 and + 4 .. + 7 of this lane's column, one 16-byte
        // channels-last store per output frame -- then pair k + 2 transposed over pair k and the rows of pair k + 3 requested.  (Measured
        // both ways, 14 x 14 x 216, Swish / pool: stores first 29.5 / 26.9 us; requests first, then stores, then the writes of the frame
        // images 31.2 / 28.3 with vmcnt(0), 31.9 / 28.9 with a counted vmcnt(2).) ----
        tz_u32x2 ua[2], ub[2];
        tz_read_tr4<0, 4 * TZ_OCS, 8 * TZ_ORS, 8 * TZ_ORS + 4 * TZ_OCS>(outi_addr + tr_off, ua[0], ub[0], ua[1], ub[1]);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int to = t + ps;
            // (the frame's offset rides in the VECTOR offset, soffset = 0: behind a 16-byte buffer store with an SGPR soffset the compiler puts
            // no wait state before a VALU write to the store's data registers -- gfx950 needs one: tools/store_hazard_scan.py, profiles/README.md)
            const unsigned off = to < t1 ? ooff + (unsigned)to * (unsigned)(oframe * 2) : TZ_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(tz_u32x4{ua[ps].x, ua[ps].y, ub[ps].x, ub[ps].y}, yrsrc, (int)off, 0, 0);
        }
        TZ_STAMP(5 + 5 * k);
        if (k + 2 <= steps) stage_pair(k + 2, raw_addr, k + 3 <= steps ? k + 3 : -1);
        TZ_STAMP(6 + 5 * k);
        tz_barrier();  // pair k + 2's frame images are complete; everyone is done with the output image
        TZ_STAMP(7 + 5 * k);
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
            float* pr = pool + ((long)n * g.nT + u) * Cp + cA;
            pr[0] = psum[0];
            pr[1] = psum[1];
        }
    }
}

// ---- host -----------------------------------------------------------------------------------------------------------------------------
DtGeom dw_tz_geom(const pasn_conv_desc& d, int dtype) {
    DtGeom g{};
    if (dtype != PASN_BF16) return g;
    // The stride-1 stencils of planes 9 .. 14 wide and at most 14 high (the 14 x 14 stage: one block = 16 channels of a whole clip; narrower
    // planes leave the 14-column tiles half empty and stay with dwmfma.hip).  PASN_DW_TZ=0: off.
    if (tune("PASN_DW_TZ") && tune("PASN_DW_TZ")[0] == '0') return g;
    if (tune("PASN_DWMFMA") && tune("PASN_DWMFMA")[0] == '0') return g;  // "no matrix-core stencil": the VALU stencil's tests and A/B runs
    const bool shape = d.kt == 3 && d.kh == 3 && d.kw == 3 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 1 && d.ph == 1 && d.pw == 1 &&
                       d.To == d.Ti && d.Ho == d.Hi && d.Wo == d.Wi && d.Cin_p == d.Cout_p && d.Cin == d.Cout && d.Cout_p % 8 == 0;
    if (!shape || (d.act != PASN_ACT_NONE && d.act != PASN_ACT_SWISH)) return g;
    if (d.Wo > TZ_BW || d.Wo < 9 || d.Ho > TZ_BW) return g;   // (16 staged rows: plane rows -1 .. 14)
    if ((long)d.Ti * d.Hi * d.Wi * d.Cin_p * 2 >= (1L << 31)) return g;  // one clip per buffer descriptor
    g.CG = ceil_div(d.Cout_p, 16);
    const int force_tc = tune("PASN_DWMFMA_TC") ? atoi(tune("PASN_DWMFMA_TC")) : 0;
    g.Tc = force_tc > 0 ? std::min(force_tc, (int)d.To) : d.To;
    g.nT = ceil_div(d.To, g.Tc);                              // = SE partial rows per clip
    if (g.nT > 64 && !force_tc) return DtGeom{};
    // One round of blocks (two per CU, 256 CUs) or none: a block's first MFMA stands behind two memory round trips, and a second round pays
    // them again -- 64 clips of the 14 x 14 stage (the paired training pass: 896 blocks) take 58-64 us here against dwmfma.hip's 54-58, 32
    // clips (448 blocks) 29.6 / 26.2 against 33.5 / 29.6 (profiles/README.md entries 143, 149).  PASN_DW_TZ=1: whatever the block count.
    if ((long)d.N * g.CG * g.nT > 512 && !(tune("PASN_DW_TZ") && tune("PASN_DW_TZ")[0] == '1')) return DtGeom{};
    g.lds = TZ_NF * TZ_FS + 16 * TZ_OCS + TZ_RAW + 1024;      // + the pool-weight table
    g.abl = tune_dev("PASN_TZ_STAMPS") ? std::max(1, atoi(tune_dev("PASN_TZ_STAMPS"))) : 0;  // 1 + the block that leaves stamps
    g.ok = 1;
    return g;
}

int launch_dw_tz(const void* x, const float* w, const float* scale, const float* bias, void* y, float* pool, const pasn_conv_desc& d, const DtGeom& g,
                 hipStream_t s) {
    const dim3 grid((unsigned)((long)d.N * g.CG * g.nT)), block(512);
#define PASN_DT(ACT_, POOL_)                                                                                                          \
    do {                                                                                                                            \
        PASN_MAX_LDS(80 * 1024, dwconv3d_tz_kernel<ACT_, POOL_>);                                                                   \
        hipLaunchKernelGGL((dwconv3d_tz_kernel<ACT_, POOL_>), grid, block, (size_t)g.lds, s, (const __bf16*)x, w, scale, bias, (__bf16*)y, pool, d, g); \
    } while (0)
    if (d.act == PASN_ACT_SWISH) {
        if (pool) PASN_DT(PASN_ACT_SWISH, true);
        else PASN_DT(PASN_ACT_SWISH, false);
    } else {
        if (pool) PASN_DT(PASN_ACT_NONE, true);
        else PASN_DT(PASN_ACT_NONE, false);
    }
#undef PASN_DT
    return check_launch("dwconv3d_tz_kernel");
}

}  // namespace pasn

#ifdef PASN_TUNING
extern "C" int pasn_debug_dt_stamps(long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pasn::dt_stamps), sizeof(long long) * 62); }
#endif

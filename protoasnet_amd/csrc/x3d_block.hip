// One launch per X3D residual block body (bf16):  depthwise 3x3x3 stencil (conv_b + norm_b + Swish) -> project conv (conv_c + norm_c +
// residual + ReLU) -> the NEXT block's expand conv (conv_a + norm_a + ReLU), the 2.25x-wide stencil output never leaving the CU.
//
// Why this cut (D -> P -> E, not E -> D -> P): the tensor that crosses launches is then the expanded activation, read WITH its halo from
// L2 / the Infinity Cache -- no recomputation of the expand conv on halo positions, no re-staging of the block input per channel quad --
// and the launch has the shape of two kernels that already exist and are parity-green: dwmfma.hip's matrix-core stencil (block-diagonal
// bf16 weight operands, frame images staged by LDS-DMA into a 2-slot ring, three accumulator sets rotating through the MFMAs) and
// pwconv_ws.hip's chained pair (MFMA 32x32x16 over an LDS tile, lane-swap epilogue, 16-byte stores).  What the fusion removes per block:
// the stencil's output tensor (written and read: 2 x 2.25 C bytes per position), one or two launches (stage 4: stencil + pair -> one;
// stages 3 and 5: expand + stencil + project -> one) and their ramps.  Squeeze-excite blocks need the clip-wide pool between the stencil
// and the project conv and stay on the two-launch path (stencil with pool partial rows | gate prologue + project (+ expand)).
//
// Work split.  Tile = TF consecutive frames x one region of BH x BW outputs (dwmfma's regions: <= 7 x 14, or <= 8 x 8 with two rows per
// 16-lane position tile on planes at most 8 wide), R = TF BH BW <= 224 rows.  A block (8 waves, one per CU: ~100-145 KB of LDS) walks a
// run of consecutive tiles (T-adjacent chunks of one region first: the halo frames of the next tile are in L2).  Per tile:
//   D  for every 64-channel quad of the inner width: march over the TF + 2 input frames of the region (+ 1-cell border, zero-filled by
//      the DMA outside the image / the clip), wave = (16-channel tile of the quad, half of the region's position tiles); an output frame that
//      has seen its three input frames is scaled, shifted, Swish'd, rounded to bf16 and written to the tile's stencil-output image
//      `dwact` [R][inner channels] in LDS -- the steps of all quads form ONE pipeline (the first frame of quad q + 1 is requested under the
//      last MFMAs of quad q);
//   P  project conv from `dwact`: unit = (32 output channels, two 32-row tiles), weight fragments straight from global memory
//      (fragment-major, L2-resident, software-pipelined two steps ahead), epilogue = lane swap -> scale / bias + residual (16-byte loads from
//      the block input) + ReLU -> 16-byte store of the block output AND the same bf16 values into `xt` [R][C] in LDS;
//   E  the next block's expand conv from `xt` (same unit shape), scale / bias + ReLU -> 16-byte stores of the expanded activation.
// Rounding points are exactly those of the separate launches (stencil output, block output, expanded activation in bf16; fp32 accumulation
// in the same k order): the fused launch is BIT-IDENTICAL to dwconv3d_mfma_kernel + pwconv_ws_kernel (pair), which the tests assert.
#include "common.h"

namespace pasn {

typedef __attribute__((ext_vector_type(4))) unsigned xb_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned xb_u32x2;
typedef __attribute__((address_space(3))) void* xb_lds_ptr_t;

constexpr unsigned XB_OOB = 0x80000000u;

// staged frame image of one quad (dwmfma.hip's layout): region rows x 16 positions x 10 slots of 16 bytes (8 used: 64 channels)
constexpr int xb_tiles(int rpt) { return rpt == 2 ? 4 : 7; }
constexpr int xb_rows(int rpt) { return xb_tiles(rpt) * rpt - 1 + 3; }
constexpr int xb_ni(int rpt) { return (xb_rows(rpt) * 16 * 10 + 63) / 64; }  // 1-KiB DMA instructions per frame: 23 / 25

__device__ __forceinline__ void xb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned xb_bf16_bits(float f) {
    const __bf16 b = (__bf16)f;
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}

struct XbArgs {
    const __bf16* e;          // expanded activation of THIS block [N][T][H][W][Cmp]
    const float* w_dw;        // stencil taps [27][Cmp] fp32
    const float *s_dw, *b_dw; // [Cmp]
    const __bf16* w_c;        // project weights, fragment-major [CTC][KSC][64][8], K zero-padded to KSC steps
    const float *s_c, *b_c;   // [>= 32 CTC]
    const __bf16* res;        // block input [M][Cop]
    __bf16* y;                // block output [M][Cop]
    const __bf16* w_a;        // next expand conv, fragment-major [CTA][KSA][64][8]
    const float *s_a, *b_a;   // [>= 32 CTA]
    __bf16* e_next;           // [M][Cnp]
    int N, T, H, W, Cm, Cmp, Cop, Cnp;
};

template <int RPT, bool EXPAND, int ABL>
__global__ __launch_bounds__(512) void x3d_block_kernel(XbArgs a, XbGeom g) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int NT = xb_tiles(RPT), NTW = (NT + 1) / 2;
    constexpr int RW = 16, SLOTS = 10;
    constexpr int NI = xb_ni(RPT), NE = (NI + 7) / 8;
    constexpr int fbytes = NI * 1024;
    constexpr int lstep = RPT * RW * SLOTS * 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, q = lane >> 4;   // stencil roles
    const int c = lane & 31, h = lane >> 5;   // pointwise roles
    const int ctw = wave & 3, ph = wave >> 2;
    const int abl = ABL ? g.abl : 0;
    char* const ring = smem;                  // [2][fbytes]; dead after the D phase
    char* const xt = smem;                    // [RTn * 32][XPL] slots: aliases the ring
    char* const dwact = smem + g.dw_off;      // [R + 1][DPL] slots (row R: dump row of the lanes that hold no output)
    unsigned* const rowtab = reinterpret_cast<unsigned*>(smem + g.tab_off);  // [RTn * 32] global position of a tile row, ~0u: none
    const int Cmp = a.Cmp, Cop = a.Cop, T = a.T, H = a.H, W = a.W;
    const int DPL = g.DPL, XPL = g.XPL, TF = g.TF, BW = g.BW, BHW = g.BH * g.BW, R = g.R, RTn = g.RTn;
    const int nsteps_q = TF + 2;

    // the pad slots of the two operand images (k beyond the real channels: the weights there are zero, the activations must be finite)
    for (int i = threadIdx.x; i < g.lds_bytes / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = uint4{0u, 0u, 0u, 0u};

    const long fstride = (long)H * W * Cmp;
    const unsigned fr_in_bytes = (unsigned)(fstride * 2);
    const long M = (long)a.N * T * H * W;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.res), 0, (unsigned)(M * Cop * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (unsigned)(M * Cop * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t ersrc = __builtin_amdgcn_make_buffer_rsrc(EXPAND ? a.e_next : a.y, 0, EXPAND ? (unsigned)(M * a.Cnp * 2) : 0u, 0x00020000);

    int tapoff[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int tap9 = min(2 * j + (q >> 1), 8);
        tapoff[j] = ((tap9 / 3) * RW + (tap9 % 3)) * (SLOTS * 16);
    }
    const int mrow = m / BW, mcol = m - mrow * BW;
    const int regions = g.RTH * g.RTW;
    const int lbl = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_end = min(g.tiles, (lbl + 1) * g.tpb);
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 S0[NTW], S1[NTW], S2[NTW];
#pragma unroll
    for (int l = 0; l < NTW; ++l) S0[l] = S1[l] = S2[l] = zero4;
    __syncthreads();

#pragma unroll 1
    for (int tile = lbl * g.tpb; tile < tile_end; ++tile) {
        const int tch = tile % g.nTch, nr = tile / g.nTch;
        const int reg = nr % regions, n = nr / regions;
        const int rth = reg / g.RTW, rtw = reg - rth * g.RTW;
        const int t0 = tch * TF, h0 = rth * g.BH, w0 = rtw * BW;
        // ---- row table of this tile: row r = (frame, region row, region column) -> global position ------------------------------------
        if ((int)threadIdx.x < RTn * 32) {
            const int r = threadIdx.x;
            const int tf = r / BHW, rem = r - tf * BHW;
            const int rh = rem / BW, rw = rem - rh * BW;
            const bool ok = r < R && t0 + tf < T && h0 + rh < H && w0 + rw < W;
            rowtab[r] = ok ? (unsigned)(((n * T + t0 + tf) * H + h0 + rh) * W + w0 + rw) : 0xffffffffu;
        }
        // ================================ D: the stencil of every quad, one step pipeline ==============================================
        unsigned goff[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int slot = (wave + 8 * e) * 64 + lane;
            const int rp = slot / SLOTS, cs = slot - rp * SLOTS;
            const int rr = rp / RW, cc = rp - rr * RW;
            const int hi = h0 - 1 + rr, wi = w0 - 1 + cc;
            const bool ok = wave + 8 * e < NI && rr < g.BH + 2 && cc < BW + 2 && cs < 8 && hi >= 0 && hi < H && wi >= 0 && wi < W;
            goff[e] = ok ? (unsigned)(((hi * W + wi) * Cmp + cs * 8) * 2) : XB_OOB;
        }
        const __bf16* eclip = a.e + (long)n * T * fstride;
        const int total = g.NQ * nsteps_q;
        auto issue = [&](int s) {
            if (s >= total || (abl & 2)) return;
            const int qd = s / nsteps_q, f = s - qd * nsteps_q;
            const int ti = t0 - 1 + f;
            const bool inclip = ti >= 0 && ti < T;  // a frame outside the clip is staged as zeros (every lane out of range): no second code path
            const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(eclip + qd * 64), 0,
                                                                               (unsigned)T * fr_in_bytes - (unsigned)(qd * 128), 0x00020000);
            const unsigned foff = inclip ? (unsigned)ti * fr_in_bytes : 0u;
            char* dst = ring + (s & 1) * fbytes;
#pragma unroll
            for (int e = 0; e < NE; ++e)
                if (wave + 8 * e < NI)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (xb_lds_ptr_t)(dst + (wave + 8 * e) * 1024), 16, (int)(inclip ? goff[e] : XB_OOB), (int)foff, 0, 0);
        };
        const bool lane_pos = m < RPT * BW && w0 + mcol < W;
        const int rows_valid = min(g.BH, H - h0);
        const int lbase0 = ((min(mrow, RPT - 1) * RW + min(mcol, BW - 1)) * SLOTS + 2 * ctw + (q & 1)) * 16;
        xb_u32x4 A[3][5];
        float sc[4], bs[4];
        bool wave_live = false;
        int ce = 0;
        issue(0);
#pragma unroll 1
        for (int s = 0; s < total; ++s) {
            const int qd = s / nsteps_q, f = s - qd * nsteps_q;
            if (f == 0) {  // this quad's weight operands and epilogue constants (dwmfma.hip: block-diagonal, built once per quad)
                const int c0 = (qd * 4 + ctw) * 16;
                wave_live = c0 < Cmp;
                const int cch = c0 + m;
                const bool mine = ((m >> 3) == (q & 1)) && cch < Cmp;
                const int dwsel = (m & 7) >> 1, sh = (m & 1) * 16;
                const int ccl = min(cch, Cmp - 1);
                float wv[3][5];
#pragma unroll
                for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                    for (int j = 0; j < 5; ++j) wv[kt][j] = a.w_dw[(kt * 9 + min(2 * j + (q >> 1), 8)) * Cmp + ccl];
                ce = c0 + 4 * q;
                const bool cev = ce < Cmp;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sc[i] = (cev && ce + i < a.Cm) ? a.s_dw[ce + i] : 0.0f;
                    bs[i] = (cev && ce + i < a.Cm) ? a.b_dw[ce + i] : 0.0f;
                }
#pragma unroll
                for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const bool livew = mine && 2 * j + (q >> 1) < 9;
                        const unsigned bits = livew ? (xb_bf16_bits(wv[kt][j]) << sh) : 0u;
                        A[kt][j] = xb_u32x4{dwsel == 0 ? bits : 0u, dwsel == 1 ? bits : 0u, dwsel == 2 ? bits : 0u, dwsel == 3 ? bits : 0u};
                    }
            }
            // frame s has landed for this wave (nothing younger is in flight), then for everyone; nobody still reads the other slot
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            xb_barrier();
            issue(s + 1);
            if (wave_live && !(abl & 1)) {
                int fbo = (s & 1) * fbytes + lbase0 + ph * NTW * lstep;
                asm volatile("" : "+v"(fbo));
                const char* ta[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) ta[j] = ring + fbo + tapoff[j];
                bf16x8 Bq[2][5];
#pragma unroll
                for (int j = 0; j < 5; ++j) Bq[0][j] = *reinterpret_cast<const bf16x8*>(ta[j]);
                __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
                for (int l = 0; l < NTW; ++l) {
                    if (l + 1 < NTW) {
#pragma unroll
                        for (int j = 0; j < 5; ++j) Bq[(l + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(ta[j] + (l + 1) * lstep);
                        __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
                    }
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const bf16x8 B = Bq[l & 1][j];
                        S0[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[2][j]), B, j == 0 ? S1[l] : S0[l], 0, 0, 0);
                        S1[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[1][j]), B, j == 0 ? S2[l] : S1[l], 0, 0, 0);
                        S2[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[0][j]), B, j == 0 ? zero4 : S2[l], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 15, 0);
                }
            }
            // S0 now holds output frame t0 + f - 2 of this quad: complete once f >= 2
            if (wave_live && f >= 2 && t0 + f - 2 < T) {
                const int tfo = f - 2;
                const bool cev = ce < Cmp;
#pragma unroll
                for (int l = 0; l < NTW; ++l) {
                    const int lr = (ph * NTW + l) * RPT + mrow;
                    const bool ok = lane_pos && cev && lr < rows_valid;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = S0[l][i] * sc[i] + bs[i];
                        v[i] = v[i] * sigmoidf_(v[i]);
                    }
                    bf16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
                    const int r = ok ? tfo * BHW + lr * BW + mcol : R;  // (row R: the dump row)
                    *reinterpret_cast<bf16x4*>(dwact + (r * DPL) * 16 + (ok ? ce : 0) * 2) = o;
                }
            }
        }
        __syncthreads();  // dwact and the row table are complete; the ring is dead (xt may overwrite it)

        // ================================ P: project conv + residual + ReLU ==============================================================
        const int npairs = (RTn + 1) >> 1;
        if (!(abl & 4)) {
#pragma unroll 1
            for (int u = wave; u < g.CTC * npairs; u += 8) {
                const int co = u / npairs, pp = u - co * npairs;
                f32x16 acc0, acc1;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.0f;
                const __bf16* ab = a.w_c + ((long)co * g.KSC * 64 + lane) * 8;
                const char* b0 = dwact + ((pp * 64 + c) * DPL + h) * 16;
                const char* b1 = b0 + 32 * DPL * 16;
                bf16x8 A0 = load_frag<__bf16>(ab), A1 = load_frag<__bf16>(ab + 512);
#pragma unroll 1
                for (int ks = 0; ks < g.KSC; ks += 2) {  // KSC is even (host pads K with zero weights)
                    const bf16x8 a0 = A0, a1 = A1;
                    A0 = load_frag<__bf16>(ab + (size_t)min(ks + 2, g.KSC - 2) * 512);
                    A1 = load_frag<__bf16>(ab + (size_t)min(ks + 3, g.KSC - 1) * 512);
                    const bf16x8 B00 = *reinterpret_cast<const bf16x8*>(b0 + ks * 32), B10 = *reinterpret_cast<const bf16x8*>(b1 + ks * 32);
                    const bf16x8 B01 = *reinterpret_cast<const bf16x8*>(b0 + ks * 32 + 32), B11 = *reinterpret_cast<const bf16x8*>(b1 + ks * 32 + 32);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, B00, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, B10, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B01, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B11, acc1, 0, 0, 0);
                }
                float scv[2][8], bsv[2][8];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    load8(a.s_c + co * 32 + 16 * pr + 8 * h, scv[pr]);
                    load8(a.b_c + co * 32 + 16 * pr + 8 * h, bsv[pr]);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int rt = pp * 2 + mt;
                    if (rt < RTn) {  // wave-uniform
                        const int r = rt * 32 + c;
                        const unsigned gp = rowtab[r];
                        uint4 rraw[2];
                        unsigned off[2];
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            const int ch = co * 32 + 16 * pr + 8 * h;
                            off[pr] = (gp != 0xffffffffu && ch < Cop) ? (gp * (unsigned)Cop + (unsigned)ch) * 2u : XB_OOB;
                            rraw[pr] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, (int)off[pr], 0, 0));
                        }
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            float v[8], r8[8];
#pragma unroll
                            for (int qq = 0; qq < 4; ++qq) {
                                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt ? acc1[8 * pr + qq] : acc0[8 * pr + qq]),
                                                                                 __float_as_uint(mt ? acc1[8 * pr + 4 + qq] : acc0[8 * pr + 4 + qq]), false, false);
                                v[qq] = __uint_as_float(sw[0]);
                                v[4 + qq] = __uint_as_float(sw[1]);
                            }
                            const uint4 rr1[1] = {rraw[pr]};
                            raw_to_f8<__bf16>(rr1, r8);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = relu_f32(v[e] * scv[pr][e] + bsv[pr][e] + r8[e]);
                            bf16x8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xb_u32x4, o), yrsrc, (int)off[pr], 0, 0);
                            const int ch = co * 32 + 16 * pr + 8 * h;
                            if (EXPAND && ch < Cop) *reinterpret_cast<bf16x8*>(xt + (r * XPL + (ch >> 3)) * 16) = o;
                        }
                    }
                }
            }
        }
        if (EXPAND) {
            __syncthreads();  // the block-output tile is complete in xt
            // ================================ E: the next block's expand conv + ReLU =====================================================
            if (!(abl & 8)) {
#pragma unroll 1
                for (int u = wave; u < g.CTA * npairs; u += 8) {
                    const int co = u / npairs, pp = u - co * npairs;
                    f32x16 acc0, acc1;
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.0f;
                    const __bf16* ab = a.w_a + ((long)co * g.KSA * 64 + lane) * 8;
                    const char* b0 = xt + ((pp * 64 + c) * XPL + h) * 16;
                    const char* b1 = b0 + 32 * XPL * 16;
                    bf16x8 A0 = load_frag<__bf16>(ab), A1 = load_frag<__bf16>(ab + 512);
#pragma unroll 1
                    for (int ks = 0; ks < g.KSA; ks += 2) {
                        const bf16x8 a0 = A0, a1 = A1;
                        A0 = load_frag<__bf16>(ab + (size_t)min(ks + 2, g.KSA - 2) * 512);
                        A1 = load_frag<__bf16>(ab + (size_t)min(ks + 3, g.KSA - 1) * 512);
                        const bf16x8 B00 = *reinterpret_cast<const bf16x8*>(b0 + ks * 32), B10 = *reinterpret_cast<const bf16x8*>(b1 + ks * 32);
                        const bf16x8 B01 = *reinterpret_cast<const bf16x8*>(b0 + ks * 32 + 32), B11 = *reinterpret_cast<const bf16x8*>(b1 + ks * 32 + 32);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, B00, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, B10, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B01, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B11, acc1, 0, 0, 0);
                    }
                    float scv[2][8], bsv[2][8];
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        load8(a.s_a + co * 32 + 16 * pr + 8 * h, scv[pr]);
                        load8(a.b_a + co * 32 + 16 * pr + 8 * h, bsv[pr]);
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const int rt = pp * 2 + mt;
                        if (rt < RTn) {
                            const unsigned gp = rowtab[rt * 32 + c];
#pragma unroll
                            for (int pr = 0; pr < 2; ++pr) {
                                float v[8];
#pragma unroll
                                for (int qq = 0; qq < 4; ++qq) {
                                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt ? acc1[8 * pr + qq] : acc0[8 * pr + qq]),
                                                                                     __float_as_uint(mt ? acc1[8 * pr + 4 + qq] : acc0[8 * pr + 4 + qq]), false, false);
                                    v[qq] = __uint_as_float(sw[0]);
                                    v[4 + qq] = __uint_as_float(sw[1]);
                                }
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = relu_f32(v[e] * scv[pr][e] + bsv[pr][e]);
                                bf16x8 o;
#pragma unroll
                                for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
                                const int ch = co * 32 + 16 * pr + 8 * h;
                                const unsigned off = (gp != 0xffffffffu && ch < a.Cnp) ? (gp * (unsigned)a.Cnp + (unsigned)ch) * 2u : XB_OOB;
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xb_u32x4, o), ersrc, (int)off, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();  // nobody reads xt / dwact / the row table any more: the next tile's first frame may land
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------------
static bool xb_pointwise(const pasn_conv_desc& d) {
    return d.kt == 1 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && !d.pt && !d.ph && !d.pw;
}

// d_dw: the stencil (inner -> inner); d_c: project (inner -> C, + residual, ReLU); d_a: the next block's expand conv (C -> inner'), or NULL
XbGeom xb_geom(const pasn_conv_desc& dd, const pasn_conv_desc& dc, const pasn_conv_desc* da, int dtype) {
    XbGeom g{};
    if (dtype != PASN_BF16) return g;
    if (const char* e = tune("PASN_NO_BLOCK"))
        if (e[0] == '1') return g;
    const bool stencil = dd.kt == 3 && dd.kh == 3 && dd.kw == 3 && dd.st == 1 && dd.sh == 1 && dd.sw == 1 && dd.pt == 1 && dd.ph == 1 && dd.pw == 1 &&
                         dd.To == dd.Ti && dd.Ho == dd.Hi && dd.Wo == dd.Wi && dd.Cin_p == dd.Cout_p && dd.Cout_p % 8 == 0 && dd.act == PASN_ACT_SWISH;
    if (!stencil || !xb_pointwise(dc) || dc.in_swish || dc.act != PASN_ACT_RELU) return g;
    if (dc.N != dd.N || dc.To != dd.To || dc.Ho != dd.Ho || dc.Wo != dd.Wo || dc.Cin != dd.Cout || dc.Cin_p != dd.Cout_p || dc.w_frag != 1) return g;
    if (dc.Cout_p % 16 != 0 || dc.Cout_p < 48 || dc.Cout_p > 256 || dd.Cout_p > 512) return g;  // (narrower blocks: byte-bound on big planes, the separate launches win)
    g.KSC = (dd.Cout_p + 31) / 32 * 2;
    if (dc.w_kc != g.KSC * 16) return g;  // the host pads K to an EVEN number of 16-wide steps
    g.CTC = (dc.Cout_p + 31) / 32;
    if (dc.w_rows < g.CTC * 32) return g;
    if (da) {
        if (!xb_pointwise(*da) || da->in_swish || da->act != PASN_ACT_RELU || da->w_frag != 1) return g;
        if (da->N != dc.N || da->To != dc.To || da->Ho != dc.Ho || da->Wo != dc.Wo || da->Cin != dc.Cout || da->Cin_p != dc.Cout_p) return g;
        g.KSA = (dc.Cout_p + 31) / 32 * 2;
        g.CTA = (da->Cout_p + 31) / 32;
        if (da->w_kc != g.KSA * 16 || da->w_rows < g.CTA * 32 || da->Cout_p % 8 != 0) return g;
    }
    const long M = (long)dd.N * dd.To * dd.Ho * dd.Wo;
    if (M * dd.Cout_p * 2 >= (1L << 30) || (da && M * da->Cout_p * 2 >= (1L << 30)) || (long)dd.Ti * dd.Hi * dd.Wi * dd.Cin_p * 2 >= (1L << 31)) return g;
    g.BW = std::min(dd.Wo, 14);
    g.RPT = g.BW <= 8 ? 2 : 1;
    g.BH = std::min(dd.Ho, xb_tiles(g.RPT) * g.RPT);
    if (g.RPT == 1) g.BH = std::min(g.BH, 7);
    g.RTH = ceil_div(dd.Ho, g.BH);
    g.RTW = ceil_div(dd.Wo, g.BW);
    g.TF = 2;
    if (const char* e = tune("PASN_BLOCK_TF")) g.TF = std::max(1, std::min(4, atoi(e)));
    while (g.TF > 1 && g.TF * g.BH * g.BW > 224) --g.TF;
    g.TF = std::min(g.TF, dd.To);
    g.nTch = ceil_div(dd.To, g.TF);
    g.R = g.TF * g.BH * g.BW;
    g.RTn = ceil_div(g.R, 32);
    g.NQ = ceil_div(dd.Cout_p, 64);
    g.DPL = (2 * g.KSC) | 1;
    g.XPL = da ? ((2 * g.KSA) | 1) : 1;
    auto kib = [](int b) { return (b + 1023) / 1024 * 1024; };
    const int ring = 2 * xb_ni(g.RPT) * 1024;
    const int xtb = da ? g.RTn * 32 * g.XPL * 16 : 0;
    g.dw_off = kib(std::max(ring, xtb));
    g.tab_off = g.dw_off + kib((g.R + 1) * g.DPL * 16);
    g.lds_bytes = g.tab_off + kib(g.RTn * 32 * 4);
    if (g.lds_bytes > 160 * 1024) return XbGeom{};
    g.tiles = dd.N * g.RTH * g.RTW * g.nTch;
    const int grid = std::min(g.tiles, 256);
    g.tpb = ceil_div(g.tiles, grid);
    g.grid = ceil_div(g.tiles, g.tpb);
    g.abl = tune_dev("PASN_BLOCK_ABL") ? atoi(tune_dev("PASN_BLOCK_ABL")) : 0;
    g.ok = 1;
    return g;
}

int launch_x3d_block(const void* e, const float* w_dw, const float* s_dw, const float* b_dw, const void* w_c, const float* s_c, const float* b_c,
                     const void* res, void* y, const void* w_a, const float* s_a, const float* b_a, void* e_next, const pasn_conv_desc& dd,
                     const pasn_conv_desc& dc, const pasn_conv_desc* da, const XbGeom& g, hipStream_t s) {
    XbArgs a{(const __bf16*)e, w_dw, s_dw, b_dw, (const __bf16*)w_c, s_c, b_c, (const __bf16*)res, (__bf16*)y, (const __bf16*)w_a, s_a, b_a,
             (__bf16*)e_next, dd.N, dd.To, dd.Ho, dd.Wo, dd.Cout, dd.Cout_p, dc.Cout_p, da ? da->Cout_p : 0};
    const dim3 grid(g.grid), block(512);
#define PASN_XB(RPT_, EXP_, ABL_)                                                                                      \
    do {                                                                                                               \
        PASN_MAX_LDS(160 * 1024, x3d_block_kernel<RPT_, EXP_, ABL_>);                                                 \
        hipLaunchKernelGGL((x3d_block_kernel<RPT_, EXP_, ABL_>), grid, block, (size_t)g.lds_bytes, s, a, g);          \
    } while (0)
#ifdef PASN_TUNING
    if (g.abl) {
        if (g.RPT == 2) PASN_XB(2, true, 1);
        else PASN_XB(1, true, 1);
        return check_launch("x3d_block_kernel (ablation)");
    }
#endif
    if (g.RPT == 2) {
        if (da) PASN_XB(2, true, 0);
        else PASN_XB(2, false, 0);
    } else {
        if (da) PASN_XB(1, true, 0);
        else PASN_XB(1, false, 0);
    }
#undef PASN_XB
    return check_launch("x3d_block_kernel");
}

}  // namespace pasn

using namespace pasn;

static bool xb_desc_ok(const pasn_conv_desc* d) { return d && d->N > 0 && d->To > 0 && d->Ho > 0 && d->Wo > 0 && d->Cin > 0 && d->Cout > 0; }

extern "C" int pasn_x3d_block_supported(const pasn_conv_desc* d_dw, const pasn_conv_desc* d_c, const pasn_conv_desc* d_a, int dtype) {
    if (!xb_desc_ok(d_dw) || !xb_desc_ok(d_c) || (d_a && !xb_desc_ok(d_a))) return 0;
    return xb_geom(*d_dw, *d_c, d_a, dtype).ok;
}

extern "C" int pasn_x3d_block_fwd(const void* e, const float* w_dw, const float* scale_dw, const float* bias_dw, const void* w_c,
                                  const float* scale_c, const float* bias_c, const void* residual, void* y, const void* w_a,
                                  const float* scale_a, const float* bias_a, void* e_next, const pasn_conv_desc* d_dw, const pasn_conv_desc* d_c,
                                  const pasn_conv_desc* d_a, int dtype, void* stream) {
    PASN_REQUIRE(e && w_dw && scale_dw && bias_dw && w_c && scale_c && bias_c && residual && y, "null pointer");
    PASN_REQUIRE((d_a != nullptr) == (w_a != nullptr) && (d_a == nullptr || (scale_a && bias_a && e_next)), "the next expand conv comes with all of its operands, or not at all");
    PASN_REQUIRE(xb_desc_ok(d_dw) && xb_desc_ok(d_c) && (!d_a || xb_desc_ok(d_a)), "bad geometry");
    const XbGeom g = xb_geom(*d_dw, *d_c, d_a, dtype);
    PASN_REQUIRE(g.ok, "block not covered (pasn_x3d_block_supported returns 0)");
    return launch_x3d_block(e, w_dw, scale_dw, bias_dw, w_c, scale_c, bias_c, residual, y, w_a, scale_a, bias_a, e_next, *d_dw, *d_c, d_a, g,
                            (hipStream_t)stream);
}

// A whole X3D residual block of the LAST stage (7 x 7 planes, block width 192, inner width 432, no squeeze-excite) in ONE launch -- bf16:
//   conv_a (1x1x1) + norm_a + ReLU -> conv_b (depthwise 3x3x3) + norm_b + Swish -> conv_c (1x1x1) + norm_c + residual + ReLU
//   (-> the NEXT block's conv_a + norm_a + ReLU), with both 2.25x-wide tensors only ever in LDS.
//
// Round 4's first whole-block launch (x3d_block.hip, retired in round 5: stencil -> project -> next expand) read the expanded activation through a (quad, frame) DMA
// pipeline whose steps serialise wait / issue / MFMAs / epilogue: 65 us against 55 for the separate launches.  On a 7 x 7 plane the OTHER cut is
// affordable: a tile = (clip, two output frames) needs the block input of FOUR whole frames -- 196 rows x 192 channels = 78 KB, no spatial halo --
// and from then on everything is on chip: per 64-channel quad of the inner width
//   E  expand conv of the four frames from the x image (MFMA 32x32x16, weight fragments streamed from L2), ReLU, rounded to bf16, written as
//      dwmfma.hip's frame images (10-position rows, zero border kept from one clearing pass; frames outside the clip are written as zeros:
//      the stencil pads the EXPANDED activation);
//   D  stencil: wave = (16-channel tile of the quad, output frame): 3 kt x 5 tap pairs x 4 position tiles of block-diagonal 16x16x32 MFMAs
//      straight from the four images -- no ring, no DMA, no rotation; scale / bias / Swish -> bf16 -> the quad's chunk [98 rows][64 ch];
//   P  project conv, this quad's four k-steps: wave = (row tile, three 32-channel output tiles), accumulators kept across the quads.
// Two barriers per quad, 14 in all; the expand conv is computed on 4 frames for 2 outputs (the T halo: 2x its MFMAs, ~1/3 of the launch's).
// Then the project epilogue (+ residual from the block input, ReLU, 16-byte stores, block-output image) and, optionally, the next block's
// expand conv from that image (xb_pointwise.h).  Rounding points and accumulation orders are those of the separate launches: bit-identical.
#include "common.h"
#include "xb_pointwise.h"

namespace pasn {

struct EdpArgs {
    const __bf16* x;          // block input (and residual) [M][Cxp]
    const __bf16* w_a;        // expand weights, fragment-major [CTA][KSA][64][8]
    const float *s_a, *b_a;   // [>= 32 CTA]
    const unsigned short* wq; // stencil weight operands (plan.stencil_operands)
    const float *s_dw, *b_dw; // [Cmp]
    const __bf16* w_c;        // project weights, fragment-major [CTC][KSC][64][8], K zero-padded to KSC (even) steps
    const float *s_c, *b_c;   // [>= 32 CTC]
    __bf16* y;                // block output [M][Cxp]
    const __bf16* w_n;        // the NEXT block's expand weights [CTN][KSA][64][8] (NULL: none)
    const float *s_n, *b_n;
    __bf16* e_next;           // [M][Cnp]
    int N, T, Cm, Cmp, Cxp, Cnp;
};

#ifdef PASN_TUNING
// 100 MHz stamps of block 0 / thread 0 (tuning builds, PASN_EDP_STAMPS=1; tools/edp_bench.py): [0] start, [1] x image landed, [2] quads done,
// [3] project epilogue done, [4] next expand done, [5..7] time inside the E / D / P phases summed over the quads
__device__ long long edp_stamps[8];
#define EDP_NOW() ((long long)__builtin_amdgcn_s_memrealtime())
#define EDP_ON (g.stamps && blockIdx.x == 0 && threadIdx.x == 0)
#endif

__device__ __forceinline__ void edp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int EDP_HW = 7, EDP_S = 49;           // the plane
constexpr int EDP_RWI = 10, EDP_SLOTS = 10;     // frame image: 9 rows x 10 positions x 10 slots of 16 bytes (8 used: one quad)
// Slots per image ROW: 100 + 2 of padding (round 5).  A stencil operand read serves 16 lanes = TWO output rows of 7 positions; with rows exactly
// 100 slots apart the second row's lanes fall on the first row's banks -- 2-way conflicts on EVERY operand read of a phase that sits at its LDS
// bound (tools/pmc_lds_audit.sh: 6.6 LDS cycles per instruction, 36 % of them conflicts).  102 (= 6 mod 16) makes the 16 addresses cover the
// 64 banks exactly, provided the two lanes without a position read a cell another lane reads (broadcast) instead of a clamped one of their own.
constexpr int EDP_ROWP = EDP_RWI * EDP_SLOTS + 2;
constexpr int EDP_FB = 9 * EDP_ROWP * 16;  // 14688 bytes per frame image
constexpr int EDP_CPL = 9;                       // slots per row of the stencil-output chunk

template <int KSA, int KSC, bool NEXT>
__global__ __launch_bounds__(512) void x3d_edp_kernel(EdpArgs a, EdpGeom g) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int XPL = 2 * KSA + 1;  // slots per row of the x image (and of the block-output image)
    constexpr int TF = 2, NF = TF + 2, R = TF * EDP_S, RI = NF * EDP_S;  // 98 output rows, 196 staged input rows
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = threadIdx.x;
    const int c = lane & 31, h = lane >> 5;   // pointwise roles
    const int m = lane & 15, q4 = lane >> 4;  // stencil roles
    char* const xin = smem;                        // [RI][XPL] slots
    char* const fimg = smem + g.fimg_off;          // [NF][EDP_FB]; later the block-output image [128][XPL]
    char* const dch = smem + g.dch_off;            // [R + 1][EDP_CPL] slots (row R: dump row)
    char* const xt = fimg;
    unsigned* const rowtab = reinterpret_cast<unsigned*>(smem + g.tab_off);  // [128]
    float* const cst = reinterpret_cast<float*>(smem + g.cst_off);           // stencil [2][Cmp], project [2][32 CTC], next expand [2][32 CTN]
    const int Cmp = a.Cmp, Cxp = a.Cxp, T = a.T;
    float* const sdw = cst, *const bdw = cst + Cmp, *const scp = cst + 2 * Cmp, *const bcp = scp + 32 * g.CTC, *const snp = bcp + 32 * g.CTC,
                *const bnp = snp + 32 * g.CTN;

#ifdef PASN_TUNING
    long long te = 0, td = 0, tp = 0, tq = 0;
    if (EDP_ON) edp_stamps[0] = EDP_NOW();
#endif
    // one clearing pass: the zero borders of the frame images (never written afterwards), the chunk's pad slot and dump row
    for (int i = tid; i < (g.tab_off - g.fimg_off) / 16; i += 512) reinterpret_cast<uint4*>(fimg)[i] = uint4{0u, 0u, 0u, 0u};
    for (int i = tid; i < Cmp; i += 512) {
        sdw[i] = a.s_dw[i];
        bdw[i] = a.b_dw[i];
    }
    for (int i = tid; i < 32 * g.CTC; i += 512) {
        scp[i] = a.s_c[i];
        bcp[i] = a.b_c[i];
    }
    if (NEXT)
        for (int i = tid; i < 32 * g.CTN; i += 512) {
            snp[i] = a.s_n[i];
            bnp[i] = a.b_n[i];
        }
    const long M = (long)a.N * T * EDP_S;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x), 0, (unsigned)(M * Cxp * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (unsigned)(M * Cxp * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t ersrc = __builtin_amdgcn_make_buffer_rsrc(NEXT ? a.e_next : a.y, 0, NEXT ? (unsigned)(M * a.Cnp * 2) : 0u, 0x00020000);
    const int PPR = Cxp >> 3;
    const int nTch = (T + TF - 1) / TF;
    const int lbl = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_end = min(g.tiles, (lbl + 1) * g.tpb);
    // stencil roles (dwmfma.hip's two-rows-per-tile layout on 10-position image rows)
    const int ctw = wave & 3, tfo = wave >> 2;
    const int mrow = m / EDP_HW, mcol = m - mrow * EDP_HW;
    const int dwsel = (m & 7) >> 1, wsh = (m & 1) * 16;
    int tapoff[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int tap9 = min(2 * j + (q4 >> 1), 8);
        tapoff[j] = ((tap9 / 3) * EDP_ROWP + (tap9 % 3) * EDP_SLOTS) * 16;
    }
    const bool mpos = m < 2 * EDP_HW;  // lanes 14, 15 hold no position: they read lane 13's cell
    const int lbase0 = ((mpos ? mrow : 1) * EDP_ROWP + (mpos ? mcol : EDP_HW - 1) * EDP_SLOTS + 2 * ctw + (q4 & 1)) * 16;
    constexpr int lstep = 2 * EDP_ROWP * 16;
    const int CT16 = (Cmp + 15) >> 4;
    // project roles: row tile rt, output tiles co = cg, cg + 2, cg + 4
    const int prt = wave & 3, pcg = wave >> 2;
    // expand roles: output tile (of the quad) ea, row tiles er, er + 4
    const int ea = wave & 1, er = wave >> 1;
    __syncthreads();

#pragma unroll 1
    for (int tile = lbl * g.tpb; tile < tile_end; ++tile) {
        const int tch = tile % nTch, n = tile / nTch;
        const int t0 = tch * TF;
        // ---- the block input of frames t0 - 1 .. t0 + 2 by LDS-DMA (frames outside the clip, K padding and the pad slot: out-of-range lanes = zeros) ----
        const int nix = (RI * XPL + 63) >> 6;
        for (int j = wave; j < nix; j += 8) {
            const int s = j * 64 + lane;
            const int r = s / XPL, p = s - r * XPL;
            const int f = r / EDP_S, pos = r - f * EDP_S, ti = t0 - 1 + f;
            const unsigned off = (r < RI && p < PPR && ti >= 0 && ti < T) ? (unsigned)(((n * T + ti) * EDP_S + pos) * Cxp + p * 8) * 2u : XB_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (xb_lds_ptr_t)(xin + j * 1024), 16, (int)off, 0, 0, 0);
        }
        if (tid < 128) {
            const int f = tid / EDP_S, pos = tid - f * EDP_S;
            rowtab[tid] = (tid < R && t0 + f < T) ? (unsigned)((n * T + t0 + f) * EDP_S + pos) : 0xffffffffu;
        }
        f32x16 pacc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[i][r] = 0.0f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#ifdef PASN_TUNING
        if (EDP_ON) edp_stamps[1] = EDP_NOW();
#endif

#pragma unroll 1
        for (int qd = 0; qd < g.NQ; ++qd) {
#ifdef PASN_TUNING
            if (g.stamps) tq = EDP_NOW();
#endif
            // ================================ E: expand conv of the four frames, this quad's 64 channels ================================
            {
                const int co = min(2 * qd + ea, g.CTA - 1);
                const bool colive = 2 * qd + ea < g.CTA;
                bf16x8 A[KSA];
                const __bf16* ab = a.w_a + ((long)co * KSA * 64 + lane) * 8;
#pragma unroll
                for (int ks = 0; ks < KSA; ++ks) A[ks] = load_frag<__bf16>(ab + ks * 512);
                float scv[2][8], bsv[2][8];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    load8(a.s_a + co * 32 + 16 * pr + 8 * h, scv[pr]);
                    load8(a.b_a + co * 32 + 16 * pr + 8 * h, bsv[pr]);
                }
                f32x16 acc[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
                const char* b0 = xin + ((er * 32 + c) * XPL + h) * 16;
#pragma unroll
                for (int ks = 0; ks < KSA; ++ks) {
                    const bf16x8 B0 = *reinterpret_cast<const bf16x8*>(b0 + ks * 32);
                    const bf16x8 B1 = *reinterpret_cast<const bf16x8*>(b0 + 4 * 32 * XPL * 16 + ks * 32);  // (row tile er + 4; for er = 3 rows beyond the image: unused)
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ks], B0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ks], B1, acc[1], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = (er + 4 * i) * 32 + c;
                    const int f = r / EDP_S, pos = r - f * EDP_S;
                    const int ph = pos / EDP_HW, pw = pos - ph * EDP_HW;
                    const int ti = t0 - 1 + f;
                    const bool wr = colive && r < RI;
                    const bool inclip = ti >= 0 && ti < T;
                    char* const cell = wr ? fimg + f * EDP_FB + ((ph + 1) * EDP_ROWP + (pw + 1) * EDP_SLOTS + 4 * ea + h) * 16 : dch + R * EDP_CPL * 16;
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        float v[8];
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][8 * pr + qq]), __float_as_uint(acc[i][8 * pr + 4 + qq]), false, false);
                            v[qq] = __uint_as_float(sw[0]);
                            v[4 + qq] = __uint_as_float(sw[1]);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] * scv[pr][e] + bsv[pr][e];
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (__bf16)relu_f32(v[e]);
                        xb_u32x4 ou = __builtin_bit_cast(xb_u32x4, o);
                        if (!inclip) ou = xb_u32x4{0u, 0u, 0u, 0u};  // the stencil pads the EXPANDED activation with zeros in T
                        *reinterpret_cast<xb_u32x4*>(cell + (wr ? pr * 32 : 0)) = ou;
                    }
                }
            }
            // the stencil's weight operands of this quad and the project conv's fragments: requested before the barrier, used behind it
            const int c0 = (qd * 4 + ctw) * 16;
            const bool dlive = c0 < Cmp;
            const int ct = min(qd * 4 + ctw, CT16 - 1);
            const uint4* wp = reinterpret_cast<const uint4*>(a.wq + ((size_t)(ct * 2) * 64 + lane) * 8);
            const uint4 w0v = wp[0], w1v = wp[64];
            bf16x8 PA[3][4];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    PA[i][kk] = load_frag<__bf16>(a.w_c + ((long)((pcg + 2 * i) * KSC + min(4 * qd + kk, KSC - 1)) * 64 + lane) * 8);
            edp_barrier();  // the four frame images of this quad are complete
#ifdef PASN_TUNING
            if (g.stamps) {
                const long long now = EDP_NOW();
                te += now - tq;
                tq = now;
            }
#endif
            // ================================ D: stencil, wave = (channel tile, output frame) =============================================
            if (dlive && t0 + tfo < T) {
                xb_u32x4 A[3][5];
                const unsigned wd[8] = {w0v.x, w0v.y, w0v.z, w0v.w, w1v.x, w1v.y, w1v.z, w1v.w};
#pragma unroll
                for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const int e = kt * 5 + j;
                        const unsigned bits = ((e & 1) ? (wd[e >> 1] >> 16) : (wd[e >> 1] & 0xffffu)) << wsh;
                        A[kt][j] = xb_u32x4{dwsel == 0 ? bits : 0u, dwsel == 1 ? bits : 0u, dwsel == 2 ? bits : 0u, dwsel == 3 ? bits : 0u};
                    }
                f32x4 S[4];
#pragma unroll
                for (int l = 0; l < 4; ++l) S[l] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                // accumulation order per output = dwmfma.hip's: kt = 0 (frame t - 1), kt = 1 (t), kt = 2 (t + 1), five tap pairs each
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    const char* fb = fimg + (tfo + kt) * EDP_FB + lbase0;
#pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        bf16x8 B[5];
#pragma unroll
                        for (int j = 0; j < 5; ++j) B[j] = *reinterpret_cast<const bf16x8*>(fb + tapoff[j] + l * lstep);
#pragma unroll
                        for (int j = 0; j < 5; ++j) S[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[kt][j]), B[j], S[l], 0, 0, 0);
                    }
                }
                const int ce = c0 + 4 * q4, cel = ctw * 16 + 4 * q4;
                const bool cev = ce < Cmp;
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(sdw + min(ce, Cmp - 4)), b4 = *reinterpret_cast<const f32x4*>(bdw + min(ce, Cmp - 4));
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    const int lr = 2 * l + mrow;
                    const bool ok = m < 2 * EDP_HW && lr < EDP_HW && cev;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = S[l][i] * (cev ? s4[i] : 0.0f) + (cev ? b4[i] : 0.0f);
                        v[i] = v[i] * sigmoidf_(v[i]);
                    }
                    bf16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
                    const int r = ok ? tfo * EDP_S + lr * EDP_HW + mcol : R;
                    *reinterpret_cast<bf16x4*>(dch + (r * EDP_CPL) * 16 + (ok ? cel : 0) * 2) = o;
                }
            }
            edp_barrier();  // the quad's chunk is complete (and everyone is past the images: the next quad's expand may overwrite them)
#ifdef PASN_TUNING
            if (g.stamps) {
                const long long now = EDP_NOW();
                td += now - tq;
                tq = now;
            }
#endif
            // ================================ P: project conv, this quad's k-steps =======================================================
            {
                const char* b0 = dch + ((prt * 32 + c) * EDP_CPL + h) * 16;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (4 * qd + kk < KSC) {  // wave-uniform (the last quad holds fewer k-steps)
                        const bf16x8 B = *reinterpret_cast<const bf16x8*>(b0 + kk * 32);
#pragma unroll
                        for (int i = 0; i < 3; ++i) pacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PA[i][kk], B, pacc[i], 0, 0, 0);
                    }
                }
            }
#ifdef PASN_TUNING
            if (g.stamps) tp += EDP_NOW() - tq;
#endif
        }
#ifdef PASN_TUNING
        if (EDP_ON) {
            edp_stamps[2] = EDP_NOW();
            edp_stamps[5] = te;
            edp_stamps[6] = td;
            edp_stamps[7] = tp;
        }
#endif
        // ---- project epilogue: scale / bias + residual (the block input, from global) + ReLU -> y (+ the block-output image) ----
        if (NEXT) __syncthreads();  // everyone is past the frame images: the block-output image takes their place
        {
            const int r = prt * 32 + c;
            const unsigned gp = rowtab[r];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int co = pcg + 2 * i;
                unsigned off[2];
                uint4 rraw[2];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int ch = co * 32 + 16 * pr + 8 * h;
                    off[pr] = (gp != 0xffffffffu && ch < Cxp) ? (gp * (unsigned)Cxp + (unsigned)ch) * 2u : XB_OOB;
                    rraw[pr] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off[pr], 0, 0));
                }
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    float v[8], r8[8], scv[8], bsv[8];
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(pacc[i][8 * pr + qq]), __float_as_uint(pacc[i][8 * pr + 4 + qq]), false, false);
                        v[qq] = __uint_as_float(sw[0]);
                        v[4 + qq] = __uint_as_float(sw[1]);
                    }
                    const int ch = co * 32 + 16 * pr + 8 * h;
                    load8(scp + ch, scv);
                    load8(bcp + ch, bsv);
                    const uint4 rr1[1] = {rraw[pr]};
                    raw_to_f8<__bf16>(rr1, r8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] * scv[e] + bsv[e];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r8[e];
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (__bf16)relu_f32(v[e]);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xb_u32x4, o), yrsrc, (int)off[pr], 0, 0);
                    if (NEXT && ch < Cxp) *reinterpret_cast<bf16x8*>(xt + (r * XPL + (ch >> 3)) * 16) = o;
                }
            }
        }
#ifdef PASN_TUNING
        if (EDP_ON) edp_stamps[3] = EDP_NOW();
#endif
        if (NEXT) {
            __syncthreads();  // the block output is complete in xt
            xb_pointwise<KSA, 4, false, false>(a.w_n, snp, bnp, xt, XPL, g.CTN, 4, rowtab, ersrc, ersrc, a.Cnp, nullptr, 0, wave, lane);
            __syncthreads();
#ifdef PASN_TUNING
            if (EDP_ON) edp_stamps[4] = EDP_NOW();
#endif
            // the next tile's expand writes frame-image cells only: restore the zero borders the block-output image overwrote
            for (int i = tid; i < (NF * EDP_FB) / 16; i += 512) reinterpret_cast<uint4*>(fimg)[i] = uint4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
    }
}

// ---- host -----------------------------------------------------------------------------------------------------------------------------
static bool edp_pointwise(const pasn_conv_desc& d) {
    return d.kt == 1 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && !d.pt && !d.ph && !d.pw;
}

// da: expand conv (C -> inner, ReLU); dd: stencil (inner, Swish); dc: project conv (inner -> C, residual = the block input, ReLU); dn: the next
// block's expand conv or NULL
EdpGeom edp_geom(const pasn_conv_desc& da, const pasn_conv_desc& dd, const pasn_conv_desc& dc, const pasn_conv_desc* dn, int dtype) {
    EdpGeom g{};
    if (dtype != PASN_BF16) return g;
    if (const char* e = tune("PASN_NO_EDP"))
        if (e[0] == '1') return g;
    const bool stencil = dd.kt == 3 && dd.kh == 3 && dd.kw == 3 && dd.st == 1 && dd.sh == 1 && dd.sw == 1 && dd.pt == 1 && dd.ph == 1 && dd.pw == 1 &&
                         dd.To == dd.Ti && dd.Ho == dd.Hi && dd.Wo == dd.Wi && dd.Cin_p == dd.Cout_p && dd.act == PASN_ACT_SWISH;
    if (!stencil || !edp_pointwise(da) || !edp_pointwise(dc) || da.act != PASN_ACT_RELU || dc.act != PASN_ACT_RELU || da.in_swish || dc.in_swish) return g;
    if (dd.Hi != EDP_HW || dd.Wi != EDP_HW) return g;  // the whole-plane layout of this kernel
    if (da.N != dd.N || da.To != dd.Ti || da.Ho != dd.Hi || da.Wo != dd.Wi || da.Cout != dd.Cin || da.Cout_p != dd.Cin_p) return g;
    if (dc.N != dd.N || dc.To != dd.To || dc.Ho != dd.Ho || dc.Wo != dd.Wo || dc.Cin != dd.Cout || dc.Cin_p != dd.Cout_p) return g;
    if (dc.Cout != da.Cin || dc.Cout_p != da.Cin_p) return g;  // residual = the block input
    if (da.w_frag != 1 || dc.w_frag != 1 || da.Cin_p != 192 || dd.Cout_p != 432) return g;  // the instantiated widths
    g.KSA = da.Cin_p / 16;
    g.KSC = (dd.Cout_p + 31) / 32 * 2;
    if (da.w_kc != g.KSA * 16 || dc.w_kc != g.KSC * 16) return EdpGeom{};
    g.CTA = (da.Cout_p + 31) / 32;
    g.CTC = (dc.Cout_p + 31) / 32;
    if (g.CTC != 6 || da.w_rows < g.CTA * 32 || dc.w_rows < g.CTC * 32) return EdpGeom{};
    if (dn) {
        if (!edp_pointwise(*dn) || dn->act != PASN_ACT_RELU || dn->in_swish || dn->w_frag != 1) return EdpGeom{};
        if (dn->N != dc.N || dn->To != dc.To || dn->Ho != dc.Ho || dn->Wo != dc.Wo || dn->Cin != dc.Cout || dn->Cin_p != dc.Cout_p) return EdpGeom{};
        g.CTN = (dn->Cout_p + 31) / 32;
        if (dn->w_kc != g.KSA * 16 || dn->w_rows < g.CTN * 32 || dn->Cout_p % 8) return EdpGeom{};
    }
    const long M = (long)dd.N * dd.To * EDP_S;
    if (M * dd.Cout_p * 2 >= (1L << 30) || (dn && M * dn->Cout_p * 2 >= (1L << 30))) return EdpGeom{};
    g.NQ = ceil_div(dd.Cout_p, 64);
    auto kib = [](int b) { return (b + 1023) / 1024 * 1024; };
    const int xpl = 2 * g.KSA + 1;
    g.fimg_off = kib(4 * EDP_S * xpl * 16);
    const int fimg_bytes = kib(std::max(4 * EDP_FB, dn ? 128 * xpl * 16 + 64 : 0));
    g.dch_off = g.fimg_off + fimg_bytes;
    g.tab_off = g.dch_off + kib((2 * EDP_S + 1) * EDP_CPL * 16 + 64);
    g.cst_off = g.tab_off + 1024;
    g.lds_bytes = g.cst_off + kib((2 * dd.Cout_p + 64 * g.CTC + 64 * g.CTN) * 4);
    if (g.lds_bytes > 160 * 1024) return EdpGeom{};
    g.tiles = dd.N * ceil_div(dd.To, 2);
    const int grid = std::min(g.tiles, 256);
    g.tpb = ceil_div(g.tiles, grid);
    g.grid = ceil_div(g.tiles, g.tpb);
    g.stamps = tune_dev("PASN_EDP_STAMPS") ? 1 : 0;
    g.ok = 1;
    return g;
}

}  // namespace pasn

using namespace pasn;

#ifdef PASN_TUNING
extern "C" int pasn_debug_edp_stamps(long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pasn::edp_stamps), sizeof(long long) * 8); }
#endif

static bool edp_desc_ok(const pasn_conv_desc* d) { return d && d->N > 0 && d->To > 0 && d->Ho > 0 && d->Wo > 0 && d->Cin > 0 && d->Cout > 0; }

extern "C" int pasn_x3d_edp_supported(const pasn_conv_desc* d_a, const pasn_conv_desc* d_dw, const pasn_conv_desc* d_c, const pasn_conv_desc* d_n, int dtype) {
    if (!edp_desc_ok(d_a) || !edp_desc_ok(d_dw) || !edp_desc_ok(d_c) || (d_n && !edp_desc_ok(d_n))) return 0;
    return edp_geom(*d_a, *d_dw, *d_c, d_n, dtype).ok;
}

extern "C" int pasn_x3d_edp_fwd(const void* x, const void* w_a, const float* scale_a, const float* bias_a, const void* w_dw, const float* scale_dw,
                                const float* bias_dw, const void* w_c, const float* scale_c, const float* bias_c, void* y, const void* w_n,
                                const float* scale_n, const float* bias_n, void* e_next, const pasn_conv_desc* d_a, const pasn_conv_desc* d_dw,
                                const pasn_conv_desc* d_c, const pasn_conv_desc* d_n, int dtype, void* stream) {
    PASN_REQUIRE(x && w_a && scale_a && bias_a && w_dw && scale_dw && bias_dw && w_c && scale_c && bias_c && y, "null pointer");
    PASN_REQUIRE((d_n != nullptr) == (w_n != nullptr) && (d_n == nullptr || (scale_n && bias_n && e_next)), "the next expand conv comes with all of its operands, or not at all");
    PASN_REQUIRE(edp_desc_ok(d_a) && edp_desc_ok(d_dw) && edp_desc_ok(d_c) && (!d_n || edp_desc_ok(d_n)), "bad geometry");
    const EdpGeom g = edp_geom(*d_a, *d_dw, *d_c, d_n, dtype);
    PASN_REQUIRE(g.ok, "block not covered (pasn_x3d_edp_supported returns 0)");
    EdpArgs a{(const __bf16*)x, (const __bf16*)w_a, scale_a, bias_a, (const unsigned short*)w_dw, scale_dw, bias_dw, (const __bf16*)w_c, scale_c, bias_c,
              (__bf16*)y, (const __bf16*)w_n, scale_n, bias_n, (__bf16*)e_next, d_dw->N, d_dw->To, d_dw->Cout, d_dw->Cout_p, d_a->Cin_p,
              d_n ? d_n->Cout_p : 0};
    const dim3 grid(g.grid), block(512);
    if (d_n) {
        PASN_MAX_LDS(160 * 1024, x3d_edp_kernel<12, 28, true>);
        hipLaunchKernelGGL((x3d_edp_kernel<12, 28, true>), grid, block, (size_t)g.lds_bytes, (hipStream_t)stream, a, g);
    } else {
        PASN_MAX_LDS(160 * 1024, x3d_edp_kernel<12, 28, false>);
        hipLaunchKernelGGL((x3d_edp_kernel<12, 28, false>), grid, block, (size_t)g.lds_bytes, (hipStream_t)stream, a, g);
    }
    return check_launch("x3d_edp_kernel");
}

// Epilogue shared by the implicit-GEMM kernels (igemm.hip, igemm_halo.hip): accumulators (position on the lane) -> scale / bias -> a
// wave-private LDS image of 32 positions x BN channels -> residual + activation + whole-row 16-byte stores.
//
// Measured on the 64 -> 144 (1,3,3) layer (8 x 32 x 56 x 56): the first version of this epilogue was 108 of the launch's 278 us.  It read
// scale / bias with one dependent global load per 4 channels (80 L2 round trips per wave, one after the other) and ran the row loop as
// load -> add -> store chains with a division and 64-bit address arithmetic per 16 bytes.  Now scale / bias sit in LDS (staged once per
// block, before the main loop), the row loop issues the image reads and residual loads of NT iterations before it touches any of them,
// walks (row, channel group) incrementally, and is a pure copy when there is no residual.
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

template <int ACT, int N>
__device__ __forceinline__ void igemm_act(float (&v)[N], int act) {
    if constexpr (ACT == PASN_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = relu_f32(v[e]);
    } else if constexpr (ACT != PASN_ACT_NONE) {
        act_vec(v, act);
    }
}

// Tile j of this wave: rows mbase .. mbase + nvalid - 1 of y (nvalid >= 1), channels n0 .. n0 + 8 * cgs - 1.
// Without a residual the activation is applied on the fp32 accumulators and the row loop is a plain LDS -> global copy; with one the image
// holds the pre-activation sums and the row loop adds, activates and rounds again (the order of the reference: norm, + identity, ReLU).
template <int NT, int MT, bool HAS_RES, int ACT>  // ACT: PASN_ACT_NONE / PASN_ACT_RELU compiled in, -1 = the descriptor's, at run time
__device__ __forceinline__ void igemm_epilogue_tile(const f32x16 (&acc)[NT][MT], int j, __bf16* img, const float* scb,
                                                    const __bf16* __restrict__ res, __bf16* __restrict__ y, long mbase, int nvalid, int n0,
                                                    int cgs, const pasn_conv_desc& d, int lane) {
    constexpr int BN = NT * 32, OROW = BN + 8;
    const int c = lane & 31, h = lane >> 5;
    const int Cout_p = d.Cout_p;
    constexpr bool early = !HAS_RES;  // a template parameter, not a branch: with both paths in one body hipcc put an s_waitcnt vmcnt(0) (for the
                                      // residual loads) after every STORE of the copy path as well -- 20 exposed store latencies per wave
    const bool ragged = n0 + BN > d.Cout;         // wave-uniform: this block holds the padded channels (they must be stored as zeros)
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
            if constexpr (early) {
                igemm_act<ACT>(o, d.act);
                if (ragged) mask_tail(o, d.Cout - (n0 + col));
            }
            store4(img + (size_t)c * OROW + col, o);
        }
    // wave-private image: no block barrier, only this wave's LDS writes must have landed (the compiler orders LDS ops of a wave)
    const int total = 32 * cgs;
    const int drow = 64 / cgs, dcg = 64 - drow * cgs;  // p -> p + 64 in (row, channel group) steps
    int row = lane / cgs, cg = lane - row * cgs;
    __bf16* const ytile = y + mbase * Cout_p + n0;
    const __bf16* const rtile = HAS_RES ? res + mbase * Cout_p + n0 : nullptr;
#pragma unroll
    for (int b0 = 0; b0 < 2; ++b0) {  // 2 * NT row-loop iterations cover 32 rows x BN / 8 channel groups: two batches of NT
        bf16x8 vi[NT], vr[NT];
        int off[NT], cgv[NT];
        bool ok[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int p = lane + 64 * (b0 * NT + u);
            ok[u] = p < total && row < nvalid;
            const int rr = ok[u] ? row : 0, cc = ok[u] ? cg : 0;
            off[u] = rr * Cout_p + cc * 8;
            cgv[u] = cc;
            vi[u] = *reinterpret_cast<const bf16x8*>(img + rr * OROW + cc * 8);
            if constexpr (HAS_RES) vr[u] = *reinterpret_cast<const bf16x8*>(rtile + off[u]);
            row += drow;
            cg += dcg;
            if (cg >= cgs) {
                cg -= cgs;
                ++row;
            }
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            if constexpr (HAS_RES) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (float)vi[u][e] + (float)vr[u][e];
                igemm_act<ACT>(v, d.act);
                if (ragged) mask_tail(v, d.Cout - (n0 + cgv[u] * 8));
                if (ok[u]) store8(ytile + off[u], v);
            } else if (ok[u]) {
                *reinterpret_cast<bf16x8*>(ytile + off[u]) = vi[u];
            }
        }
    }
}

// All MT tiles of this wave.  `rows(j, mbase, nvalid)` names the output rows of tile j.  The residual / activation variants are separate
// bodies behind ONE block-uniform dispatch: with the activation switch inside the unrolled loops the epilogue was 10 k instructions (40
// branch ladders per tile, every activation's code 40 times) -- larger than the instruction cache, and the copy path carried the waits of the
// residual path.  ReLU and identity are what the ResNet trunks use; everything else takes the run-time body.
template <int NT, int MT, bool HAS_RES, int ACT, typename Rows>
__device__ __forceinline__ void igemm_epilogue_body(const f32x16 (&acc)[NT][MT], __bf16* img, const float* scb, const __bf16* __restrict__ res,
                                                    __bf16* __restrict__ y, int n0, int cgs, const pasn_conv_desc& d, int lane, Rows rows) {
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        long mbase;
        int nvalid;
        rows(j, mbase, nvalid);
        if (nvalid > 0) igemm_epilogue_tile<NT, MT, HAS_RES, ACT>(acc, j, img, scb, res, y, mbase, nvalid, n0, cgs, d, lane);
    }
}

// ---- epilogue WITHOUT the LDS image ------------------------------------------------------------------------------------------------
// The 32x32 accumulator has the position on the lane and 4 consecutive channels per register quad; one v_permlane32_swap per register
// pair gives every lane 8 consecutive channels of its position (the pwconv_ws.hip epilogue): scale / bias (LDS, read once per channel
// piece and reused for the MT position tiles), residual (16-byte loads, all of a channel tile's requested before any is used),
// activation, one 16-byte store per piece.  Per tile a buffer descriptor over exactly its valid rows: rows beyond them fall out of
// range (loads return zero, stores are dropped), channel pieces beyond the block's width carry an out-of-range offset.
typedef __attribute__((ext_vector_type(4))) unsigned ige_u32x4;
typedef __attribute__((ext_vector_type(2))) float ige_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 ige_bf16x2;
typedef __attribute__((ext_vector_type(2))) short ige_s16x2;

template <int NT, int MT, bool HAS_RES, int ACT, typename Rows>
__device__ __forceinline__ void igemm_epilogue_direct_body(const f32x16 (&acc)[NT][MT], const float* scb, const __bf16* __restrict__ res,
                                                           __bf16* __restrict__ y, int n0, int cgs, const pasn_conv_desc& d, int lane, Rows rows) {
    constexpr int BN = NT * 32;
    const int c = lane & 31, h = lane >> 5;
    const int Cout_p = d.Cout_p, width = cgs * 8;
    const bool ragged = n0 + BN > d.Cout;  // wave-uniform: this block holds the padded channels (stored as zeros)
    __amdgpu_buffer_rsrc_t yr[MT], rr[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        long mbase;
        int nvalid;
        rows(j, mbase, nvalid);
        const unsigned bytes = nvalid > 0 ? (unsigned)(((nvalid - 1) * Cout_p + width) * 2) : 0u;
        yr[j] = __builtin_amdgcn_make_buffer_rsrc(y + mbase * Cout_p + n0, 0, bytes, 0x00020000);
        rr[j] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(HAS_RES ? res + mbase * Cout_p + n0 : y), 0, HAS_RES ? bytes : 0u, 0x00020000);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        unsigned off[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int col = i * 32 + 16 * pr + 8 * h;
            off[pr] = col < width ? (unsigned)((c * Cout_p + col) * 2) : 0x80000000u;
        }
        if constexpr (!HAS_RES && (ACT == PASN_ACT_RELU || ACT == PASN_ACT_NONE)) {
            // No residual, ReLU or nothing: scale / bias on the accumulator rows as they are, then bf16 rounding and ReLU on PACKED pairs
            // (v_cvt_pk_bf16_f32, v_pk_max_i16: a bf16 is negative iff it is negative as an int16, and rounding keeps the sign, so the
            // result equals round(max(v, 0))), and only then the half-wave exchange, on half as many registers: 36 vector instructions
            // per 32 x 32 tile instead of 48 (the epilogue is ~1/3 of the R(2+1)D layers' time).
            float scr[16], bsr[16];  // rows acc_row(r, h) = 8 (r >> 2) + 4 h + (r & 3): four consecutive channels per register quad
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(scb + i * 32 + 8 * k + 4 * h);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(scb + BN + i * 32 + 8 * k + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    scr[4 * k + e] = s4[e];
                    bsr[4 * k + e] = b4[e];
                }
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                unsigned P[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const ige_f32x2 a = {acc[i][j][2 * k] * scr[2 * k] + bsr[2 * k], acc[i][j][2 * k + 1] * scr[2 * k + 1] + bsr[2 * k + 1]};
                    ige_s16x2 m = __builtin_bit_cast(ige_s16x2, __builtin_convertvector(a, ige_bf16x2));
                    if constexpr (ACT == PASN_ACT_RELU) m = __builtin_elementwise_max(m, ige_s16x2{0, 0});
                    P[k] = __builtin_bit_cast(unsigned, m);
                }
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    // registers 8 pr + 0..3 | 8 pr + 4..7 = packed P[4 pr + 0, 1 | 2, 3]: after the exchange lanes < 32 hold channels
                    // 16 pr .. + 7 and lanes >= 32 channels 16 pr + 8 .. + 15 of their position
                    const auto s0 = __builtin_amdgcn_permlane32_swap(P[4 * pr + 0], P[4 * pr + 2], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(P[4 * pr + 1], P[4 * pr + 3], false, false);
                    ige_u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    if (ragged) {  // wave-uniform: zero the channels at or beyond the real count (element e = channel col + e)
                        const int nvalid = d.Cout - (n0 + i * 32 + 16 * pr + 8 * h);
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] &= nvalid >= 2 * k + 2 ? 0xFFFFFFFFu : nvalid == 2 * k + 1 ? 0x0000FFFFu : 0u;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(o, yr[j], (int)off[pr], 0, 0);
                }
            }
            continue;
        }
        bf16x8 rv[2][MT];
        if constexpr (HAS_RES) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int j = 0; j < MT; ++j) rv[pr][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rr[j], (int)off[pr], 0, 0));
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int col = i * 32 + 16 * pr + 8 * h;
            float sc[8], bs[8];
            load8(scb + col, sc);
            load8(scb + BN + col, bs);
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][8 * pr + q]), __float_as_uint(acc[i][j][8 * pr + 4 + q]), false, false);
                    v[q] = __uint_as_float(sw[0]);
                    v[4 + q] = __uint_as_float(sw[1]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + bs[e];
                if constexpr (HAS_RES) {
                    // the image path rounds norm(conv) to bf16 before adding the identity: keep that rounding point
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (float)(__bf16)v[e] + (float)rv[pr][j][e];
                }
                igemm_act<ACT>(v, d.act);
                if (ragged) mask_tail(v, d.Cout - (n0 + col));
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ige_u32x4, o), yr[j], (int)off[pr], 0, 0);
            }
        }
    }
}

template <int NT, int MT, typename Rows>
__device__ __forceinline__ void igemm_epilogue_direct(const f32x16 (&acc)[NT][MT], const float* scb, const __bf16* __restrict__ res,
                                                      __bf16* __restrict__ y, int n0, int cgs, const pasn_conv_desc& d, int lane, Rows rows) {
    if (res) {
        if (d.act == PASN_ACT_RELU) igemm_epilogue_direct_body<NT, MT, true, PASN_ACT_RELU>(acc, scb, res, y, n0, cgs, d, lane, rows);
        else if (d.act == PASN_ACT_NONE) igemm_epilogue_direct_body<NT, MT, true, PASN_ACT_NONE>(acc, scb, res, y, n0, cgs, d, lane, rows);
        else igemm_epilogue_direct_body<NT, MT, true, -1>(acc, scb, res, y, n0, cgs, d, lane, rows);
    } else {
        if (d.act == PASN_ACT_RELU) igemm_epilogue_direct_body<NT, MT, false, PASN_ACT_RELU>(acc, scb, res, y, n0, cgs, d, lane, rows);
        else if (d.act == PASN_ACT_NONE) igemm_epilogue_direct_body<NT, MT, false, PASN_ACT_NONE>(acc, scb, res, y, n0, cgs, d, lane, rows);
        else igemm_epilogue_direct_body<NT, MT, false, -1>(acc, scb, res, y, n0, cgs, d, lane, rows);
    }
}

template <int NT, int MT, typename Rows>
__device__ __forceinline__ void igemm_epilogue(const f32x16 (&acc)[NT][MT], __bf16* img, const float* scb, const __bf16* __restrict__ res,
                                               __bf16* __restrict__ y, int n0, int cgs, const pasn_conv_desc& d, int lane, Rows rows) {
    if (res) {
        if (d.act == PASN_ACT_RELU) igemm_epilogue_body<NT, MT, true, PASN_ACT_RELU>(acc, img, scb, res, y, n0, cgs, d, lane, rows);
        else if (d.act == PASN_ACT_NONE) igemm_epilogue_body<NT, MT, true, PASN_ACT_NONE>(acc, img, scb, res, y, n0, cgs, d, lane, rows);
        else igemm_epilogue_body<NT, MT, true, -1>(acc, img, scb, res, y, n0, cgs, d, lane, rows);
    } else {
        if (d.act == PASN_ACT_RELU) igemm_epilogue_body<NT, MT, false, PASN_ACT_RELU>(acc, img, scb, res, y, n0, cgs, d, lane, rows);
        else if (d.act == PASN_ACT_NONE) igemm_epilogue_body<NT, MT, false, PASN_ACT_NONE>(acc, img, scb, res, y, n0, cgs, d, lane, rows);
        else igemm_epilogue_body<NT, MT, false, -1>(acc, img, scb, res, y, n0, cgs, d, lane, rows);
    }
}

}  // namespace pasn

// Shared by x3d_pe.hip and x3d_edp.hip: one pointwise conv (1x1x1 + folded norm (+ residual) + ReLU) of a row tile whose operand image lies in
// LDS, weight fragments streamed from global memory (fragment-major, L2-resident), lane-swap epilogue, 16-byte stores.
#pragma once
#include "common.h"

namespace pasn {

typedef __attribute__((ext_vector_type(4))) unsigned xb_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned xb_u32x2;
typedef __attribute__((address_space(3))) void* xb_lds_ptr_t;

constexpr unsigned XB_OOB = 0x80000000u;

// One pointwise conv of the tile: out[r][ch] = act(scale * sum_k W[ch][k] img[r][k] + bias (+ residual)), unit = (32 channels, MT 32-row tiles).
// MT: row tiles that share a unit's weight fragments -- 2: twice as many units (all eight waves busy on narrow outputs), every fragment read by
// ceil(RTn / 2) units; 4: half the weight stream (what bounds the 432-channel layers: each CU ingests the layer's whole weight set per tile).
// scale / bias: the launch's tables in LDS (read where they are used: as registers held across the weight burst they were 32 of the 256).
// The whole-K weight fragments of a unit are requested in one burst (one L2 round trip, then streaming); the NEXT unit's burst goes out right
// after the MFMAs of this one, under its epilogue.  RES: + residual rows from global memory; TOLDS: the bf16 outputs also go to `xt`.
template <int KS, int MT, bool RES, bool TOLDS>
__device__ __forceinline__ void xb_pointwise(const __bf16* __restrict__ w, const float* __restrict__ scale, const float* __restrict__ bias,
                                             const char* img, int IPL, int ctiles, int RTn, const unsigned* rowtab,
                                             const __amdgpu_buffer_rsrc_t& rrsrc, const __amdgpu_buffer_rsrc_t& orsrc, int Cout_p, char* xt, int XPL,
                                             int wave, int lane) {
    const int c = lane & 31, h = lane >> 5;
    const int ngroups = (RTn + MT - 1) / MT;
    const int units = ctiles * ngroups;
    // Software pipeline with ONE load site for the weight burst (two sites -- a prologue and the loop tail -- made the compiler keep two copies
    // of the 4 KS fragment registers around the back edge): iteration = [request the PREVIOUS unit's epilogue operands] [request this unit's
    // weights] [previous unit's epilogue: waits for its own, older loads only] [this unit's MFMAs].
    f32x16 acc[MT];
    int pco = 0, ppp = 0;
    bool have_prev = false;
#pragma unroll 1
    for (int u = wave;; u += 8) {
        const bool cur = u < units;  // wave-uniform
        unsigned off[MT][2];
        uint4 rraw[MT][2];
        if (have_prev) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned gp = rowtab[min(ppp * MT + mt, RTn - 1) * 32 + c];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int ch = pco * 32 + 16 * pr + 8 * h;
                    off[mt][pr] = (ppp * MT + mt < RTn && gp != 0xffffffffu && ch < Cout_p) ? (gp * (unsigned)Cout_p + (unsigned)ch) * 2u : XB_OOB;
                    if (RES) rraw[mt][pr] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, (int)off[mt][pr], 0, 0));
                }
            }
        }
        bf16x8 A[KS];
        constexpr int KH = KS > 16 ? KS / 2 : KS;  // K > 256: the burst goes out in two halves around the epilogue (all of it at once + the epilogue's operands
                                                   // did not fit 256 registers: 50 spilled); the second half lands under the MFMAs of the first
        const int co = cur ? u / ngroups : 0, pp = cur ? u - co * ngroups : 0;
        const __bf16* ab = w + ((long)co * KS * 64 + lane) * 8;
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) A[ks] = load_frag<__bf16>(ab + ks * 512);  // (a finished wave re-reads tile 0: harmless, and no branch around the burst)
        if (have_prev) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    float v[8];
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mt][8 * pr + qq]), __float_as_uint(acc[mt][8 * pr + 4 + qq]), false, false);
                        v[qq] = __uint_as_float(sw[0]);
                        v[4 + qq] = __uint_as_float(sw[1]);
                    }
                    float scv[8], bsv[8];
                    load8(scale + pco * 32 + 16 * pr + 8 * h, scv);
                    load8(bias + pco * 32 + 16 * pr + 8 * h, bsv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] * scv[e] + bsv[e];
                    if (RES) {
                        float r8[8];
                        const uint4 rr1[1] = {rraw[mt][pr]};
                        raw_to_f8<__bf16>(rr1, r8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += r8[e];
                    }
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (__bf16)relu_f32(v[e]);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xb_u32x4, o), orsrc, (int)off[mt][pr], 0, 0);
                    if (TOLDS) {
                        const int ch = pco * 32 + 16 * pr + 8 * h;
                        // (channels beyond the width would spill into the next row; a row tile beyond the tile's rows has no image rows)
                        if (ch < Cout_p && ppp * MT + mt < RTn) *reinterpret_cast<bf16x8*>(xt + (((ppp * MT + mt) * 32 + c) * XPL + (ch >> 3)) * 16) = o;
                    }
                }
            }
        }
        if (!cur) break;
#pragma unroll
        for (int ks = KH; ks < KS; ++ks) A[ks] = load_frag<__bf16>(ab + ks * 512);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] = 0.0f;
        const char* b0 = img + ((pp * MT * 32 + c) * IPL + h) * 16;
        // explicit two-deep operand pipeline (left to itself the scheduler hoists all MT KS reads to the top: 4 MT KS registers, spilled)
        bf16x8 Bq[2][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) Bq[0][mt] = *reinterpret_cast<const bf16x8*>(b0 + mt * 32 * IPL * 16);
        __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) Bq[(ks + 1) & 1][mt] = *reinterpret_cast<const bf16x8*>(b0 + mt * 32 * IPL * 16 + (ks + 1) * 32);
                __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ks], Bq[ks & 1][mt], acc[mt], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);
        }
        pco = co;
        ppp = pp;
        have_prev = true;
    }
}

}  // namespace pasn

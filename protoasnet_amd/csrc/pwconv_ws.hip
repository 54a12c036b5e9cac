// Weight-stationary pointwise conv (bf16): a block keeps the weights of its output channels in REGISTERS for the whole launch
// and walks a run of position tiles whose rows arrive by LDS-DMA into a ring of stages -- X3D stages 3-5 and the prototype head.
//
// These layers are skinny GEMMs (M = 25k..400k positions, K and N 48..432 channels) on tensors that largely live in L2 / the
// Infinity Cache.  pwconv_xtile.hip gives every 64-position tile its own block: the block fetches its waves' whole-K weight
// fragments (4 x 28.7 KB for a 55 KB X tile at 432 -> 192: 70-85 MB of L2 weight reads for a 41 MB layer), stages X through
// registers, and runs one memory round trip, one MFMA phase and one epilogue, serialised, per block -- 20-36 us for the 31-41 MB
// stage-5 layers (0.14-0.2 of the HBM rate) in 1.5-3 "rounds" of blocks.  Here:
//   * block = CT x PT waves (<= 8): wave (ct, pt) owns channel tile ct (32 output channels, ALL K: KS fragments = 4 KS VGPRs,
//     loaded ONCE) and MT 32-position sub-tiles of every block tile (BM = 32 PT MT positions); one block per CU (or two), each
//     walking `tpb` CONSECUTIVE tiles, so the weights are read once per CU and a block mostly stays inside one clip;
//   * a stage = the X tile [BM][2 KS + 1 slots] (+ the two gate rows the tile can touch, + the residual tile), filled by
//     `buffer_load ... lds` (no staging registers, no ds_write, out-of-range pieces zero-filled by the hardware: the K padding,
//     the pad slot that makes the row stride odd -- conflict-free ds_read_b128 -- and the rows beyond M cost nothing); NS = 2-3
//     stages, the DMA group of tile i + NS - 1 is issued right after the barrier of tile i, under its MFMAs;
//   * ONE fence-free barrier per tile (two with the input transform); what must have landed is waited for by COUNT
//     (`s_waitcnt vmcnt(n)`: every DMA and store is issued unconditionally -- masked lanes carry an out-of-range offset -- so the
//     wave knows n); the loop holds no register-destination load at all, so hipcc inserts no vmcnt(0) of its own;
//   * the fused input transform x' = swish(x * gate[clip]) is applied IN PLACE on the landed tile, each element once, by all waves;
//   * epilogue without an LDS bounce: the 32x32 accumulator has the position on the lane and 4 consecutive channels per quad;
//     one v_permlane32_swap per register pair gives every lane 8 consecutive channels of its position (guide T21), then
//     scale / bias / residual (16-byte ds_read of the staged residual tile) / activation / one 16-byte buffer store per piece
//     (rows beyond M and channels beyond Cout_p dropped by the range check).
#include "common.h"

namespace pasn {

typedef __attribute__((address_space(3))) void* ws_lds_ptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned ws_u32x4;

constexpr unsigned WS_OOB = 0x80000000u;  // per-lane offset tag: beyond every num_records (all of them < 2^30)

__device__ __forceinline__ void ws_wait_all_but(int n) {  // n wave-uniform: all but this wave's n most recent vector-memory ops are done
    switch (n) {
#define WS_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        WS_W(0) WS_W(1) WS_W(2) WS_W(3) WS_W(4) WS_W(5) WS_W(6) WS_W(7) WS_W(8) WS_W(9) WS_W(10) WS_W(11) WS_W(12) WS_W(13) WS_W(14) WS_W(15)
        WS_W(16) WS_W(17) WS_W(18) WS_W(19) WS_W(20) WS_W(21) WS_W(22) WS_W(23) WS_W(24) WS_W(25) WS_W(26) WS_W(27) WS_W(28) WS_W(29) WS_W(30)
        WS_W(31) WS_W(32) WS_W(33) WS_W(34) WS_W(35) WS_W(36) WS_W(37) WS_W(38) WS_W(39) WS_W(40) WS_W(41) WS_W(42) WS_W(43) WS_W(44) WS_W(45)
        WS_W(46) WS_W(47) WS_W(48) WS_W(49) WS_W(50) WS_W(51) WS_W(52) WS_W(53) WS_W(54) WS_W(55) WS_W(56) WS_W(57) WS_W(58) WS_W(59) WS_W(60)
#undef WS_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // stricter than needed, never weaker (launch_pw_ws keeps n <= 60)
    }
}
// barrier WITHOUT the fence of __syncthreads() (that fence is `s_waitcnt vmcnt(0)`: it would drain the DMA groups in flight)
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// KS2 > 0: CHAINED PAIR (an X3D block's project conv and the next block's expand conv): after the first conv's epilogue the finished
// block-output tile goes to HBM AND, as bf16, into an LDS tile; behind one more barrier every wave w < ctiles2 computes channel tile w of the
// second conv for BOTH 32-position halves of the 64-position tile from it (second weight set in registers too).  Pair mode: PT = 2, MT = 1.
template <int KS, int MT, bool XF, bool RES, int KS2 = 0>
__global__ __launch_bounds__(512) void pwconv_ws_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ bias,
                                                        const __bf16* __restrict__ res, const float* __restrict__ gate,
                                                        __bf16* __restrict__ y, int M, int S, int N, int Cin_p, int Cout, int Cout_p,
                                                        int nks, int act, int in_swish, WsGeom g, WsSe se, WsPair pr2) {
    static_assert(KS2 == 0 || MT == 1, "pair mode: one 32-position sub-tile per wave in the first conv");
    constexpr int MT2 = 2;             // pair mode: 32-position halves per wave in the second conv (PT = 2)
    constexpr int Y1PL = 2 * KS2 + 1;  // 16-byte slots per row of the handed-over tile (odd)
    extern __shared__ __attribute__((aligned(1024))) char smem[];  // [NS] stages: X tile | gate rows | residual tile, each a whole number of KiB
    constexpr int PPRL = 2 * KS + 1;  // 16-byte slots per staged X row: exactly KS k-steps (zero beyond Cin_p) + one pad slot (odd stride)
    constexpr int GPR = KS * 4;       // 16-byte slots per staged gate row (KS * 16 floats)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int CT = g.CT, PT = g.PT, NW = g.NW, NS = g.NS, LA = NS - 1;  // NW >= CT * PT: waves beyond the (channel, position) grid only stage and transform
    const int ctl = wave % CT, pt = wave / CT;
    const int c = lane & 31, h = lane >> 5;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int by = lb % g.gy, slot = lb / g.gy;
    const int ctiles = (Cout_p + 31) >> 5;
    const int ct = by * CT + ctl;
    const bool live = ct < ctiles && pt < PT;  // wave-uniform: a wave without a channel tile still stages, transforms and syncs
    const int BM = 32 * PT * MT;
    // this block's rows: an equal share of M for every block (NOT a whole number of tiles: the launch is bound by the bytes a CU takes
    // in, and rows beyond the share's end are out of range of the block's descriptors -- fetched as zeros at no cost, never stored)
    const unsigned row0 = (unsigned)slot * (unsigned)g.rpb, row1 = min((unsigned)M, row0 + (unsigned)g.rpb);
    const int nt = (int)((row1 - row0 + (unsigned)BM - 1) / (unsigned)BM);
    const int PPR = Cin_p >> 3;
    const int RPL = (Cout_p >> 3) | 1;  // 16-byte slots per staged residual row (odd)
    const unsigned xrow = (unsigned)Cin_p * 2u, yrow = (unsigned)Cout_p * 2u;
#ifdef PASN_WS_ABLATE
    const int abl = g.abl;  // timing ablations (PASN_WS_ABL; results are wrong when set) -- a build with -DPASN_WS_ABLATE only (tools/ws_abl.sh)
#else
    constexpr int abl = 0;  // (as a run-time value its tests sat in front of every store and DMA group of the product kernel)
#endif

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x), 0, row1 * xrow, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(y, 0, row1 * yrow, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(RES ? res : x), 0, RES ? row1 * yrow : 0u, 0x00020000);
    const bool has_gate = XF && gate != nullptr;
    // squeeze-excite gate computed HERE (se.pool != NULL) from the stencil's pool partial rows: the stand-alone gate launch between the
    // stencil and this conv, and its two kernel boundaries, are gone.  A block's rows touch at most two clips (host: rpb <= S)
    const bool se_on = XF && se.pool != nullptr;
    const unsigned n_first = row0 / (unsigned)S;
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_gate ? gate : scale), 0, has_gate ? (unsigned)N * (unsigned)Cin_p * 4u : 0u, 0x00020000);

    // ---- stationary weights: fragment-major (tile ct, step ks) = 64 lanes x 16 bytes; steps beyond w_kc are zeroed below ---------
    bf16x8 A[KS];
    {
        const int ctc = live ? ct : ctiles - 1;
        const __bf16* ab = w + ((long)ctc * nks * 64 + lane) * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) A[ks] = load_frag<__bf16>(ab + (size_t)(ks < nks ? ks : nks - 1) * 512);
    }
    // epilogue constants of this lane's two 8-channel pieces per 32x32 tile (AFTER the lane swap: piece pr = channels 32 ct + 16 pr + 8 h ..)
    float sc[2][8], bs[2][8];
    unsigned yoff[2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const int ch = ct * 32 + 16 * pr + 8 * h;
        const bool ok = live && ch < Cout_p;
        const int chc = ok ? ch : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[pr][j] = 1.0f;
            bs[pr][j] = 0.0f;
        }
        if (scale) load8(scale + chc, sc[pr]);
        if (bias) load8(bias + chc, bs[pr]);
        yoff[pr] = ok ? (unsigned)(pt * MT * 32 + c) * yrow + (unsigned)ch * 2u : WS_OOB;
    }
    const bool tail = live && ct * 32 + 32 > Cout;  // wave-uniform: this tile holds channels beyond the real count (stored as zeros)
    // ---- pair mode: the second conv's stationary weights and epilogue constants (channel tile = wave) ----------------------------------
    const int ctiles2 = KS2 ? (pr2.Cout2_p + 31) >> 5 : 0;
    const bool live2 = KS2 && wave < ctiles2;
    const unsigned y2row = KS2 ? (unsigned)pr2.Cout2_p * 2u : 0u;
    const __amdgpu_buffer_rsrc_t y2rsrc = __builtin_amdgcn_make_buffer_rsrc(KS2 ? pr2.y2 : y, 0, KS2 ? row1 * y2row : 0u, 0x00020000);
    bf16x8 A2[KS2 ? KS2 : 1];
    float sc2[KS2 ? 2 : 1][8], bs2[KS2 ? 2 : 1][8];
    unsigned y2off[2] = {WS_OOB, WS_OOB};
    char* const y1t = smem + NS * g.stage_bytes;  // [BM][Y1PL] slots: the first conv's output tile as the second conv's operand
    if (KS2) {
        const int c2 = live2 ? wave : ctiles2 - 1;
        const __bf16* ab2 = pr2.w2 + ((long)c2 * pr2.nks2 * 64 + lane) * 8;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) A2[ks] = load_frag<__bf16>(ab2 + (size_t)(ks < pr2.nks2 ? ks : pr2.nks2 - 1) * 512);
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int ch = wave * 32 + 16 * pr + 8 * h;
            const bool ok = live2 && ch < pr2.Cout2_p;
            const int chc = ok ? ch : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sc2[pr][j] = 1.0f;
                bs2[pr][j] = 0.0f;
            }
            if (pr2.scale2) load8(pr2.scale2 + chc, sc2[pr]);
            if (pr2.bias2) load8(pr2.bias2 + chc, bs2[pr]);
            y2off[pr] = ok ? (unsigned)c * y2row + (unsigned)ch * 2u : WS_OOB;
        }
        // the handed-over tile's pad slot and the k columns beyond the first conv's channels stay zero: cleared once, never written
        for (int i = threadIdx.x; i < BM * Y1PL; i += NW * 64) reinterpret_cast<uint4*>(y1t)[i] = uint4{0u, 0u, 0u, 0u};
    }
    const bool tail2 = live2 && wave * 32 + 32 > pr2.Cout2;

    // ---- DMA roles.  Instruction j of a region covers its LDS slots 64 j .. 64 j + 63; wave `wave` issues j = wave, wave + NW, ... -------
    const int nix = (abl & 4) ? 0 : (BM * PPRL + 63) >> 6;
    const int nig = has_gate ? (2 * GPR + 63) >> 6 : 0;
    const int nir = RES && !(abl & 16) ? (BM * RPL + 63) >> 6 : 0;
    auto mine = [&](int ni) -> int { return ni > wave ? (ni - wave + NW - 1) / NW : 0; };
    const int kgrp = mine(nix) + mine(nig) + mine(nir);  // DMA instructions of this wave per tile
    const int kst = ((live ? 2 * MT : 0) + (live2 ? 2 * MT2 : 0)) * (abl & 2 ? 0 : 1);  // stores of this wave per tile
    const float rpl_inv = 1.0f / (float)RPL;
    auto issue = [&](int i, int stg) {  // the DMA group of this block's tile i into stage stg
        char* sb = smem + stg * g.stage_bytes;
        const unsigned m0 = row0 + (unsigned)i * (unsigned)BM;
        for (int j = wave; j < nix; j += NW) {
            const int s = j * 64 + lane;
            const int r = s / PPRL, p = s - r * PPRL;  // compile-time divisor
            const unsigned off = (r < BM && p < PPR) ? (m0 + (unsigned)r) * xrow + (unsigned)p * 16u : WS_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (ws_lds_ptr_t)(sb + j * 1024), 16, (int)off, 0, 0, 0);
        }
        if (XF) {
            const unsigned n0 = m0 / (unsigned)S;
            for (int j = wave; j < nig; j += NW) {
                const int s = j * 64 + lane;
                const int r = s / GPR, p = s - r * GPR;
                const unsigned off = (r < 2 && p * 4 < Cin_p) ? (n0 + (unsigned)r) * (unsigned)Cin_p * 4u + (unsigned)p * 16u : WS_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(grsrc, (ws_lds_ptr_t)(sb + g.xreg + j * 1024), 16, (int)off, 0, 0, 0);
            }
        }
        if (RES) {
            for (int j = wave; j < nir; j += NW) {
                const int s = j * 64 + lane;
                const int r = (int)(((float)s + 0.5f) * rpl_inv), p = s - r * RPL;  // exact: s < 2^16
                const unsigned off = (r < BM && p * 8 < Cout_p) ? (m0 + (unsigned)r) * yrow + (unsigned)p * 16u : WS_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rrsrc, (ws_lds_ptr_t)(sb + g.xreg + g.greg + j * 1024), 16, (int)off, 0, 0, 0);
            }
        }
    };
    // x' = swish(x * gate[clip][k]) IN PLACE on a landed tile, each element once (all waves share the work), rounded back to bf16
    auto transform = [&](int i, int stg) {
        if (abl & 32) return;
        char* sb = smem + stg * g.stage_bytes;
        const unsigned m0 = row0 + (unsigned)i * (unsigned)BM;
        const unsigned n0 = m0 / (unsigned)S;
        // gate rows of the tile's clip and the next one: staged with the tile, or (se_on) the block's own two rows kept in stage 0's gate region
        const float* gl = se_on ? reinterpret_cast<const float*>(smem + g.xreg) + (n0 > n_first ? GPR * 4 : 0) : reinterpret_cast<const float*>(sb + g.xreg);
        const int r0 = (int)(m0 - n0 * (unsigned)S);
        const int nsl = BM * PPRL;
        for (int idx = threadIdx.x; idx < nsl; idx += NW * 64) {
            const int r = idx / PPRL, p = idx - r * PPRL;
            if (p < 2 * KS) {
                bf16x8 v = *reinterpret_cast<const bf16x8*>(sb + idx * 16);
                const float* gp = gl + ((r0 + r >= S) ? GPR * 4 : 0) + p * 8;
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
                if (!(abl & 8)) {
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = (float)v[e] * (e < 4 ? g0[e & 3] : g1[e & 3]);
                    if (in_swish) {  // ONE wave-uniform branch around the eight: inside the element loop it compiled to eight selects
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = f[e] * sigmoidf_(f[e]);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (__bf16)f[e];
                }
                *reinterpret_cast<bf16x8*>(sb + idx * 16) = v;
            }
        }
    };
    if (XF && !has_gate) {  // Swish without a gate: rows of ones, written once (nothing refills this region)
        for (int sidx = 0; sidx < NS; ++sidx)
            for (int i = threadIdx.x; i < 2 * GPR * 4; i += NW * 64) reinterpret_cast<float*>(smem + sidx * g.stage_bytes + g.xreg)[i] = 1.0f;
    }
    for (int j = 0; j < LA; ++j)
        if (j < nt) issue(j, j);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        if (ks >= nks) A[ks] = zero_frag<__bf16>();  // wave-uniform; only the template steps beyond w_kc
    if (KS2) {
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks)
            if (ks >= pr2.nks2) A2[ks] = zero_frag<__bf16>();
    }
    if (se_on && (abl & 64)) {
        float* G = reinterpret_cast<float*>(smem + g.xreg);
        for (int i = threadIdx.x; i < 2 * GPR * 4; i += NW * 64) G[i] = 1.0f;
    } else if (se_on) {
        // both FC weight sets are requested first (they do not depend on the clip), then per clip: mean over positions from the partial rows
        // (fixed order) -> fc1 + ReLU (a wave per hidden unit, lanes over channels) -> fc2 + sigmoid (a thread per channel).  Scratch: the X
        // region of stage 2 (its first DMA group is issued in iteration 0, after this)
        float* G = reinterpret_cast<float*>(smem + g.xreg);
        float* mean = reinterpret_cast<float*>(smem + 2 * g.stage_bytes);  // [2][Cin_p]
        float* hid = mean + 2 * Cin_p;                                      // [2][cse]
        const int C = se.C, cse = se.cse, tid = threadIdx.x;
        const unsigned n_last = (row1 - 1) / (unsigned)S;
        const int ncl = (int)(n_last - n_first) + 1;  // 1 or 2
        float w1r[4][8];  // fc1 rows of this wave's hidden units j = wave + 8 u (cse <= 32), columns lane + 64 k (C <= 512)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = wave + 8 * u, ch = lane + 64 * k;
                w1r[u][k] = se.w1[(j < cse && ch < C) ? j * C + ch : 0];
            }
        f32x4 w2r[8];  // fc2 row of this thread's channel (tid < 512 covers C), cse <= 32 floats
        const float b2r = se.b2[tid < C ? tid : 0];
#pragma unroll
        for (int u = 0; u < 8; ++u) w2r[u] = *reinterpret_cast<const f32x4*>(se.w2 + ((tid < C && 4 * u < cse) ? tid * cse + 4 * u : 0));
        float b1r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) b1r[u] = se.b1[wave + 8 * u < cse ? wave + 8 * u : 0];
        {
            // partial rows of both clips, eight rows each in flight together; every sum in row order (as se_gate_kernel)
            const float* pp0 = se.pool + (long)n_first * se.pool_blocks * Cin_p + (tid < Cin_p ? tid : 0);
            const float* pp1 = pp0 + (ncl > 1 ? (long)se.pool_blocks * Cin_p : 0);
            float sum0 = 0.0f, sum1 = 0.0f;
            for (int b = 0; b < se.pool_blocks; b += 8) {
                float t0[8], t1[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const long o = (long)min(b + e, se.pool_blocks - 1) * Cin_p;
                    t0[e] = pp0[o];
                    t1[e] = pp1[o];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (b + e < se.pool_blocks) {
                        sum0 += t0[e];
                        sum1 += t1[e];
                    }
            }
            if (tid < Cin_p) {
                mean[tid] = sum0 * se.inv_positions;
                if (ncl > 1) mean[Cin_p + tid] = sum1 * se.inv_positions;
            }
        }
        ws_barrier();
        for (int q = 0; q < ncl; ++q) {
            float sacc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ch = lane + 64 * k;
                const float mv = ch < C ? mean[q * Cin_p + ch] : 0.0f;
#pragma unroll
                for (int u = 0; u < 4; ++u) sacc[u] = fmaf((ch < C && wave + 8 * u < cse) ? w1r[u][k] : 0.0f, mv, sacc[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float t = sacc[u];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
                const int j = wave + 8 * u;
                if (lane == 0 && j < cse) hid[q * cse + j] = fmaxf(t + b1r[u], 0.0f);
            }
        }
        ws_barrier();
        for (int q = 0; q < ncl; ++q) {
            float gv = 0.0f;
            if (tid < C) {
                float sacc = b2r;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (4 * u < cse) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) sacc = fmaf(w2r[u][e], hid[q * cse + 4 * u + e], sacc);
                    }
                gv = sigmoidf_(sacc);
            }
            if (tid < GPR * 4) G[q * GPR * 4 + tid] = gv;  // zeros beyond C: the padded k columns
        }
    }
    if (XF && nt > 0) {  // tile 0 is transformed here; tile i + 1 during iteration i, beside the MFMAs and the epilogue of tile i (host: NS = 3)
        ws_wait_all_but(min(LA - 1, nt - 1) * kgrp);
        ws_barrier();
        transform(0, 0);
    }

    int stg = 0;
#pragma unroll 1
    for (int i = 0; i < nt; ++i) {
        const int nxt = stg + 1 == NS ? 0 : stg + 1;
        if (XF) {
            // the pieces of tile i + 1 (transformed in this iteration) have landed: this wave has since issued only the stores of tile i - 1
            // (iteration 0: nothing, the prologue ended with the group of tile 1).  Behind the barrier everyone's have, tile i is
            // transformed by everyone, and nobody still reads the stage of tile i - 1, which takes tile i + 2.
            if (i + 1 < nt) ws_wait_all_but(i > 0 ? kst : 0);
            ws_barrier();
            if (i + 2 < nt) issue(i + 2, nxt + 1 == NS ? 0 : nxt + 1);
        } else {
            // this wave's pieces of tile i have landed: everything it issued since is the groups of the look-ahead tiles and the stores of
            // the last min(i, LA) tiles; then everyone's have, and nobody still reads the stage of tile i - 1, which takes tile i + LA
            ws_wait_all_but(min(LA - 1, nt - 1 - i) * kgrp + min(i, LA) * kst);
            ws_barrier();
            if (i + LA < nt) issue(i + LA, stg + LA >= NS ? stg + LA - NS : stg + LA);
        }
        char* sb = smem + stg * g.stage_bytes;
        const unsigned m0 = row0 + (unsigned)i * (unsigned)BM;
        // (A phase stagger -- waves 4-7 transforming the next tile BEFORE their MFMAs, waves 0-3 after their epilogue, so that the two waves of
        // a SIMD sit in opposite phases -- was measured at +-0: 9664 vs 9643 clips/s, profiles/README round-3 entry 62.)
        if (live) {
            f32x16 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][e] = 0.0f;
            const char* xb = sb + ((pt * MT * 32 + c) * PPRL + h) * 16;
            if (!(abl & 1)) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const bf16x8 b = *reinterpret_cast<const bf16x8*>(xb + (mt * 32 * PPRL + 2 * ks) * 16);
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ks], b, acc[mt], 0, 0, 0);
                    }
                }
            }
            const char* rb = sb + g.xreg + g.greg + ((pt * MT * 32 + c) * RPL + ct * 4 + h) * 16;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    // registers 8 pr + q (channels 32 ct + 16 pr + 4 h + q) and 8 pr + 4 + q (+ 8): after the half-wave exchange lanes < 32 hold
                    // channels 16 pr .. + 7 and lanes >= 32 channels 16 pr + 8 .. + 15 of their position, in (first, second) order
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mt][8 * pr + q]), __float_as_uint(acc[mt][8 * pr + 4 + q]), false, false);
                        v[q] = __uint_as_float(sw[0]);
                        v[4 + q] = __uint_as_float(sw[1]);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[pr][e] + bs[pr][e];
                    if (RES) {
                        float r8[8];
                        load8(reinterpret_cast<const __bf16*>(rb + (mt * 32 * RPL + 2 * pr) * 16), r8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += r8[e];
                    }
                    act_vec(v, act);
                    if (tail) mask_tail(v, Cout - (ct * 32 + 16 * pr + 8 * h));
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
                    if (!(abl & 2))
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_u32x4, o), yrsrc, (int)(yoff[pr] + (m0 + (unsigned)(mt * 32)) * yrow), 0, 0);
                    if (KS2 && yoff[pr] != WS_OOB)  // the same 8 channels of this position, as the second conv's operand
                        *reinterpret_cast<bf16x8*>(y1t + ((pt * 32 + c) * Y1PL + ct * 4 + 2 * pr + h) * 16) = o;
                }
            }
        }
        if (KS2) {
            ws_barrier();  // the block-output tile is complete (and every wave is past its reads of the X tile)
            if (live2) {
                f32x16 acc2[MT2];
#pragma unroll
                for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc2[mt][e] = 0.0f;
                const char* yb = y1t + (c * Y1PL + h) * 16;
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks) {
#pragma unroll
                    for (int mt = 0; mt < MT2; ++mt) {
                        const bf16x8 b = *reinterpret_cast<const bf16x8*>(yb + (mt * 32 * Y1PL + 2 * ks) * 16);
                        acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2[ks], b, acc2[mt], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < MT2; ++mt) {
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc2[mt][8 * pr + q]), __float_as_uint(acc2[mt][8 * pr + 4 + q]), false, false);
                            v[q] = __uint_as_float(sw[0]);
                            v[4 + q] = __uint_as_float(sw[1]);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] * sc2[pr][e] + bs2[pr][e];
                        act_vec(v, pr2.act2);
                        if (tail2) mask_tail(v, pr2.Cout2 - (wave * 32 + 16 * pr + 8 * h));
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
                        if (!(abl & 2))
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_u32x4, o), y2rsrc, (int)(y2off[pr] + (m0 + (unsigned)(mt * 32)) * y2row), 0, 0);
                    }
                }
            }
        }
        if (XF && i + 1 < nt) transform(i + 1, nxt);
        stg = nxt;
    }
}

static int GPR_fits(int ks, int cin_p) { return ks * 16 >= cin_p && 2 * cin_p * 4 + 2 * 32 * 4 <= 32 * (2 * ks + 1) * 16 ? 1 : 0; }  // gate row holds Cin_p floats; scratch fits one X tile

static int ws_ks(int nks) {
    const int opts[] = {4, 6, 8, 12, 14, 16, 28};
    for (int o : opts)
        if (nks <= o) return o;
    return 0;
}

// Geometry of the launch; ok = 0: the layer stays on the other pointwise kernels.
WsGeom pw_ws_geom(const pasn_conv_desc& d, int dtype, bool has_gate, bool has_res, bool se_prologue, const pasn_conv_desc* d2) {
    WsGeom g{};
    if (d2) {  // chained pair: the second conv must be a plain stride-1 pointwise conv on the first one's output
        if (const char* e = tune("PASN_WSPAIR"))
            if (e[0] == '0') return g;
        const bool ok2 = d2->kt == 1 && d2->kh == 1 && d2->kw == 1 && !d2->pt && !d2->ph && !d2->pw && d2->st == 1 && d2->sh == 1 && d2->sw == 1 &&
                         d2->N == d.N && d2->To == d.To && d2->Ho == d.Ho && d2->Wo == d.Wo && d2->Cin == d.Cout && d2->Cin_p == d.Cout_p &&
                         d2->w_frag == 1 && d2->w_kc % 16 == 0 && d2->w_kc >= d2->Cin_p && !d2->in_swish && has_res &&
                         d2->w_rows >= ((d2->Cout_p + 31) / 32) * 32 && (long)d.N * d.To * d.Ho * d.Wo * d2->Cout_p * 2 < (1L << 30);
        if (!ok2) return g;
    }
    if (const char* e = tune("PASN_WS"))
        if (e[0] == '0') return g;
    if (dtype != PASN_BF16 || d.w_frag != 1) return g;
    if (d.kt != 1 || d.kh != 1 || d.kw != 1 || d.pt || d.ph || d.pw || d.st != 1 || d.sh != 1 || d.sw != 1) return g;
    const int mink = tune("PASN_WS_MINK") ? atoi(tune("PASN_WS_MINK")) : 48;  // stage-2 layers (Cin_p 24 / 56): the register-resident kernel is faster (48: lets the 48 -> 216 expand conv in, 10.57 -> 10.60 k clips/s)
    if (d.Cin_p < mink || d.w_kc % 16 != 0 || d.w_kc < d.Cin_p) return g;
    const int nks = d.w_kc / 16, ks = ws_ks(nks);
    if (!ks) return g;
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const int S = d.To * d.Ho * d.Wo;
    if (M * d.Cin_p * 2 >= (1L << 30) || M * d.Cout_p * 2 >= (1L << 30) || (long)d.N * d.Cin_p * 4 >= (1L << 30)) return g;  // 32-bit buffer offsets below the tag
    const int ctiles = (d.Cout_p + 31) / 32;
    if (d.w_rows < ctiles * 32) return g;
    const bool xf = has_gate || d.in_swish;
    if (xf && tune("PASN_WS_GATED") && tune("PASN_WS_GATED")[0] == '0') return g;
    // Routing by measurement (X3D-S at 32 x 16 x 224^2, in the pipeline, us per launch old -> new; profiles/README round-3 entry 59):
    //   project + residual, plain:  108->48 37.5 -> 32, 432->192 22.7 -> 18.8            (216->96 rides the chained pair launch)
    //   project + residual, gated:  432->192 33 -> 27;   108->48 45 -> 54, 216->96 31 -> 34: the in-place transform pass costs more there
    //   expand:                     96->432 38 -> 30.5, 48->216 48 -> 45;   48->108 28 -> 29.5, 192->432 20 -> 21.5, head convs 10.6 -> 12.8
    // PASN_WS=2 takes every layer the kernel covers (the parity tests do).
    const char* mode = tune("PASN_WS");
    if (!(mode && mode[0] == '2') && !d2) {
        bool take;
        if (xf) take = ks == 28 || (se_prologue && ks == 14);  // (with the gate in the prologue the stand-alone gate launch is saved as well)
        else if (has_res) take = ks != 14;
        else take = ctiles >= 7 && ks <= 6;
        if (!take) return g;
    }
    g.KS = ks;
    g.gy = (ctiles + 7) / 8;
    g.CT = (ctiles + g.gy - 1) / g.gy;
    g.PT = g.CT >= 5 ? 1 : g.CT >= 3 ? 2 : g.CT == 2 ? 4 : 8;
    if (const char* e = tune("PASN_WS_PT")) g.PT = max(1, min(8 / g.CT, atoi(e)));
    g.MT = (ks <= 16 && g.PT < 4) ? 2 : 1;
    if (const char* e = tune("PASN_WS_MT")) g.MT = ks <= 16 ? max(1, min(2, atoi(e))) : 1;
    int ks2 = 0, y1bytes = 0;
    if (d2) {  // pair mode: 64-position tiles, (channel tile, position half) per wave in the first conv, channel tile = wave in the second
        const int nks2 = d2->w_kc / 16, ct2 = (d2->Cout_p + 31) / 32;
        ks2 = nks2 <= 4 ? 4 : nks2 <= 6 ? 6 : 0;
        if (!ks2 || !((ks == 14 && ks2 == 6) || (ks == 8 && ks2 == 4)) || g.gy != 1 || g.CT * 2 > 8 || ct2 > 8) return WsGeom{};
        // measured in the pipeline (profiles/README entry 61): 216 -> 96 -> 216 plain 34.6 us vs 36.4 (pwconv_xpair), gated 46 vs 43 -- but 39 vs 53 once
        // the gate is computed in the prologue and its launch is gone; 108 -> 48 -> 108: 70 / 92 us against 61 / 73 for the two separate launches
        const char* pm = tune("PASN_WSPAIR");
        const bool all = pm && pm[0] == '2';
        if (!all && (ks != 14 || (xf && !se_prologue))) return WsGeom{};
        g.PT = 2;
        g.MT = 1;
    }
    auto kib = [](int b) { return (b + 1023) / 1024 * 1024; };
    const int rpl = (d.Cout_p / 8) | 1;
    for (;;) {  // the largest tile whose two stages fit the LDS: halve the sub-tiles per wave, then the waves along the positions
        const int BM = 32 * g.PT * g.MT;
        g.xreg = kib(BM * (2 * ks + 1) * 16);
        g.greg = xf ? kib(2 * ks * 64) : 0;
        g.rreg = has_res ? kib(BM * rpl * 16) : 0;
        g.stage_bytes = g.xreg + g.greg + g.rreg;
        y1bytes = d2 ? kib(BM * (2 * ks2 + 1) * 16) : 0;
        if ((xf ? 3 : 2) * g.stage_bytes + y1bytes <= 160 * 1024) break;  // the input transform works one tile ahead of the MFMAs: three stages
        if (d2) return WsGeom{};
        if (g.MT == 2) g.MT = 1;
        else if (g.PT > 1) g.PT /= 2;
        else return g;
    }
    const int BM = 32 * g.PT * g.MT;
    if (S < BM) return g;  // a tile may touch at most two clips (two staged gate rows)
    g.NS = (xf || 3 * g.stage_bytes + y1bytes <= 150 * 1024) ? 3 : 2;
    if (const char* e = tune("PASN_WS_NS")) g.NS = xf ? 3 : max(2, min(4, atoi(e)));
    while (g.NS > 2 && g.NS * g.stage_bytes + y1bytes > 160 * 1024) --g.NS;
    g.lds_bytes = g.NS * g.stage_bytes + y1bytes;
    g.KS2 = ks2;
    int bpc = g.lds_bytes <= 78 * 1024 ? 2 : 1;
    if (const char* e = tune("PASN_WS_BPC")) bpc = max(1, atoi(e));
    const long max_slots = max(1, 256 * bpc / g.gy);
    long rpb = (M + max_slots - 1) / max_slots;             // equal row shares ...
    if (const char* e = tune_dev("PASN_WS_ROWS")) rpb = max(rpb, (long)atoi(e));
    if (rpb >= 16L * BM) rpb = (rpb + BM - 1) / BM * BM;     // ... whole tiles where the ragged last tile would not matter anyway
    g.rpb = (int)rpb;
    g.nslots = (int)((M + rpb - 1) / rpb);
    if (se_prologue && (rpb > S || 3 * g.stage_bytes > 160 * 1024 || d.Cin_p > 512 || GPR_fits(ks, d.Cin_p) == 0)) return WsGeom{};  // a block touches <= 2 clips; scratch in stage 2
    g.abl = tune_dev("PASN_WS_ABL") ? atoi(tune_dev("PASN_WS_ABL")) : 0;
    g.NW = g.CT * g.PT;
    if (xf && !(tune_dev("PASN_WS_HELP") && tune_dev("PASN_WS_HELP")[0] == '0')) g.NW = 8;
    if (d2) g.NW = 8;  // covers the second conv's channel tiles (<= 8) and the first conv's (channel tile, half) grid  // helper waves: the input transform spread evenly over the four SIMDs
    g.ok = 1;
    return g;
}

int pw_ws_variant(const pasn_conv_desc& d, int dtype, bool has_gate, bool has_res) {
    const WsGeom g = pw_ws_geom(d, dtype, has_gate, has_res, false);
    return g.ok ? 7000 + g.KS * 10 + g.MT : 0;
}

int launch_pw_ws(const void* x, const void* w, const float* scale, const float* bias, const void* res, const float* gate, void* y,
                 const pasn_conv_desc& d, const WsGeom& g, hipStream_t s, const WsSe* sep, const WsPair* pairp) {
    const WsSe se = sep ? *sep : WsSe{nullptr, 0, 0.0f, nullptr, nullptr, nullptr, nullptr, 0, 0};
    const WsPair pr2 = pairp ? *pairp : WsPair{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
    PASN_REQUIRE((g.KS2 != 0) == (pairp != nullptr), "pwconv_ws: pair geometry does not match the call");
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const int S = d.To * d.Ho * d.Wo;
    PASN_REQUIRE(g.ok && g.lds_bytes <= 160 * 1024 && (g.rreg != 0) == (res != nullptr), "pwconv_ws: geometry does not match the call");
    const bool xf = gate != nullptr || d.in_swish != 0 || sep != nullptr;
    PASN_REQUIRE(!(sep && gate), "pwconv_ws: either a gate tensor or the squeeze-excite operands");
    const dim3 grid((unsigned)(g.nslots * g.gy)), block((unsigned)(64 * g.NW));
#define PASN_WS3(KS_, MT_, XF_, RES_) PASN_WS4(KS_, MT_, XF_, RES_, 0)
#define PASN_WS4(KS_, MT_, XF_, RES_, KS2_)                                                                                        \
    do {                                                                                                                          \
        PASN_MAX_LDS(160 * 1024, pwconv_ws_kernel<KS_, MT_, XF_, RES_, KS2_>);                                                   \
        hipLaunchKernelGGL((pwconv_ws_kernel<KS_, MT_, XF_, RES_, KS2_>), grid, block, (size_t)g.lds_bytes, s, (const __bf16*)x,  \
                           (const __bf16*)w, scale, bias, (const __bf16*)res, gate, (__bf16*)y, (int)M, S, d.N, d.Cin_p, d.Cout,   \
                           d.Cout_p, d.w_kc / 16, d.act, d.in_swish, g, se, pr2);                                                 \
    } while (0)
#define PASN_WS2(KS_, MT_)                                   \
    do {                                                     \
        if (xf) {                                            \
            if (res) PASN_WS3(KS_, MT_, true, true);         \
            else PASN_WS3(KS_, MT_, true, false);            \
        } else if (res) PASN_WS3(KS_, MT_, false, true);     \
        else PASN_WS3(KS_, MT_, false, false);               \
    } while (0)
#define PASN_WS1(KS_)                         \
    do {                                      \
        if (g.MT == 2) PASN_WS2(KS_, 2);      \
        else PASN_WS2(KS_, 1);                \
    } while (0)
    if (g.KS2) {  // chained pairs: (216 -> 96 -> 216) and (108 -> 48 -> 108), always with the residual
        if (g.KS == 14 && g.KS2 == 6) {
            if (xf) PASN_WS4(14, 1, true, true, 6);
            else PASN_WS4(14, 1, false, true, 6);
        } else if (g.KS == 8 && g.KS2 == 4) {
            if (xf) PASN_WS4(8, 1, true, true, 4);
            else PASN_WS4(8, 1, false, true, 4);
        } else {
            PASN_REQUIRE(false, "pwconv_ws: no pair instance");
        }
        return check_launch("pwconv_ws_kernel (pair)");
    }
    switch (g.KS) {
        case 4: PASN_WS1(4); break;
        case 6: PASN_WS1(6); break;
        case 8: PASN_WS1(8); break;
        case 12: PASN_WS1(12); break;
        case 14: PASN_WS1(14); break;
        case 16: PASN_WS1(16); break;
        default: PASN_WS2(28, 1); break;
    }
#undef PASN_WS1
#undef PASN_WS2
#undef PASN_WS3
#undef PASN_WS4
    return check_launch("pwconv_ws_kernel");
}

}  // namespace pasn

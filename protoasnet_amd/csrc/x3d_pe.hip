// A block's project conv chained with the NEXT block's expand conv in ONE launch for the WIDE X3D stage (inner width 432, block width 192,
// 7 x 7 planes) -- bf16:   x' = x            (block without squeeze-excite: the stencil applied Swish)
//                          x' = swish(x * gate[clip]),  gate = sigmoid(fc2(relu(fc1(mean over positions))))  computed HERE from the stencil's
//                                                       pool partial rows (squeeze-excite block)
//                          y  = relu(norm_c(conv_c(x')) + residual)            e_next = relu(norm_a(conv_a(y)))
// replaces pasn_conv3d_(se_)fwd + pasn_conv3d_fwd there: 30 + 19-21 us (SE) / 20 + 19 us of launches whose floor is each CU's ingest of the
// layer's weight set and the launch ramp ("What round 3 learned"): one ramp instead of two, the block output never re-read.
//
// pwconv_ws.hip's pair mode keeps BOTH weight sets in registers; at 432 -> 192 -> 432 that is 112 + 96 registers per wave and 14 channel tiles
// for 8 waves: not instantiable.  Here nothing is stationary: a tile's rows are staged ONCE (LDS-DMA), transformed in place, and both convs stream
// their weight fragments from L2 per (32 channels x 64 rows) unit (xb_pointwise.h) -- the launch is bound by that stream (2 x 166 KB per tile,
// each fragment read by two units), which is why only the 432-channel stage, whose blocks hold one tile, routes here.
// Tile = R consecutive rows (R = the block's equal share of the rows, <= 128); a tile touches at most two clips (two gate rows).
// Rounding points and accumulation order are those of the launches replaced: bit-identical results (tests/test_gpu_kernels.py).
#include "common.h"
#include "xb_pointwise.h"

namespace pasn {

struct PeArgs {
    const __bf16* x;          // stencil output [M][Cmp]
    const __bf16* w_c;        // project weights, fragment-major [CTC][KSC][64][8], K zero-padded to KSC (even) steps
    const float *s_c, *b_c;   // [>= 32 CTC]
    const __bf16* res;        // block input [M][Cop]
    __bf16* y;                // block output [M][Cop]
    const __bf16* w_a;        // next expand conv, fragment-major [CTA][KSA][64][8]
    const float *s_a, *b_a;   // [>= 32 CTA]
    __bf16* e_next;           // [M][Cnp]
    const float* pool;        // squeeze-excite: partial rows [N][pool_blocks][Cmp] of the stencil (NULL: no gate)
    int pool_blocks;
    float inv_positions;
    const float *w1, *b1, *w2, *b2;  // fc1 [cse][C] + bias, fc2 [C][cse] + bias
    int C, cse;
    int M, S, Cmp, Cop, Cnp;
};

#ifdef PASN_TUNING
// 100 MHz stamps of block 0 / thread 0 (tuning builds, PASN_PE_STAMPS=1; tools/pe_bench.py prints them): start, tables written, rows landed (+ gate),
// transformed, project done, expand done
__device__ long long pe_stamps[8];
#define PE_STAMP(i) do { if (g.se >= 0 && g.stamps && blockIdx.x == 0 && threadIdx.x == 0) pe_stamps[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PE_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ void pe_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The squeeze-excite gate rows of clips n0 .. n0 + ncl - 1 into G ([2][GP] floats in LDS): pwconv_ws.hip's prologue, thread for thread (same
// summation orders: bit-identical gate rows) -- mean over positions from the stencil's partial rows (fixed order) -> fc1 + ReLU (a wave per
// hidden unit, lanes over channels) -> fc2 + sigmoid (a thread per channel).  NOT inlined: its 70-odd registers of FC weights, live across two
// barriers, made the register allocator spill inside the kernel's MFMA loops (46 VGPRs with four row tiles per unit).  `scratch`: [2][Cmp + cse] floats of LDS.
__device__ __attribute__((noinline)) void pe_se_gate(const float* __restrict__ pool, int pool_blocks, float inv_positions, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2, int C,
                                                     int cse, int Cmp, int n0, int ncl, int GP, float* scratch, float* G) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* mean = scratch;        // [2][Cmp]
    float* hid = mean + 2 * Cmp;  // [2][cse]
    float w1r[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = wave + 8 * u, ch = lane + 64 * k;
            w1r[u][k] = w1[(j < cse && ch < C) ? j * C + ch : 0];
        }
    f32x4 w2r[8];
    const float b2r = b2[tid < C ? tid : 0];
#pragma unroll
    for (int u = 0; u < 8; ++u) w2r[u] = *reinterpret_cast<const f32x4*>(w2 + ((tid < C && 4 * u < cse) ? tid * cse + 4 * u : 0));
    float b1r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) b1r[u] = b1[wave + 8 * u < cse ? wave + 8 * u : 0];
    {
        // (the SUMS run in the fixed order b = 0, 1, ... of pwconv_ws.hip; the loads of both clips' partial rows are in flight together, eight
        // rows each at a time)
        const float* pp0 = pool + (long)n0 * pool_blocks * Cmp + (tid < Cmp ? tid : 0);
        const float* pp1 = pp0 + (ncl > 1 ? (long)pool_blocks * Cmp : 0);
        float sum0 = 0.0f, sum1 = 0.0f;
        for (int b = 0; b < pool_blocks; b += 8) {
            float t0[8], t1[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const long o = (long)min(b + e, pool_blocks - 1) * Cmp;
                t0[e] = pp0[o];
                t1[e] = pp1[o];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (b + e < pool_blocks) {
                    sum0 += t0[e];
                    sum1 += t1[e];
                }
        }
        if (tid < Cmp) {
            mean[tid] = sum0 * inv_positions;
            if (ncl > 1) mean[Cmp + tid] = sum1 * inv_positions;
        }
    }
    pe_barrier();
    for (int q = 0; q < ncl; ++q) {
        float sacc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ch = lane + 64 * k;
            const float mv = ch < C ? mean[q * Cmp + ch] : 0.0f;
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
    pe_barrier();
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
        if (tid < GP) G[q * GP + tid] = gv;  // zeros beyond C: the padded k columns
    }
}

template <int KSC, int KSA, bool SE, int MT>
__global__ __launch_bounds__(512) void x3d_pe_kernel(PeArgs a, PeGeom g) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int GP = KSC * 16;  // floats per staged gate row (zero beyond C)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = threadIdx.x;
    char* const img = smem;                       // [R + slack][DPL] slots: the tile's rows, transformed in place
    char* const xt = smem + g.xt_off;             // [RTn * 32][XPL] slots: the block output (SE prologue scratch before that)
    float* const G = reinterpret_cast<float*>(smem + g.gate_off);  // [2][GP] gate rows of the tile's clip and the next one
    unsigned* const rowtab = reinterpret_cast<unsigned*>(smem + g.tab_off);
    float* const cst = reinterpret_cast<float*>(smem + g.cst_off);
    float* const scp = cst, *const bcp = scp + 32 * g.CTC, *const sap = bcp + 32 * g.CTC, *const bap = sap + 32 * g.CTA;
    const int Cmp = a.Cmp, Cop = a.Cop, M = a.M, S = a.S, R = g.R, RTn = g.RTn, DPL = g.DPL, XPL = g.XPL;

    PE_STAMP(0);
    // No clearing pass: the DMA zero-fills every pad slot of the rows image, the gate rows and tables are written in full, rows beyond the tile
    // only ever feed output rows that are never stored.  One exception: the 64-byte gap behind the image's last row, which that row's last
    // (zero-weight) k-step reads -- it must be finite.
    if (tid < 4) reinterpret_cast<uint4*>(img + R * DPL * 16)[tid] = uint4{0u, 0u, 0u, 0u};
    for (int i = tid; i < 32 * g.CTC; i += 512) {
        scp[i] = a.s_c[i];
        bcp[i] = a.b_c[i];
    }
    for (int i = tid; i < 32 * g.CTA; i += 512) {
        sap[i] = a.s_a[i];
        bap[i] = a.b_a[i];
    }
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x), 0, (unsigned)((long)M * Cmp * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.res), 0, (unsigned)((long)M * Cop * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (unsigned)((long)M * Cop * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t ersrc = __builtin_amdgcn_make_buffer_rsrc(a.e_next, 0, (unsigned)((long)M * a.Cnp * 2), 0x00020000);
    const int PPR = Cmp >> 3;
    const int lbl = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_end = min(g.tiles, (lbl + 1) * g.tpb);
    int gate_clip = -1;  // the clip whose gate sits in G[0] (G[1]: the next clip's)
    __syncthreads();
    PE_STAMP(1);

#pragma unroll 1
    for (int tile = lbl * g.tpb; tile < tile_end; ++tile) {
        const unsigned m0 = (unsigned)tile * (unsigned)R;
        const int n0 = (int)(m0 / (unsigned)S);
        if (tid < RTn * 32) rowtab[tid] = (tid < R && m0 + (unsigned)tid < (unsigned)M) ? m0 + (unsigned)tid : 0xffffffffu;
        // ---- the tile's rows by LDS-DMA: instruction j covers slots 64 j ..; K padding, pad slot and rows beyond M are out-of-range lanes (zeros) ----
        const int nix = (R * DPL + 63) >> 6;
        for (int j = wave; j < nix; j += 8) {
            const int s = j * 64 + lane;
            const int r = s / DPL, p = s - r * DPL;
            const unsigned off = (r < R && p < PPR && m0 + (unsigned)r < (unsigned)M) ? (m0 + (unsigned)r) * (unsigned)(Cmp * 2) + (unsigned)p * 16u : XB_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (xb_lds_ptr_t)(img + j * 1024), 16, (int)off, 0, 0, 0);
        }
        if (SE && n0 != gate_clip) {
            const int n_last = (int)(min((unsigned)M, m0 + (unsigned)R) - 1u) / S;
            pe_se_gate(a.pool, a.pool_blocks, a.inv_positions, a.w1, a.b1, a.w2, a.b2, a.C, a.cse, Cmp, n0, max(1, min(2, n_last - n0 + 1)), GP,
                       reinterpret_cast<float*>(xt), G);
            gate_clip = n0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // the rows have landed, the gate rows and the row table are written
        PE_STAMP(2);
        if (SE) {
            // x' = swish(x * gate[clip][k]) IN PLACE, each element once, rounded back to bf16 (pwconv_ws.hip's input transform)
            const int r0 = (int)(m0 - (unsigned)n0 * (unsigned)S);
            for (int idx = tid; idx < R * DPL; idx += 512) {
                const int r = idx / DPL, p = idx - r * DPL;
                if (p < PPR) {
                    bf16x8 v = *reinterpret_cast<const bf16x8*>(img + idx * 16);
                    const float* gp = G + ((r0 + r >= S) ? GP : 0) + p * 8;
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = (float)v[e] * (e < 4 ? g0[e & 3] : g1[e & 3]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = f[e] * sigmoidf_(f[e]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (__bf16)f[e];
                    *reinterpret_cast<bf16x8*>(img + idx * 16) = v;
                }
            }
            __syncthreads();
        }
        PE_STAMP(3);
        // (Requesting the project conv's first weight burst BEFORE this wait and the transform pass was tried -- a `pre` hook in xb_pointwise --
        // and changed nothing: 9.9 us for the phase either way.  It is bound by its 6 units on 4 SIMDs and their epilogues, not by the burst.)
        xb_pointwise<KSC, MT, true, true>(a.w_c, scp, bcp, img, DPL, g.CTC, RTn, rowtab, rrsrc, yrsrc, Cop, xt, XPL, wave, lane);
        __syncthreads();  // the block-output tile is complete in xt
        PE_STAMP(4);
        xb_pointwise<KSA, MT, false, false>(a.w_a, sap, bap, xt, XPL, g.CTA, RTn, rowtab, ersrc, ersrc, a.Cnp, nullptr, 0, wave, lane);
        __syncthreads();  // nobody reads the images / the row table any more
        PE_STAMP(5);
    }
}

// ---- host -----------------------------------------------------------------------------------------------------------------------------
static bool pe_pointwise(const pasn_conv_desc& d) {
    return d.kt == 1 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && !d.pt && !d.ph && !d.pw;
}

// d1: project conv (inner -> C, residual, ReLU; in_swish with the squeeze-excite gate), d2: the next block's expand conv (C -> inner', ReLU)
PeGeom pe_geom(const pasn_conv_desc& d1, const pasn_conv_desc& d2, int dtype, bool se, int cse) {
    PeGeom g{};
    if (dtype != PASN_BF16) return g;
    if (const char* e = tune("PASN_NO_PE"))
        if (e[0] == '1') return g;
    if (!pe_pointwise(d1) || !pe_pointwise(d2) || d1.act != PASN_ACT_RELU || d2.act != PASN_ACT_RELU || d2.in_swish) return g;
    if ((d1.in_swish != 0) != se) return g;  // the gate comes with Swish (X3D's SE blocks); a plain block's input is final
    if (d2.N != d1.N || d2.To != d1.To || d2.Ho != d1.Ho || d2.Wo != d1.Wo || d2.Cin != d1.Cout || d2.Cin_p != d1.Cout_p) return g;
    if (d1.w_frag != 1 || d2.w_frag != 1 || d1.Cin_p % 8 || d1.Cout_p % 16 || d2.Cout_p % 8) return g;
    g.KSC = (d1.Cin_p + 31) / 32 * 2;
    g.KSA = (d1.Cout_p + 31) / 32 * 2;
    // the instantiated pair: 432 -> 192 -> 432 (K padded to an even number of 16-wide steps by the host).  Narrower pairs stay on the
    // weight-stationary kernel / separate launches: their blocks walk several tiles and re-streaming the weights per tile loses
    if (g.KSC != 28 || g.KSA != 12 || d1.w_kc != g.KSC * 16 || d2.w_kc != g.KSA * 16) return PeGeom{};
    g.CTC = (d1.Cout_p + 31) / 32;
    g.CTA = (d2.Cout_p + 31) / 32;
    if (d1.w_rows < g.CTC * 32 || d2.w_rows < g.CTA * 32) return PeGeom{};
    const long M = (long)d1.N * d1.To * d1.Ho * d1.Wo;
    const int S = d1.To * d1.Ho * d1.Wo;
    if (M * d1.Cin_p * 2 >= (1L << 30) || M * d2.Cout_p * 2 >= (1L << 30)) return PeGeom{};
    if (se && (cse <= 0 || cse > 32 || cse % 4 || d1.Cin > 512)) return PeGeom{};
    long r = (M + 255) / 256;  // equal row shares over the CUs ...
    r = std::max(32L, std::min(128L, r));
    if (r > S) return PeGeom{};  // a tile touches at most two clips
    g.R = (int)r;
    g.RTn = ceil_div(g.R, 32);
    g.DPL = (d1.Cin_p / 8) | 1;
    g.XPL = (d1.Cout_p / 8) | 1;
    auto kib = [](int b) { return (b + 1023) / 1024 * 1024; };
    // [rows image (+ slack: a k-step beyond the channels reads the next row's first slots, the last row the zeroed gap)] [block-output image]
    // [gate rows] [row table] [scale / bias tables]
    // (the image holds R rows, not RTn * 32: the last row tile's rows beyond R read on into the next region -- garbage in rows that are never stored)
    g.xt_off = kib(g.R * g.DPL * 16 + 64);
    const int xtb = kib(std::max(g.RTn * 32 * g.XPL * 16 + 64, (2 * d1.Cin_p + 2 * 32) * 4));
    g.gate_off = g.xt_off + xtb;
    g.tab_off = g.gate_off + kib(2 * g.KSC * 16 * 4);
    g.cst_off = g.tab_off + kib(g.RTn * 32 * 4);
    g.lds_bytes = g.cst_off + kib((64 * g.CTC + 64 * g.CTA) * 4);
    if (g.lds_bytes > 160 * 1024) return PeGeom{};
    g.tiles = (int)((M + g.R - 1) / g.R);
    const int grid = std::min(g.tiles, 256);
    g.tpb = ceil_div(g.tiles, grid);
    g.grid = ceil_div(g.tiles, g.tpb);
    g.se = se ? 1 : 0;
    g.stamps = tune_dev("PASN_PE_STAMPS") ? 1 : 0;
    g.ok = 1;
    return g;
}

int launch_x3d_pe(const void* x, const void* w1, const float* s1, const float* b1, const void* res, const float* pool, int pool_blocks,
                  int positions, const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b, int cse, void* y1,
                  const pasn_conv_desc& d1, const void* w2, const float* s2, const float* b2, void* y2, const pasn_conv_desc& d2, const PeGeom& g,
                  hipStream_t s) {
    const long M = (long)d1.N * d1.To * d1.Ho * d1.Wo;
    PeArgs a{(const __bf16*)x, (const __bf16*)w1, s1, b1, (const __bf16*)res, (__bf16*)y1, (const __bf16*)w2, s2, b2, (__bf16*)y2, pool, pool_blocks,
             positions > 0 ? 1.0f / (float)positions : 0.0f, fc1_w, fc1_b, fc2_w, fc2_b, d1.Cin, cse, (int)M, d1.To * d1.Ho * d1.Wo, d1.Cin_p,
             d1.Cout_p, d2.Cout_p};
    const dim3 grid(g.grid), block(512);
    const bool mt4 = !(tune("PASN_PE_MT") && tune("PASN_PE_MT")[0] == '2');
#define PASN_PE(SE_, MT_)                                                                                           \
    do {                                                                                                            \
        PASN_MAX_LDS(160 * 1024, x3d_pe_kernel<28, 12, SE_, MT_>);                                                  \
        hipLaunchKernelGGL((x3d_pe_kernel<28, 12, SE_, MT_>), grid, block, (size_t)g.lds_bytes, s, a, g);           \
    } while (0)
    if (g.se) {
        if (mt4) PASN_PE(true, 4);
        else PASN_PE(true, 2);
    } else {
        if (mt4) PASN_PE(false, 4);
        else PASN_PE(false, 2);
    }
#undef PASN_PE
    return check_launch("x3d_pe_kernel");
}

}  // namespace pasn

using namespace pasn;

#ifdef PASN_TUNING
extern "C" int pasn_debug_pe_stamps(long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pasn::pe_stamps), sizeof(long long) * 8); }
#endif

static bool pe_desc_ok(const pasn_conv_desc* d) { return d && d->N > 0 && d->To > 0 && d->Ho > 0 && d->Wo > 0 && d->Cin > 0 && d->Cout > 0; }

extern "C" int pasn_x3d_pe_supported(const pasn_conv_desc* d1, const pasn_conv_desc* d2, int dtype, int Cse) {
    if (!pe_desc_ok(d1) || !pe_desc_ok(d2)) return 0;
    return pe_geom(*d1, *d2, dtype, Cse > 0, Cse).ok;
}

extern "C" int pasn_x3d_pe_fwd(const void* x, const void* w1, const float* scale1, const float* bias1, const void* residual,
                               const float* pool_partial, int pool_blocks, int positions, const float* fc1_w, const float* fc1_b,
                               const float* fc2_w, const float* fc2_b, int Cse, void* y1, const pasn_conv_desc* d1, const void* w2,
                               const float* scale2, const float* bias2, void* y2, const pasn_conv_desc* d2, int dtype, void* stream) {
    PASN_REQUIRE(x && w1 && scale1 && bias1 && w2 && scale2 && bias2 && y1 && y2 && residual, "null pointer");
    PASN_REQUIRE(pe_desc_ok(d1) && pe_desc_ok(d2), "bad geometry");
    const bool se = pool_partial != nullptr;
    PASN_REQUIRE(!se || (fc1_w && fc1_b && fc2_w && fc2_b && pool_blocks > 0 && positions > 0 && Cse > 0), "the squeeze-excite gate comes with all of its operands");
    const PeGeom g = pe_geom(*d1, *d2, dtype, se, se ? Cse : 0);
    PASN_REQUIRE(g.ok, "pair not covered (pasn_x3d_pe_supported returns 0)");
    return launch_x3d_pe(x, w1, scale1, bias1, residual, pool_partial, pool_blocks, positions, fc1_w, fc1_b, fc2_w, fc2_b, Cse, y1, *d1, w2, scale2, bias2,
                         y2, *d2, g, (hipStream_t)stream);
}

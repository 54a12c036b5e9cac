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
#include <type_traits>

#include "common.h"
#include "xb_pointwise.h"

namespace pasn {

// staged frame image of one quad (dwmfma.hip's layout, tighter on narrow planes): region rows x RW positions x 10 slots of 16 bytes (8 used:
// 64 channels).  RW = 16 for regions up to 14 wide; 10 for the two-rows-per-tile regions (<= 8 wide): 16 KiB instead of 25 per frame, which
// is what lets the ring hold four frames next to the 87 KB stencil-output image of the 432-channel blocks
constexpr int xb_tiles(int rpt) { return rpt == 2 ? 4 : 7; }
constexpr int xb_rows(int rpt) { return xb_tiles(rpt) * rpt - 1 + 3; }
constexpr int xb_rw(int rpt) { return rpt == 2 ? 10 : 16; }
constexpr int xb_ni(int rpt) { return (xb_rows(rpt) * xb_rw(rpt) * 10 + 63) / 64; }  // 1-KiB DMA instructions per frame: 23 / 16

__device__ __forceinline__ void xb_wait_all_but(int n) {  // n wave-uniform: all but this wave's n most recent vector-memory ops are done
    switch (n) {
#define XB_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        XB_W(0) XB_W(1) XB_W(2) XB_W(3) XB_W(4) XB_W(5) XB_W(6) XB_W(7) XB_W(8) XB_W(9) XB_W(10) XB_W(11) XB_W(12)
#undef XB_W
        default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;  // never weaker than asked: at most (NS - 2) x 3 DMA instructions are younger
    }
}

__device__ __forceinline__ void xb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned xb_bf16_bits(float f) {
    const __bf16 b = (__bf16)f;
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}

#ifdef PASN_TUNING
// cycle stamps of block 0 / wave 0 (PASN_BLOCK_ABL bit 64; tools/block_bench.py prints them): [0] start, [1] LDS cleared, then per tile
// 8 entries: D start, D end, P end, E end, cycles inside the D-step waits + barriers, steps, -, -
__device__ long long xb_stamps[8 + 8 * 16];
#define XB_STAMP(i) do { if (ABL && (g.abl & 64) && blockIdx.x == 0 && threadIdx.x == 0 && (i) < 8 + 8 * 16) xb_stamps[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define XB_STAMP(i) do { } while (0)
#endif

struct XbArgs {
    const __bf16* e;          // expanded activation of THIS block [N][T][H][W][Cmp]
    const unsigned short* wq; // stencil weight operands [16-channel tile][2 halves][64 lanes][8]: entry e = kt * 5 + j (half e >> 3, slot e & 7) = bf16
                              // bits of the lane's ONE possibly nonzero element of the block-diagonal operand A[kt][pair j], zero where the lane
                              // holds none (host: plan.stencil_operands; the values dwmfma.hip builds in its prologue).  One half of a tile = 1 KiB
                              // = one LDS-DMA instruction
    const float *s_dw, *b_dw; // [Cmp], zero beyond the real channels
    const __bf16* w_c;        // project weights, fragment-major [CTC][KSC][64][8], K zero-padded to KSC (even) steps
    const float *s_c, *b_c;   // [>= 32 CTC]
    const __bf16* res;        // block input [M][Cop]
    __bf16* y;                // block output [M][Cop]
    const __bf16* w_a;        // next expand conv, fragment-major [CTA][KSA][64][8]
    const float *s_a, *b_a;   // [>= 32 CTA]
    __bf16* e_next;           // [M][Cnp]
    int N, T, H, W, Cm, Cmp, Cop, Cnp;
};

// KSA = 0: no chained expand conv (the last block of a stage)
template <int RPT, int KSC, int KSA, int ABL>
__global__ __launch_bounds__(512) void x3d_block_kernel(XbArgs a, XbGeom g) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr bool EXPAND = KSA != 0;
    constexpr int NT = xb_tiles(RPT), NTW = (NT + 1) / 2;
    constexpr int RW = xb_rw(RPT), SLOTS = 10;
    constexpr int NI = xb_ni(RPT), NE = (NI + 7) / 8;
    constexpr int fbytes = NI * 1024;
    constexpr int lstep = RPT * RW * SLOTS * 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, q = lane >> 4;   // stencil roles
    const int ctw = wave & 3, ph = wave >> 2;
    const int abl = ABL ? g.abl : 0;
    char* const ring = smem;                  // [NS][fbytes]; dead after the D phase
    char* const xt = smem;                    // [RTn * 32][XPL] slots: aliases the ring
    char* const dwact = smem + g.dw_off;      // [R + 1][DPL] slots (row R: dump row of the lanes that hold no output)
    unsigned* const rowtab = reinterpret_cast<unsigned*>(smem + g.tab_off);  // [RTn * 32] global position of a tile row, ~0u: none
    float* const cst = reinterpret_cast<float*>(smem + g.cst_off);           // scale / bias tables: stencil [2][Cmp], project [2][32 CTC], expand [2][32 CTA]
    char* const wop = smem + g.wop_off;       // [4 channel tiles][2 halves][64 lanes][16 bytes]: the current quad's stencil weight operands
    const int Cmp = a.Cmp, Cop = a.Cop, T = a.T, H = a.H, W = a.W;
    const int DPL = g.DPL, XPL = g.XPL, TF = g.TF, BW = g.BW, BHW = g.BH * g.BW, R = g.R, RTn = g.RTn, NS = g.NS, LA = g.NS - 1;
    const int nsteps_q = TF + 2;
    float* const sdw = cst, *const bdw = cst + Cmp, *const scp = cst + 2 * Cmp, *const bcp = scp + 32 * g.CTC, *const sap = bcp + 32 * g.CTC,
                *const bap = sap + 32 * g.CTA;

    XB_STAMP(0);
    // the pad slots of the two operand images (k beyond the real channels: the weights there are zero, the activations must be finite)
    for (int i = threadIdx.x; i < g.lds_bytes / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
    for (int i = threadIdx.x; i < Cmp; i += 512) {
        sdw[i] = a.s_dw[i];
        bdw[i] = a.b_dw[i];
    }
    for (int i = threadIdx.x; i < 32 * g.CTC; i += 512) {
        scp[i] = a.s_c[i];
        bcp[i] = a.b_c[i];
    }
    if (EXPAND)
        for (int i = threadIdx.x; i < 32 * g.CTA; i += 512) {
            sap[i] = a.s_a[i];
            bap[i] = a.b_a[i];
        }

    const long fstride = (long)H * W * Cmp;
    const unsigned fr_in_bytes = (unsigned)(fstride * 2);
    const long M = (long)a.N * T * H * W;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.res), 0, (unsigned)(M * Cop * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (unsigned)(M * Cop * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t ersrc = __builtin_amdgcn_make_buffer_rsrc(EXPAND ? a.e_next : a.y, 0, EXPAND ? (unsigned)(M * a.Cnp * 2) : 0u, 0x00020000);
    const int CT16 = (Cmp + 15) >> 4;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wq), 0, (unsigned)CT16 * 2048u, 0x00020000);

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
    const int kdma = NI > wave ? (NI - wave + 7) >> 3 : 0;  // frame-DMA instructions of this wave per step
    const int dwsel = (m & 7) >> 1, wsh = (m & 1) * 16;
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    // the stencil weight operands of quad qd: wave (ctw, ph) fetches half ph of channel tile 4 qd + ctw -- ONE 1-KiB LDS-DMA instruction.  (As register
    // loads, issued a quad ahead, hipcc guarded their use with vmcnt(0): the frame pipeline drained once per quad.)
    auto issue_ops = [&](int qd) {
        const int ct = min(qd * 4 + ctw, CT16 - 1);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (xb_lds_ptr_t)(wop + (ctw * 2 + ph) * 1024), 16, (int)(((ct * 2 + ph) * 64 + lane) * 16), 0, 0, 0);
    };
    __syncthreads();
    XB_STAMP(1);
    int tix = 0;
    long long twait = 0;

#pragma unroll 1
    for (int tile = lbl * g.tpb; tile < tile_end; ++tile, ++tix) {
        // (all stencil state lives inside the tile: nothing of it is held across the pointwise phases, whose weight bursts need the registers)
        f32x4 S0[NTW], S1[NTW], S2[NTW];
#pragma unroll
        for (int l = 0; l < NTW; ++l) S0[l] = S1[l] = S2[l] = zero4;
        const int tch = tile % g.nTch, nr = tile / g.nTch;
        const int reg = nr % regions, n = nr / regions;
        const int rth = reg / g.RTW, rtw = reg - rth * g.RTW;
        const int t0 = tch * TF, h0 = rth * g.BH, w0 = rtw * BW;
        issue_ops(0);  // (older than every frame group: landed when the first group has)
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
        // the DMA group of step s (quad s / (TF + 2), frame t0 - 1 + s % (TF + 2)) into ring slot `slot`
        auto issue = [&](int s, int slot) {
            if (s >= total || (abl & 2)) return;
            const int qd = s / nsteps_q, f = s - qd * nsteps_q;
            const int ti = t0 - 1 + f;
            const bool inclip = ti >= 0 && ti < T;  // a frame outside the clip is staged as zeros (every lane out of range): no second code path
            const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(eclip + qd * 64), 0,
                                                                               (unsigned)T * fr_in_bytes - (unsigned)(qd * 128), 0x00020000);
            const unsigned foff = inclip ? (unsigned)ti * fr_in_bytes : 0u;
            char* dst = ring + slot * fbytes;
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
        for (int j = 0; j < LA; ++j) issue(j, j);
        XB_STAMP(8 + 8 * tix);
        twait = 0;
        long long tparts[3] = {0, 0, 0};
        int slot = 0, islot = LA, qd = 0, f = 0;  // ring slot of step s; ring slot the next group goes to; (quad, frame) of step s
#pragma unroll 1
        for (int s = 0; s < total; ++s) {
#ifdef PASN_TUNING
            long long tw0 = 0;
            if (ABL && (g.abl & 64)) tw0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
            // Group s has landed for this wave -- counted: vmcnt retires in order, everything this wave issued AFTER group s may still fly: the
            // groups of the look-ahead steps that exist and, while it is younger than group s, the operand fetch of the next quad (issued at
            // f == 1).  At f == 0 the operands themselves must have landed: they are older than group s when TF + 1 >= LA, else only the
            // groups issued since then may be outstanding.  Then everyone's have; nobody still reads the slot of step s - 1.
            {
                const int ahead = min(LA - 1, total - 1 - s);
                int n = ahead * kdma;
                if (f == 0) n = min(ahead, nsteps_q - 1) * kdma;
                else if (f - 1 >= 1 && f - 1 <= LA - 1 && qd + 1 < g.NQ) n += 1;
                xb_wait_all_but((abl & 2) ? 0 : n);
            }
            xb_barrier();
#ifdef PASN_TUNING
            if (ABL && (g.abl & 64)) twait += (long long)__builtin_amdgcn_s_memrealtime() - tw0;
#endif
#ifdef PASN_TUNING
            long long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0;
            if (ABL && (g.abl & 64)) tq0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
            if (f == 1 && qd + 1 < g.NQ) issue_ops(qd + 1);  // everyone built quad qd's operands before this step's barrier: the buffer is free
            issue(s + LA, islot);
            if (f == 0) {  // this quad's weight operands and epilogue constants
                const int c0 = (qd * 4 + ctw) * 16;
                wave_live = c0 < Cmp;
                ce = c0 + 4 * q;
                const bool cev = ce < Cmp;
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(sdw + min(ce, Cmp - 4)), b4 = *reinterpret_cast<const f32x4*>(bdw + min(ce, Cmp - 4));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sc[i] = cev ? s4[i] : 0.0f;
                    bs[i] = cev ? b4[i] : 0.0f;
                }
                const uint4 w0v = *reinterpret_cast<const uint4*>(wop + ((ctw * 2 + 0) * 64 + lane) * 16);
                const uint4 w1v = *reinterpret_cast<const uint4*>(wop + ((ctw * 2 + 1) * 64 + lane) * 16);
                const unsigned wd[8] = {w0v.x, w0v.y, w0v.z, w0v.w, w1v.x, w1v.y, w1v.z, w1v.w};
#pragma unroll
                for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const int e = kt * 5 + j;
                        const unsigned u16v = (e & 1) ? (wd[e >> 1] >> 16) : (wd[e >> 1] & 0xffffu);
                        const unsigned bits = wave_live ? (u16v << wsh) : 0u;
                        A[kt][j] = xb_u32x4{dwsel == 0 ? bits : 0u, dwsel == 1 ? bits : 0u, dwsel == 2 ? bits : 0u, dwsel == 3 ? bits : 0u};
                    }
            }
#ifdef PASN_TUNING
            if (ABL && (g.abl & 64)) tq1 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
            if (wave_live && !(abl & 1)) {
                int fbo = slot * fbytes + lbase0 + ph * NTW * lstep;
                asm volatile("" : "+v"(fbo));
                const char* ta[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) ta[j] = ring + fbo + tapoff[j];
                // Only the chains whose output frame lies in this tile run: frame f feeds output t0 + f - 2 through kt = 2 (set S0), t0 + f - 1
                // through kt = 1 (S1), t0 + f through kt = 0 (S2).  With all three chains on every staged frame -- what dwmfma.hip does, at T chunks
                // of 8-16 -- a 2-frame tile did 60 MFMAs per position tile for the 30 it needs.  The sets still rotate through the chains' first
                // MFMA; a skipped chain's set is never read before it is restarted.
                auto frame = [&](auto dop, auto doc, auto don) {
                    constexpr bool DOP = decltype(dop)::value, DOC = decltype(doc)::value, DON = decltype(don)::value;
                    constexpr int NCH = (DOP ? 1 : 0) + (DOC ? 1 : 0) + (DON ? 1 : 0);
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
                            if (DOP) S0[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[2][j]), B, j == 0 ? S1[l] : S0[l], 0, 0, 0);
                            if (DOC) S1[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[1][j]), B, j == 0 ? S2[l] : S1[l], 0, 0, 0);
                            if (DON) S2[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[0][j]), B, j == 0 ? zero4 : S2[l], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 5 * NCH, 0);
                    }
                };
                using T1 = std::true_type;
                using T0 = std::false_type;
                const int mask = ((f >= 2 && f <= TF + 1) ? 4 : 0) | ((f >= 1 && f <= TF) ? 2 : 0) | (f <= TF - 1 ? 1 : 0);  // wave-uniform
                switch (mask) {
                    case 1: frame(T0{}, T0{}, T1{}); break;
                    case 2: frame(T0{}, T1{}, T0{}); break;
                    case 3: frame(T0{}, T1{}, T1{}); break;
                    case 4: frame(T1{}, T0{}, T0{}); break;
                    case 6: frame(T1{}, T1{}, T0{}); break;
                    default: frame(T1{}, T1{}, T1{}); break;
                }
            }
#ifdef PASN_TUNING
            if (ABL && (g.abl & 64)) {
                asm volatile("s_nop 0" ::: "memory");
                tq2 = (long long)__builtin_amdgcn_s_memrealtime();
            }
#endif
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
#ifdef PASN_TUNING
            if (ABL && (g.abl & 64)) {
                tq3 = (long long)__builtin_amdgcn_s_memrealtime();
                tparts[0] += tq1 - tq0;
                tparts[1] += tq2 - tq1;
                tparts[2] += tq3 - tq2;
            }
#endif
            slot = slot + 1 == NS ? 0 : slot + 1;
            islot = islot + 1 == NS ? 0 : islot + 1;
            if (++f == nsteps_q) {
                f = 0;
                ++qd;
            }
        }
        __syncthreads();  // dwact and the row table are complete; the ring is dead (xt may overwrite it)
        XB_STAMP(8 + 8 * tix + 1);
#ifdef PASN_TUNING
        if (ABL && (g.abl & 64) && blockIdx.x == 0 && threadIdx.x == 0 && tix < 16) {
            xb_stamps[8 + 8 * tix + 4] = twait;
            xb_stamps[8 + 8 * tix + 5] = total;
            xb_stamps[8 + 8 * tix + 6] = tparts[0] | (tparts[1] << 20) | (tparts[2] << 40);
        }
#endif

        // ================================ P: project conv + residual + ReLU ==============================================================
        if (!(abl & 4))
            xb_pointwise<KSC, 2, true, EXPAND>(a.w_c, scp, bcp, dwact, DPL, g.CTC, RTn, rowtab, rrsrc, yrsrc, Cop, xt, XPL, wave, lane);
        XB_STAMP(8 + 8 * tix + 2);
        if (EXPAND) {
            __syncthreads();  // the block-output tile is complete in xt
            // ================================ E: the next block's expand conv + ReLU =====================================================
            if (!(abl & 8))
                xb_pointwise<EXPAND ? KSA : 2, 2, false, false>(a.w_a, sap, bap, xt, XPL, g.CTA, RTn, rowtab, ersrc, ersrc, a.Cnp, nullptr, 0, wave, lane);
        }
        __syncthreads();  // nobody reads xt / dwact / the row table any more: the next tile's first frames may land
        XB_STAMP(8 + 8 * tix + 3);
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------------
static bool xb_is_pointwise(const pasn_conv_desc& d) {
    return d.kt == 1 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && !d.pt && !d.ph && !d.pw;
}

// d_dw: the stencil (inner -> inner); d_c: project (inner -> C, + residual, ReLU); d_a: the next block's expand conv (C -> inner'), or NULL
XbGeom xb_geom(const pasn_conv_desc& dd, const pasn_conv_desc& dc, const pasn_conv_desc* da, int dtype) {
    XbGeom g{};
    if (dtype != PASN_BF16) return g;
    // Routing by measurement (profiles/README.md round-4 entries 3-6): at the X3D-S benchmark shapes the fused launch LOSES to the separate
    // launches on every stage (28 x 28: 171 vs 109 us, 14 x 14: 84 vs 63, 7 x 7: 65 vs 55 in isolation) -- its stencil steps serialise wait,
    // DMA issue, MFMAs and the Swish epilogue in eight lock-step waves where dwmfma.hip runs two independent blocks per CU, and the 85-90 KB
    // stencil-output image leaves no LDS for a deeper ring or a second block.  OFF by default; PASN_BLOCK=1: every covered block (what the
    // parity tests run), PASN_BLOCK=5: the 432-channel blocks only.
    const char* mode = tune("PASN_BLOCK");
    if (!mode || mode[0] == '0') return g;
    if (mode[0] == '5' && dd.Cout_p < 432) return g;
    const bool stencil = dd.kt == 3 && dd.kh == 3 && dd.kw == 3 && dd.st == 1 && dd.sh == 1 && dd.sw == 1 && dd.pt == 1 && dd.ph == 1 && dd.pw == 1 &&
                         dd.To == dd.Ti && dd.Ho == dd.Hi && dd.Wo == dd.Wi && dd.Cin_p == dd.Cout_p && dd.Cout_p % 8 == 0 && dd.act == PASN_ACT_SWISH;
    if (!stencil || !xb_is_pointwise(dc) || dc.in_swish || dc.act != PASN_ACT_RELU) return g;
    if (dc.N != dd.N || dc.To != dd.To || dc.Ho != dd.Ho || dc.Wo != dd.Wo || dc.Cin != dd.Cout || dc.Cin_p != dd.Cout_p || dc.w_frag != 1) return g;
    if (dc.Cout_p % 16 != 0 || dc.Cout_p < 48 || dc.Cout_p > 256 || dd.Cout_p > 512) return g;  // (narrower blocks: byte-bound on big planes, the separate launches win)
    g.KSC = (dd.Cout_p + 31) / 32 * 2;
    if (dc.w_kc != g.KSC * 16) return g;  // the host pads K to an EVEN number of 16-wide steps
    g.CTC = (dc.Cout_p + 31) / 32;
    if (dc.w_rows < g.CTC * 32) return g;
    if (da) {
        if (!xb_is_pointwise(*da) || da->in_swish || da->act != PASN_ACT_RELU || da->w_frag != 1) return g;
        if (da->N != dc.N || da->To != dc.To || da->Ho != dc.Ho || da->Wo != dc.Wo || da->Cin != dc.Cout || da->Cin_p != dc.Cout_p) return g;
        g.KSA = (dc.Cout_p + 31) / 32 * 2;
        g.CTA = (da->Cout_p + 31) / 32;
        if (da->w_kc != g.KSA * 16 || da->w_rows < g.CTA * 32 || da->Cout_p % 8 != 0) return g;
    }
    const long M = (long)dd.N * dd.To * dd.Ho * dd.Wo;
    if (M * dd.Cout_p * 2 >= (1L << 30) || (da && M * da->Cout_p * 2 >= (1L << 30)) || (long)dd.Ti * dd.Hi * dd.Wi * dd.Cin_p * 2 >= (1L << 31)) return g;
    if (g.KSC != 8 && g.KSC != 14 && g.KSC != 28) return XbGeom{};                // the instantiated widths: inner 112 / 216 / 432 (X3D stages 3-5)
    if (da && !((g.KSC == 8 && g.KSA == 4) || (g.KSC == 14 && g.KSA == 6) || (g.KSC == 28 && g.KSA == 12))) return XbGeom{};
    g.BW = std::min(dd.Wo, 14);
    g.RPT = g.BW <= 8 ? 2 : 1;
    g.BH = std::min(dd.Ho, g.RPT == 2 ? 8 : 7);
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
    // operand images: every real 8-channel slot + an odd row stride.  A k-step beyond the real channels reads the row's (zeroed) pad slot or the
    // next row's first slots -- finite values under zero weights; the image is followed by a zeroed gap for the last row's sake
    g.DPL = (dd.Cout_p / 8) | 1;
    g.XPL = da ? ((dc.Cout_p / 8) | 1) : 1;
    auto kib = [](int b) { return (b + 1023) / 1024 * 1024; };
    const int fb = xb_ni(g.RPT) * 1024;
    const int xtb = da ? g.RTn * 32 * g.XPL * 16 + 64 : 0;
    const int dwb = kib((g.R + 1) * g.DPL * 16 + 64), tabb = kib(g.RTn * 32 * 4);
    const int cstb = kib((2 * dd.Cout_p + 64 * g.CTC + 64 * g.CTA) * 4), wopb = 8 * 1024;
    g.NS = 4;
    if (const char* e = tune("PASN_BLOCK_NS")) g.NS = std::max(2, std::min(4, atoi(e)));
    while (g.NS > 2 && kib(std::max(g.NS * fb, xtb)) + dwb + tabb + cstb + wopb > 160 * 1024) --g.NS;
    g.dw_off = kib(std::max(g.NS * fb, xtb));
    g.tab_off = g.dw_off + dwb;
    g.cst_off = g.tab_off + tabb;
    g.wop_off = g.cst_off + cstb;
    g.lds_bytes = g.wop_off + wopb;
    if (g.lds_bytes > 160 * 1024) return XbGeom{};
    g.tiles = dd.N * g.RTH * g.RTW * g.nTch;
    const int grid = std::min(g.tiles, 256);
    g.tpb = ceil_div(g.tiles, grid);
    g.grid = ceil_div(g.tiles, g.tpb);
    g.abl = tune_dev("PASN_BLOCK_ABL") ? atoi(tune_dev("PASN_BLOCK_ABL")) : 0;
    g.ok = 1;
    return g;
}

int launch_x3d_block(const void* e, const unsigned short* wq, const float* s_dw, const float* b_dw, const void* w_c, const float* s_c, const float* b_c,
                     const void* res, void* y, const void* w_a, const float* s_a, const float* b_a, void* e_next, const pasn_conv_desc& dd,
                     const pasn_conv_desc& dc, const pasn_conv_desc* da, const XbGeom& g, hipStream_t s) {
    XbArgs a{(const __bf16*)e, wq, s_dw, b_dw, (const __bf16*)w_c, s_c, b_c, (const __bf16*)res, (__bf16*)y, (const __bf16*)w_a, s_a, b_a,
             (__bf16*)e_next, dd.N, dd.To, dd.Ho, dd.Wo, dd.Cout, dd.Cout_p, dc.Cout_p, da ? da->Cout_p : 0};
    const dim3 grid(g.grid), block(512);
#define PASN_XB(RPT_, KSC_, KSA_, ABL_)                                                                                         \
    do {                                                                                                                        \
        PASN_MAX_LDS(160 * 1024, x3d_block_kernel<RPT_, KSC_, KSA_, ABL_>);                                                    \
        hipLaunchKernelGGL((x3d_block_kernel<RPT_, KSC_, KSA_, ABL_>), grid, block, (size_t)g.lds_bytes, s, a, g);             \
    } while (0)
#define PASN_XB_K(RPT_, ABL_)                                         \
    do {                                                              \
        if (g.KSC == 8) {                                             \
            if (da) PASN_XB(RPT_, 8, 4, ABL_);                        \
            else PASN_XB(RPT_, 8, 0, ABL_);                           \
        } else if (g.KSC == 14) {                                     \
            if (da) PASN_XB(RPT_, 14, 6, ABL_);                       \
            else PASN_XB(RPT_, 14, 0, ABL_);                          \
        } else {                                                      \
            if (da) PASN_XB(RPT_, 28, 12, ABL_);                      \
            else PASN_XB(RPT_, 28, 0, ABL_);                          \
        }                                                             \
    } while (0)
#ifdef PASN_TUNING
    if (g.abl) {
        if (g.RPT == 2) PASN_XB_K(2, 1);
        else PASN_XB_K(1, 1);
        return check_launch("x3d_block_kernel (ablation)");
    }
#endif
    if (g.RPT == 2) PASN_XB_K(2, 0);
    else PASN_XB_K(1, 0);
#undef PASN_XB_K
#undef PASN_XB
    return check_launch("x3d_block_kernel");
}

}  // namespace pasn

using namespace pasn;

static bool xb_desc_ok(const pasn_conv_desc* d) { return d && d->N > 0 && d->To > 0 && d->Ho > 0 && d->Wo > 0 && d->Cin > 0 && d->Cout > 0; }

#ifdef PASN_TUNING
extern "C" int pasn_debug_block_stamps(long long* host_out, int n) {  // tuning builds only; not part of the product C-ABI
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pasn::xb_stamps), sizeof(long long) * (size_t)std::min(n, 8 + 8 * 16));
}
#endif

extern "C" int pasn_x3d_block_supported(const pasn_conv_desc* d_dw, const pasn_conv_desc* d_c, const pasn_conv_desc* d_a, int dtype) {
    if (!xb_desc_ok(d_dw) || !xb_desc_ok(d_c) || (d_a && !xb_desc_ok(d_a))) return 0;
    return xb_geom(*d_dw, *d_c, d_a, dtype).ok;
}

extern "C" int pasn_x3d_block_fwd(const void* e, const void* w_dw, const float* scale_dw, const float* bias_dw, const void* w_c,
                                  const float* scale_c, const float* bias_c, const void* residual, void* y, const void* w_a,
                                  const float* scale_a, const float* bias_a, void* e_next, const pasn_conv_desc* d_dw, const pasn_conv_desc* d_c,
                                  const pasn_conv_desc* d_a, int dtype, void* stream) {
    PASN_REQUIRE(e && w_dw && scale_dw && bias_dw && w_c && scale_c && bias_c && residual && y, "null pointer");
    PASN_REQUIRE((d_a != nullptr) == (w_a != nullptr) && (d_a == nullptr || (scale_a && bias_a && e_next)), "the next expand conv comes with all of its operands, or not at all");
    PASN_REQUIRE(xb_desc_ok(d_dw) && xb_desc_ok(d_c) && (!d_a || xb_desc_ok(d_a)), "bad geometry");
    const XbGeom g = xb_geom(*d_dw, *d_c, d_a, dtype);
    PASN_REQUIRE(g.ok, "block not covered (pasn_x3d_block_supported returns 0)");
    return launch_x3d_block(e, (const unsigned short*)w_dw, scale_dw, bias_dw, w_c, scale_c, bias_c, residual, y, w_a, scale_a, bias_a, e_next, *d_dw, *d_c, d_a, g,
                            (hipStream_t)stream);
}

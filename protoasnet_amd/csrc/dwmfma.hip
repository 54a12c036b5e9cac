// Depthwise 3x3x3 stencil (stride 1, pad 1, bf16) on the MATRIX CORES -- third generation of the X3D conv_b stencil, the default for
// every stride-1 layer that does not ride the fused SE-gate launch (dw_mfma_geom; profiles/README entry 45).  (A stride-2 instance --
// staged rows 32 positions wide, 9 slots per position -- was built and measured in round 2: 210 vs 224 us on the first stage, slower on the
// others, -0.8 % end to end; both kernels sit at the fabric's rate there.  Retired in round 3; the findings are profiles/README entry 53.)
//
// Why: both VALU generations (dwmarch.hip and round 2's dwmarch2.hip, retired in round 3) are bound by vector-instruction issue, not by bytes: per output element 27
// fp32 FMAs plus the bf16->fp32 conversions, padding selects and accumulator moves around them (FMAs are ~1/3 of the issued
// instructions; 0.29 of the HBM rate on the 54-channel 56x56 layer).  The matrix cores take bf16 operands as they lie in memory
// and issue beside the vector unit, so the whole inner loop moves there with a BLOCK-DIAGONAL weight operand:
//
//   v_mfma_f32_16x16x32_bf16:  D[16 channels][16 positions] += A[16 channels][K = 32] * B[K = 32][16 positions]
//   K = 2 taps x 16 channels;  B[(tap, c')][p] = x[p + tap][c0 + c'] -- for lane (p = lane & 15, g = lane >> 4) ONE 16-byte read:
//   8 consecutive channels c0 + 8 (g & 1) .. of the position shifted by tap (g >> 1) of the pair, no conversion;
//   A[c][(tap, c')] = w[tap][c0 + c] if c' == c else 0 -- 15 such operands (3 kt x 5 pairs of the 9 (kh, kw) taps), built once per
//   wave and kept in registers.
//
// 1/16 of every MFMA is useful work, which is still 27 useful MACs per 16 x 16 outputs per 15 MFMAs x 16 cycles -- what the packed fp32
// FMAs alone would take if nothing else had to be issued -- and the vector unit is left to the epilogue (scale, bias, Swish, SE partial
// sums, bf16 stores).  As in dwmarch.hip a wave MARCHES ALONG T with three accumulator sets per position tile (outputs t-1, t, t+1):
// every operand read feeds the three kt taps.  The sets have FIXED registers per role and the rotation is done by the MFMAs themselves
// (the first MFMA of a chain reads the previous role's set as C and writes its own): no register moves, 239 VGPRs with 7 tiles per wave.
// The kernel is bound by vector-instruction ISSUE next to the MFMAs (per frame and wave ~180 VALU + ~180 SALU around 105 MFMAs), not by
// MFMA time: what is compiled in (activation), hoisted (pad masking per wave) or counted (vmcnt) below is there for that reason.
//
// Operand supply.  A first version read the B operands straight from global memory (16 positions x 2 x 32-byte pieces per
// wave-instruction): 30 % SLOWER than the VALU kernel, bound by L1 tag lookups (~44 cycles per wave-load, ring depth irrelevant).  Here a
// block owns BH x BW outputs (7 x 14) of one 64-channel quad and stages the (BH+2) x (BW+2) input region of every frame by LDS-DMA
// (whole 128-byte position rows through a buffer descriptor; cells outside the image are out-of-range lanes: zero-filled by the hardware) into a 2-frame ring; B operands are
// ds_read_b128 from a position stride of 160 bytes (10 slots: conflict-free for the read's four 16-lane groups, see the bank rule in
// the guide).  One fence-free barrier per frame; a wave's DMA for frame t+1 is issued right after it, under the 105 MFMAs of frame t.
//
// Weights are rounded to bf16 here (round-to-nearest-even), like the weights of every other bf16 conv of the path; accumulation is
// fp32.  A zero weight times a non-finite activation of ANOTHER channel of the tile would leak (0 x inf); the trunk's activations are
// finite.  Work split: block = 4 waves = the 4 channel tiles of a quad; unit = (T chunk, region); a block walks `upb` units; SE partial
// sums: one row per (clip, chunk), every channel written by exactly one wave, fixed summation order.
#include <type_traits>

#include "common.h"

namespace pasn {

typedef __attribute__((ext_vector_type(4))) unsigned dwm_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned dwm_u32x2;
typedef __attribute__((address_space(3))) void* dwm_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* dwm_gbl_ptr_t;

constexpr int DWF_NTL = 7;      // position tiles per wave (7 x 14 outputs: the X3D planes are 56 / 28 / 14 / 7 high)
constexpr int DWF_RING = 2;      // frame images in LDS: frame t + RING - 1 is requested right after the barrier of frame t (3: measured 4-5 % slower)
// two-row tiles (planes <= 8 wide): 4 x 2 rows, the LDS image stays <= 25 KB; stride 2: 3 rows of 14 outputs = 7 x 29 staged positions
constexpr int dwf_tiles(int rpt, int ss = 1) { return ss == 2 ? 3 : rpt == 2 ? 4 : DWF_NTL; }
constexpr int dwf_pitch(int ss) { return ss == 2 ? 32 : 16; }  // staged positions per region row
// 16-byte slots per staged position (8 used).  The 16 lanes of an operand read are SS positions apart: SS * slots = 2 (mod 4) keeps
// them on different banks (10 at stride 1, 9 at stride 2: 20 slots apart would put lanes m and m + 4 on the same banks)
constexpr int dwf_slots(int ss) { return ss == 2 ? 9 : 10; }
constexpr int dwf_rows(int rpt, int ss) { return (dwf_tiles(rpt, ss) * rpt - 1) * ss + 3; }  // staged rows per region
// 16-byte slots per staged region ROW: the positions' slots + 6 of padding at stride 1 (round 5).  On planes <= 8 wide a position tile holds TWO
// output rows; with rows exactly 160 slots apart the second row's lanes of an operand read fall on the first row's banks: 2-way conflicts on every
// read (tools/pmc_lds_audit.sh: 7.7 LDS cycles per instruction, 48 % conflicts in the two-row instance).  166 (= 6 mod 16) makes the 16 addresses
// of a read cover the 64 banks exactly -- provided the lanes without a position read a cell another lane reads.  One-row tiles are unaffected.
constexpr int dwf_rowp(int ss) { return dwf_pitch(ss) * dwf_slots(ss) + (ss == 1 ? 6 : 0); }
constexpr int dwf_ni(int rpt, int ss) { return (dwf_rows(rpt, ss) * dwf_rowp(ss) + 63) / 64; }  // 1-KiB DMA instructions per frame

__device__ __forceinline__ unsigned bf16_bits_rne(float f) {
    const __bf16 b = (__bf16)f;
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}

__device__ __forceinline__ void dwf_wait_all_but(int n) {  // n wave-uniform: everything but this wave's n most recent vector-memory ops is done
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;  // n <= (DWF_RING - 1) * DWF_NTL + (DWF_RING - 2) * DWF_NE = 21
    }
}

// Barrier of the frame loop WITHOUT the fence of __syncthreads(): that fence is `s_waitcnt vmcnt(0)`, which drains the DMA groups of the
// next frames (and the output stores) at every frame and serialises the ring.  Here only LDS traffic is drained; what must have landed
// from memory is waited for by count (dwf_wait_all_but) just before.
__device__ __forceinline__ void dwf_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ACT: the epilogue activation compiled in (none / Swish: what X3D uses), -1 = the descriptor's.
// STATS (training forward): `pool` receives [N][chunks][2][Cp] = (sum, sum of squares) of the raw outputs per (clip, chunk) -- the batch
// statistics' partial rows in the layout bn_finalize_kernel reads (dwmarch.hip's `stats` launch writes the same).
template <int RPT, bool ABLB = false, int ACT = -1, bool STATS = false>
__global__ __launch_bounds__(256, 2) void dwconv3d_mfma_kernel(const __bf16* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ scale, const float* __restrict__ bias,
                                                               __bf16* __restrict__ y, float* __restrict__ pool, pasn_conv_desc d,
                                                               DwMfmaGeom g, const float* __restrict__ shift) {
    constexpr int SS = 1;  // stride in H and W (the stride-2 instance of round 2 was retired: see the file header)
    extern __shared__ __attribute__((aligned(1024))) char ring[];  // [DWF_RING][NI x 1024]: frame images, position stride 160 bytes
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int n = lb / g.bpc, bx = lb - n * g.bpc;
    const int chunk = bx / g.CQ, cq = bx - chunk * g.CQ;
    const int Cp = d.Cout_p;
    const int c0 = (cq * 4 + wave) * 16;
    const bool wave_live = c0 < Cp;                  // wave-uniform: the last channel quad may be short (the wave still stages and syncs)
    const bool wave_tail = c0 + 16 > d.Cout;         // wave-uniform: this tile holds channels beyond the real count (stored as zeros)
    const int npieces = min(8, (Cp - cq * 64) / 8);  // 16-byte pieces per position of this quad

    // ---- block-diagonal weight operands A[kt][pair]: lane (m, q) holds k = 8q .. 8q+7 = tap (q >> 1) of the pair, channels 8 (q & 1) ..;
    // only element (m & 7) can be nonzero, and only when m's half matches
    dwm_u32x4 A[3][5];
    {
        const int c = c0 + m;
        const bool mine = ((m >> 3) == (q & 1)) && c < Cp;
        const int dwsel = (m & 7) >> 1, sh = (m & 1) * 16;
        // all 15 loads first, from clamped (always valid) addresses: predicated loads became 15 dependent round trips (~35 us per block)
        float wv[3][5];
        const int cc = min(c, Cp - 1);
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int j = 0; j < 5; ++j) wv[kt][j] = w[(kt * 9 + min(2 * j + (q >> 1), 8)) * Cp + cc];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const bool live = mine && 2 * j + (q >> 1) < 9;
                const unsigned bits = live ? (bf16_bits_rne(wv[kt][j]) << sh) : 0u;
                A[kt][j] = dwm_u32x4{dwsel == 0 ? bits : 0u, dwsel == 1 ? bits : 0u, dwsel == 2 ? bits : 0u, dwsel == 3 ? bits : 0u};
            }
    }
    // epilogue constants of this lane's 4 output channels c0 + 4q + i
    const int ce = c0 + 4 * q;
    const bool cev = ce < Cp;
    float sc[4], bs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // (zero for the padded channels: act(0 * P + 0) = 0 for none / ReLU / Swish -- the epilogue stores without a tail mask)
        sc[i] = (cev && ce + i < d.Cout) ? scale[ce + i] : 0.0f;
        bs[i] = (cev && ce + i < d.Cout) ? bias[ce + i] : 0.0f;
    }
    float psum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float psq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    // STATS: moments of (y - k) with a per-channel shift k known BEFORE the launch (the running mean): sum (y - k)^2 does not cancel
    // against the squared mean when |mean| >> std.  NULL: k = 0.
    float kshift[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (STATS && shift && cev) {
#pragma unroll
        for (int i = 0; i < 4; ++i) kshift[i] = ce + i < d.Cout ? shift[ce + i] : 0.0f;  // `shift` holds Cout floats, not Cout_p
    }

    const int Ti = d.Ti, Hi = d.Hi, Wi = d.Wi;
    const long fstride = (long)Hi * Wi * Cp;  // elements per frame
    const __bf16* xclip = x + (long)n * Ti * fstride + cq * 64;
    const unsigned fr_in_bytes = (unsigned)(fstride * 2);
    // this clip from the quad's first channel on; the last bytes of the clip's last row belong to the quad's own channels or lie beyond
    // num_records (channels of LATER quads sit below offset fr_in_bytes * Ti - cq * 128: reading them as "padding slots" is harmless)
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(xclip), 0, (unsigned)Ti * fr_in_bytes - (unsigned)(cq * 128), 0x00020000);
    constexpr int NT = dwf_tiles(RPT, SS);  // position tiles per wave
    const int abl = ABLB ? g.abl : 0;   // timing ablations: a separate instance, the product kernel carries none of the checks
    constexpr int RW = dwf_pitch(SS);  // staged positions per region row ((BW - 1) SS + 3 <= RW used)
    constexpr int SLOTS = dwf_slots(SS);
    constexpr int ROWP = dwf_rowp(SS);  // slots per staged region row
    constexpr int NE = (dwf_ni(RPT, SS) + 3) / 4;  // DMA instructions per wave and frame
    // frame image size is a compile-time constant of the instance (the ring slots, the DMA destinations and the operand reads are then
    // immediates: as run-time scalars they cost ~90 spilled SGPRs, reloaded lane by lane at every frame)
    constexpr int NI = dwf_ni(RPT, SS);
    constexpr int fbytes = NI * 1024;
    // tap offsets of this lane inside the staged region (pair j -> tap 2j + (q >> 1); the absent 10th tap reads the 9th's cell: its
    // weights are zero)
    int tapoff[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int tap9 = min(2 * j + (q >> 1), 8);
        tapoff[j] = ((tap9 / 3) * ROWP + (tap9 % 3) * SLOTS) * 16;
    }
    const int regions = g.RTH * g.RTW;
    const int units = g.nT * regions;
    const int u_end = min(units, (chunk + 1) * g.upb);

#pragma unroll 1
    for (int u = chunk * g.upb; u < u_end; ++u) {
        const int tch = u / regions, reg = u - tch * regions;
        const int rth = reg / g.RTW, rtw = reg - rth * g.RTW;
        const int t0 = tch * g.Tc, t1 = min(t0 + g.Tc, d.To);
        const int h0 = rth * g.BH, w0 = rtw * g.BW;
        // ---- DMA roles: instruction i = wave + 4e covers ring slots 64 i .. 64 i + 63; this lane's slot -> (region position, piece).
        // Buffer addressing (descriptor = this clip's quad of channels, all frames): the source is a wave-uniform frame offset (SGPR) + this
        // lane's 32-bit offset; pieces outside the image carry an out-of-range offset: the hardware fetches nothing and writes ZEROS to their cells.  No predicate, exec juggling or pointer arithmetic per instruction, and every
        // instruction is issued: the wave counts them.
        unsigned goff[NE];  // byte offset inside a frame (this quad), 2^31 = not fetched
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int slot = (wave + 4 * e) * 64 + lane;
            const int rr = slot / ROWP, rem = slot - rr * ROWP;  // staged row, slot inside it (the row's padding slots fetch nothing)
            const int cc = rem / SLOTS, c = rem - cc * SLOTS;
            const int hi = h0 * SS - 1 + rr, wi = w0 * SS - 1 + cc;
            const bool ok = !(abl & 16) && wave + 4 * e < NI && rr * RW < g.RP && cc < (g.BW - 1) * SS + 3 && c < npieces && hi >= 0 && hi < Hi && wi >= 0 && wi < Wi;
            goff[e] = ok ? (unsigned)(((hi * Wi + wi) * Cp + c * 8) * 2) : 0x80000000u;
        }
        const int kdma = max(0, (NI - wave + 3) >> 2);  // DMA instructions of this wave per frame (i = wave + 4e < NI)
        auto staged = [&](int ti) -> bool { return ti >= 0 && ti < Ti && ti >= t0 - 1 && ti <= t1; };
        auto issue = [&](int ti, int slot) {
            if (!staged(ti) || (abl & 2)) return;
            const unsigned foff = (unsigned)ti * fr_in_bytes;  // wave-uniform
            char* dst = ring + slot * fbytes;
#pragma unroll
            for (int e = 0; e < NE; ++e)
                if (wave + 4 * e < NI)  // wave-uniform
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (dwm_lds_ptr_t)(dst + (wave + 4 * e) * 1024), 16, (int)goff[e], (int)foff, 0, 0);
        };
        // ---- fragment roles: position tile l holds RPT whole output rows of the region (rows l RPT ..), lane m -> (row m / BW, column
        // m % BW): every per-tile address is the tile-0 address plus a wave-uniform multiple of l
        const int mrow = m / g.BW, mcol = m - mrow * g.BW;
        const bool lane_ok = m < RPT * g.BW && w0 + mcol < d.Wo && cev;
        const int mrow_lim = lane_ok ? mrow : (1 << 20);  // row of this lane inside its tile, or "never valid"
        const int rows_valid = min(g.BH, d.Ho - h0);                                       // output rows of this region
        const int ntl = (rows_valid + RPT - 1) / RPT;                                      // tiles that hold any of them (wave-uniform)
        const bool mpos = m < RPT * g.BW;  // a lane without a position reads the cell of the tile's last position (a broadcast, not an address of its own)
        const int lbase0 = ((mpos ? mrow : RPT - 1) * SS * ROWP + (mpos ? mcol : g.BW - 1) * SS * SLOTS + 2 * wave + (q & 1)) * 16;
        constexpr int lstep = RPT * SS * ROWP * 16;                                         // bytes between tiles in the staged image: an immediate
        const int ystep = RPT * d.Wo * Cp;
        __bf16* yclip = y + (long)n * d.To * d.Ho * d.Wo * Cp;
        const long ofs = (long)d.Ho * d.Wo * Cp;
        // Output stores go through a per-frame buffer descriptor (num_records = one output frame): rows below the plane fall out of range
        // and are dropped by the hardware, lanes that hold no output position carry an out-of-range offset -- no per-tile predicate, exec
        // juggling or 64-bit address arithmetic in the epilogue, and every tile's store is ISSUED, so the wave can count them: vmcnt retires
        // in issue order, the wait for a frame's DMA group must name every younger DMA and store (otherwise it waits for the previous
        // frame's stores to be acknowledged: ~1.5 us per frame, the whole step serialised)
        const unsigned yvoff = lane_ok ? (unsigned)((((h0 + mrow) * d.Wo + w0 + mcol) * Cp + ce) * 2) : 0x80000000u;
        const unsigned fr_bytes = (unsigned)(ofs * 2);
        const int kst = wave_live ? NT : 0;  // stores per emitted frame: one per tile, whether or not its rows exist

        f32x4 S0[NT], S1[NT], S2[NT];
        const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int l = 0; l < NT; ++l) S0[l] = S1[l] = S2[l] = zero4;

        // one input frame ti from ring slot `slot`: P = output ti-1 (kt = 2), C = output ti (kt = 1), N = output ti+1 (kt = 0)
        // The MFMA chains of one staged frame.  Only the chains whose OUTPUT frame lies in this T chunk run (round 4): frame ti feeds output
        // ti - 1 through kt = 2 (set P), ti through kt = 1 (C), ti + 1 through kt = 0 (N); the two halo frames of a chunk need one chain each,
        // its first and last frame two -- with all three chains on every staged frame a chunk of 8 ran 30 chain-frames for the 24 it needs.
        // The sets still rotate through the chains' first MFMA; a skipped chain's set is never read before it is restarted from zero.
        auto chains = [&](int slot, f32x4 (&P)[NT], f32x4 (&C)[NT], f32x4 (&N)[NT], auto dop, auto doc, auto don) {
            constexpr bool DOP = decltype(dop)::value, DOC = decltype(doc)::value, DON = decltype(don)::value;
            constexpr int NCH = (DOP ? 1 : 0) + (DOC ? 1 : 0) + (DON ? 1 : 0);
            // per-frame operand addresses (kept out of the loop-invariant hoisting: three slots x five taps of them otherwise stay live
            // across the whole march)
            int fbo = slot * fbytes + lbase0;
            asm volatile("" : "+v"(fbo));
            const char* ta[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) ta[j] = ring + fbo + tapoff[j];
            // explicit two-deep operand pipeline: the 5 reads of tile l + 1 are issued before the MFMAs of tile l (left to itself the
            // scheduler serialises read -> lgkmcnt(0) -> 3 MFMAs, one LDS round trip per tap pair: ~2500 cycles per frame)
            bf16x8 Bq[2][5];
#pragma unroll
            for (int j = 0; j < 5; ++j) Bq[0][j] = *reinterpret_cast<const bf16x8*>(ta[j]);
            __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
            for (int l = 0; l < NT; ++l) {
                if (l + 1 < NT) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) Bq[(l + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(ta[j] + (l + 1) * lstep);
                    __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
                }
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const bf16x8 B = Bq[l & 1][j];
                    // The role rotation rides in the first MFMA of every chain (D and C are different registers there): the new P is
                    // the old C plus this frame's kt = 2 taps, the new C the old N plus kt = 1, the new N starts from a constant zero.
                    // No register moves (2 x NT x 4 per frame otherwise).  Order P, C, N: each reads a set before it is overwritten.
                    if (DOP) P[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[2][j]), B, j == 0 ? C[l] : P[l], 0, 0, 0);
                    if (DOC) C[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[1][j]), B, j == 0 ? N[l] : C[l], 0, 0, 0);
                    if (DON) N[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[0][j]), B, j == 0 ? zero4 : N[l], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 5 * NCH, 0);
            }
        };
        // one input frame ti from ring slot `slot`: P = output ti-1 (kt = 2), C = output ti (kt = 1), N = output ti+1 (kt = 0)
        auto frame = [&](int ti, int slot, f32x4 (&P)[NT], f32x4 (&C)[NT], f32x4 (&N)[NT]) {
            if (wave_live && ti >= 0 && ti < Ti && !(abl & 1)) {  // wave-uniform
                using T1 = std::true_type;
                using T0 = std::false_type;
                const bool np = ti - 1 >= t0 && ti - 1 < t1, nc = ti >= t0 && ti < t1, nn = ti + 1 >= t0 && ti + 1 < t1;
                // Specialised bodies only for the 4-tile instance (planes <= 8 wide), and there only for the chunk's two halo frames (one chain each):
                // every extra copy of the 7-tile body costs this compiler 60+ VGPRs (see the note at the step loop below) -- with a copy per
                // mask the 7-tile instance spilled 56, with three copies 57.
                const int mask = RPT == 2 ? ((np ? 4 : 0) | (nc ? 2 : 0) | (nn ? 1 : 0)) : 7;  // wave-uniform
                if (RPT == 2 && mask == 1) chains(slot, P, C, N, T0{}, T0{}, T1{});
                else if (RPT == 2 && mask == 4) chains(slot, P, C, N, T1{}, T0{}, T0{});
                else chains(slot, P, C, N, T1{}, T1{}, T1{});
            } else {  // a frame outside the clip (zero padding in T), or an idle wave: only the roles move on
#pragma unroll
                for (int l = 0; l < NT; ++l) {
                    P[l] = C[l];
                    C[l] = N[l];
                    N[l] = zero4;
                }
            }
            const int to = ti - 1;  // has now seen frames ti-2, ti-1, ti
            if (wave_live && to >= t0 && to < t1 && !(abl & 4)) {
                // straight to memory: a wave owns 32 bytes (16 channels) of each of its 16 positions per store.  (Routing the outputs through
                // an LDS image so that the block stores whole 128-byte rows was measured 2-4 % SLOWER once the stores were counted in
                // the vmcnt wait: 7 ds_write + 7 ds_read + the second pass cost more than the partial lines.)
                int mr = mrow_lim;
                asm volatile("" : "+v"(mr));
                const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(yclip + (long)to * ofs, 0, fr_bytes, 0x00020000);
                // Straight-line over ALL tiles (no per-tile branch: a tile below the plane stores out of the descriptor's range and counts
                // nothing; the pool sums are formed whether or not the launch has a row to write them to; the padded channels carry zero
                // scale and bias instead of a tail mask) -- with three wave-uniform branches per tile every tile's epilogue was its own
                // scheduling region and its store waited for its own arithmetic only
#pragma unroll
                for (int l = 0; l < NT; ++l)
                    {
                        float v[4];
                        const bool ok = l * RPT + mr < rows_valid;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = P[l][i] * sc[i] + bs[i];
                        {
                            if (STATS) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const float dv = ok ? v[i] - kshift[i] : 0.0f;
                                    psum[i] += dv;
                                    psq[i] = fmaf(dv, dv, psq[i]);
                                }
                            } else {
#pragma unroll
                                for (int i = 0; i < 4; ++i) psum[i] += ok ? v[i] : 0.0f;
                            }
                        }
                        // (a run-time activation switch per tile is ~10 scalar branches x NT per frame on a kernel bound by instruction issue)
                        if constexpr (ACT == PASN_ACT_SWISH) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = v[i] * sigmoidf_(v[i]);
                        } else if constexpr (ACT != PASN_ACT_NONE) {
                            act_vec(v, d.act);
                        }
                        if (ACT == -1 && wave_tail) mask_tail(v, d.Cout - ce);  // (run-time activation: sigmoid(0) is not 0)
                        bf16x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(dwm_u32x2, o), yrsrc, (int)yvoff, l * ystep * 2, 0);
                    }
            }
        };
        // one pipeline step: this wave's pieces of frame ti have landed (everything it issued since is the group of frame ti+1), then
        // everyone's have and nobody still reads the slot of frame ti-1, which takes frame ti+2
        auto step = [&](int ti, int slot, f32x4 (&P)[NT], f32x4 (&C)[NT], f32x4 (&N)[NT]) {
            constexpr int LA = DWF_RING - 1;  // frames of look-ahead: group(ti) was issued in step ti - LA, right after that step's barrier
            if (!(abl & 8)) {
                // younger than the group of frame ti: that step's stores (output ti-LA-1), then per later step its group and its stores
                auto stored = [&](int to) -> int { return (to >= t0 && to < t1 && !(abl & 4)) ? kst : 0; };
                int younger = stored(ti - LA - 1);
#pragma unroll
                for (int j = 1; j < LA; ++j) younger += (staged(ti + j) && !(abl & 2) ? kdma : 0) + stored(ti - LA - 1 + j);
                dwf_wait_all_but(younger);
                dwf_barrier();
            }
            issue(ti + LA, slot + LA >= DWF_RING ? slot + LA - DWF_RING : slot + LA);  // that image was last read in step ti-1: everyone is past it
            frame(ti, slot, P, C, N);
        };

        // (no zeroing of the images: out-of-range lanes of `buffer_load ... lds` WRITE ZEROS to their cells -- every cell of a frame image,
        // padding slots included, is rewritten by every frame's DMA; verified by the ragged-shape tests, which fail on stale border cells)
#pragma unroll
        for (int j = 0; j < DWF_RING - 1; ++j) issue(t0 - 1 + j, j);
        // ONE step per iteration with FIXED role registers (S0 = P, S1 = C, S2 = N; the rotation is done by the MFMAs, see frame()).
        // Unrolling by three with the sets passed by name costs 70-250 VGPRs per extra copy of the step with this compiler (7 tiles: 156
        // VGPRs rolled, 225 with two copies, 256 + 225 spilled with three), which is what limited the kernel to 4 tiles per wave.
        int slot = 0;
#pragma unroll 1
        for (int ti = t0 - 1; ti <= ((abl & 64) ? t0 - 2 : t1); ++ti) {
            step(ti, slot, S0, S1, S2);
            slot = slot + 1 == DWF_RING ? 0 : slot + 1;
        }
        __syncthreads();  // nobody reads the ring any more (the next unit zeroes it)
    }

    if (pool && wave_live) {
        // sum over the 16 positions of the tile (lanes sharing q): fixed butterfly order
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s = psum[i];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            s += __shfl_xor(s, 8);
            psum[i] = s;
        }
        if (STATS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = psq[i];
                s += __shfl_xor(s, 1);
                s += __shfl_xor(s, 2);
                s += __shfl_xor(s, 4);
                s += __shfl_xor(s, 8);
                psq[i] = s;
            }
        }
        if (m == 0 && cev) {
            float* pr = pool + ((long)n * g.chunks + chunk) * (STATS ? 2 : 1) * Cp + ce;
            *reinterpret_cast<f32x4*>(pr) = f32x4{psum[0], psum[1], psum[2], psum[3]};
            if (STATS) *reinterpret_cast<f32x4*>(pr + Cp) = f32x4{psq[0], psq[1], psq[2], psq[3]};
        }
    }
}

// Geometry: ok = 0 means "not this kernel".
DwMfmaGeom dw_mfma_geom(const pasn_conv_desc& d, int dtype) {
    DwMfmaGeom g = {};
    if (dtype != PASN_BF16) return g;
    // Default: every stride-1 layer (per launch, with / without the Swish epilogue: 118 vs 150 / 141 vs 170 us at 56 x 56, 57 vs 70 /
    // 68 vs 79 at 28 x 28, 35 vs 41 / 40 vs 44 at 14 x 14, 25 vs 35 / 27 vs 37 at 7 x 7; end to end, three alternating runs each
    // without event bracketing: planes <= 14 wide 8582 clips/s, <= 28 wide 8611, all 8633 -- profiles/README entry 45).
    // PASN_DWMFMA=0: none; PASN_DWMFMA_MAXW: only planes up to this width.
    const char* on = tune("PASN_DWMFMA");
    if (on && on[0] == '0') return g;
    const int maxw = tune("PASN_DWMFMA_MAXW") ? atoi(tune("PASN_DWMFMA_MAXW")) : (1 << 30);
    if (d.Wo > maxw) return g;
    const int ss = 1;
    const bool shape = d.kt == 3 && d.kh == 3 && d.kw == 3 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 1 && d.ph == 1 &&
                       d.pw == 1 && d.To == d.Ti && d.Ho == (d.Hi - 1) / ss + 1 && d.Wo == (d.Wi - 1) / ss + 1 && d.Cin_p == d.Cout_p &&
                       d.Cout_p % 8 == 0;
    if (!shape) return g;
    if ((long)d.Ti * d.Hi * d.Wi * d.Cin_p * 2 >= (1L << 31)) return g;  // one clip per buffer descriptor, 2^31 marks "outside"
    g.CT = ceil_div(d.Cout_p, 16);
    g.CQ = ceil_div(g.CT, 4);
    // region = 7 rows x 14 columns of outputs (the X3D planes are 56 / 28 / 14 / 7 wide and high: no ragged regions); a 16-lane position
    // tile is one output row (14 lanes used) or, on planes at most 8 wide, two
    g.BW = std::min(d.Wo, 14);
    g.RPT = g.BW <= 8 ? 2 : 1;
    g.BH = std::min(d.Ho, dwf_tiles(g.RPT, ss) * g.RPT);
    g.RTH = ceil_div(d.Ho, g.BH);
    g.RTW = ceil_div(d.Wo, g.BW);
    g.RP = ((g.BH - 1) * ss + 3) * dwf_pitch(ss);
    g.NI = dwf_ni(g.RPT, ss);  // the instance's constant
    // (Tc, upb): blocks run two per CU, a block costs a setup (weight operands, pipeline fill) plus upb units of Tc + 2 frames; at most
    // 64 chunks per clip where it costs nothing (the chunk count is the number of SE partial rows the gate has to sum)
    const int force_tc = tune("PASN_DWMFMA_TC") ? atoi(tune("PASN_DWMFMA_TC")) : 0;
    const int force_upb = tune("PASN_DWMFMA_UPB") ? atoi(tune("PASN_DWMFMA_UPB")) : 0;
    const int regions = g.RTH * g.RTW;
    double best = 1e30;
    for (int tc = d.To;; tc = (tc + 1) / 2) {
        const int tcu = force_tc ? std::min(force_tc, d.To) : tc;
        const int nT = ceil_div(d.To, tcu), units = nT * regions;
        for (int upb = 1; upb <= units; ++upb) {
            if (force_upb && upb != std::min(force_upb, units)) continue;
            const int chunks = ceil_div(units, upb);
            if (chunks > 64 && upb < units && !force_upb) continue;
            const long blocks = (long)d.N * g.CQ * chunks;
            const double t = (double)ceil_div(blocks, 512L) * (4.0 + upb * (tcu + 2.0));
            if (t < best) {
                best = t;
                g.Tc = tcu;
                g.nT = nT;
                g.upb = upb;
                g.chunks = chunks;
            }
        }
        if (force_tc || tc <= 4) break;
    }
    g.bpc = g.CQ * g.chunks;
    g.abl = tune_dev("PASN_DWMFMA_ABL") ? atoi(tune_dev("PASN_DWMFMA_ABL")) : 0;  // timing ablations (wrong results)
    g.ok = 1;
    return g;
}

int launch_dw_mfma(const void* x, const float* w, const float* scale, const float* bias, void* y, float* pool, const pasn_conv_desc& d,
                   const DwMfmaGeom& g, hipStream_t s, int stats, const float* shift) {
    const dim3 grid(g.bpc * d.N), block(256);
    const size_t lds = (size_t)DWF_RING * g.NI * 1024;
    if (stats) {  // training forward: raw outputs + (sum, sum of squares) partial rows
        PASN_REQUIRE(pool && d.act == PASN_ACT_NONE && !g.abl, "dwconv3d_mfma: the statistics instance writes the raw conv output");
        if (g.RPT == 2)
            hipLaunchKernelGGL((dwconv3d_mfma_kernel<2, false, PASN_ACT_NONE, true>), grid, block, lds, s, (const __bf16*)x, w, scale, bias, (__bf16*)y, pool, d, g, shift);
        else
            hipLaunchKernelGGL((dwconv3d_mfma_kernel<1, false, PASN_ACT_NONE, true>), grid, block, lds, s, (const __bf16*)x, w, scale, bias, (__bf16*)y, pool, d, g, shift);
        return check_launch("dwconv3d_mfma_kernel (statistics)");
    }
#define PASN_DWF(RPT_, ABL_, ACT_) \
    hipLaunchKernelGGL((dwconv3d_mfma_kernel<RPT_, ABL_, ACT_>), grid, block, lds, s, (const __bf16*)x, w, scale, bias, (__bf16*)y, pool, d, g, (const float*)nullptr)
#ifdef PASN_TUNING
    if (g.abl) {  // timing-ablation instance: -DPASN_TUNING builds only, never in the product library
        PASN_DWF(1, true, -1);
        return check_launch("dwconv3d_mfma_kernel (ablation)");
    }
#endif
    if (g.RPT == 2) {
        if (d.act == PASN_ACT_NONE) PASN_DWF(2, false, PASN_ACT_NONE);
        else if (d.act == PASN_ACT_SWISH) PASN_DWF(2, false, PASN_ACT_SWISH);
        else PASN_DWF(2, false, -1);
    } else {
        if (d.act == PASN_ACT_NONE) PASN_DWF(1, false, PASN_ACT_NONE);
        else if (d.act == PASN_ACT_SWISH) PASN_DWF(1, false, PASN_ACT_SWISH);
        else PASN_DWF(1, false, -1);
    }
#undef PASN_DWF
    return check_launch("dwconv3d_mfma_kernel");
}

}  // namespace pasn

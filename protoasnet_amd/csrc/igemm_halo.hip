// Stride-1 windowed dense convolution (bf16) as an implicit GEMM whose activation tile is a HALO TILE kept in LDS across the taps -- the
// (1,3,3) and (3,1,1) convs of R(2+1)D-18 and the 3x3 convs of ResNet-18 (reference resnet_features.py:49-66,202-213,307-327).
//
// igemm.hip stages one 32-channel slice of ONE tap per step: a (3,1,1) conv fetches every input row three times and a (1,3,3) conv nine
// times, each by a different tile.  PMC on the 144 -> 64 (3,1,1) layer at 8 x 32 x 56 x 56: 632 MB fetched from the fabric for a 231 MB
// input (55 % L2 misses: the three readers of a row sat on three XCDs), 219 us against an HBM floor of ~70, and 5 LDS-DMA wave-instructions
// per 8 MFMAs -- the DMA issue rate, not the matrix pipe, set the pace.  Here a block loads, per 32-channel slice, the rows its outputs need
// for ALL taps once (outputs + halo) and every tap reads its fragments from that tile at a row offset:
//   MODE 0 (kt == 1, "same" kh x kw): 128*MT consecutive flattened positions (n,t,h,w) plus ph*W + pw rows on either side; a tap is the
//          row offset b*W + e.  Flattening wraps at the image borders, so a lane whose tap leaves the image zeroes its fragment (one
//          precomputed bit per tap).  Tile = BM + 2*(W+1) rows instead of 9 * BM.
//   MODE 1 (kt == 3, kh == kw == 1): 4*MT frames x 32 positions plus one frame before and after; a tap is the row offset a*32; frames
//          outside the clip come from the page of zeros at load time.  Tile = (BT+2)*32 rows instead of 3 * BM.
// The weight tile still moves per (tap, slice) step (two stages, one barrier per step); the next slice's halo tile is spread over the
// current slice's steps.  LDS image, swizzle (slot = piece ^ ((row >> 2) & 3), now with the ROW of the shifted read), fragment roles and
// epilogue are igemm.hip's.  DMA issues per wave and step: 3.5 per 20 MFMAs (64 -> 144, was 6.5), 3 per 8 (144 -> 64, was 5).
#include "common.h"
#include "igemm_epilogue.h"

#ifndef PASN_HALO_PIPE
#define PASN_HALO_PIPE 0  // 1: measured no gain (231 vs 228 us on 64 -> 144), 20 more registers
#endif

namespace pasn {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;


constexpr int HG_MAX = 8;  // 16-row DMA groups of the halo tile per wave (tile <= 512 rows)

// SP (MODE 1 only): SLICE-granular pipeline.  The per-tap schedule below gives a (3,1,1) conv steps of 8 MFMAs between barriers, and because
// vmcnt retires in order every step's wait for its weight tile is also a wait for the halo groups issued one step earlier: ~1.2 us per step,
// 15 steps per block, the matrix pipe ~15 % busy.  With SP the whole of slice cs + 1 (halo tile + the three taps' weight tiles, 6 weight
// stages) is requested at the start of slice cs, the three taps run back to back without a barrier, and ONE wait + barrier ends the slice.
template <int NT, int MT, int MODE, int ABL = 0, bool SP = false>  // ABL: timing-only ablation builds (tools), 0 in the product
__global__ __launch_bounds__(256, 2) void igemm_halo_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ w,
                                                            const float* __restrict__ scale, const float* __restrict__ bias,
                                                            const __bf16* __restrict__ res, __bf16* __restrict__ y, pasn_conv_desc d,
                                                            int rows16, int scb_off) {
    constexpr int BN = NT * 32, BM = 128 * MT, BT = 4 * MT;
    constexpr bool PIPE = PASN_HALO_PIPE;
    // (Round 4, tried and removed: a start-up delay for the first dispatch round's second-slot blocks, so that one block's epilogue -- 43 of the
    // 64 -> 144 layer's ~185 event-timed us by the ablation builds, the main loop ~107 -- would run under its CU partner's MFMAs: +-0 at 3-14 us of
    // delay, slower beyond; profiles/README.md entry 117.)
    constexpr int WBYTES = BN * 64, WGROUPS = BN / 16;
    constexpr int OROW = BN + 8;  // epilogue image row (elements)
    extern __shared__ __attribute__((aligned(1024))) char smem[];  // [2][rows16 * 64] halo tiles, [3][WBYTES] weight tiles; the epilogue aliases
    const int XBYTES = rows16 * 64;
    char* const wsm = smem + 2 * XBYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: its branches are uniform
    const int c = lane & 31, h = lane >> 5;
    const int Cin_p = d.Cin_p, Cout_p = d.Cout_p, kc = d.w_kc;
    const int taps = d.kt * d.kh * d.kw;
    const int Ktot = taps * kc;
    const int ncs = (Cin_p + 31) >> 5;
    const int FR = d.Hi * d.Wi, T = d.Ti;
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const int n0 = blockIdx.y * BN;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);  // neighbouring boxes share halo rows: keep them on one L2

    // ---- box of this block ----
    long m0 = 0;          // MODE 0: first flattened output position
    int HS = 0;           // MODE 0: halo rows on either side
    int bn = 0, t0 = 0, p0 = 0;  // MODE 1: clip, first frame, first in-frame position
    if (MODE == 0) {
        m0 = (long)lb * BM;
        HS = d.ph * d.Wi + d.pw;
    } else {
        const int ntc = (T + BT - 1) / BT, nsc = (FR + 31) >> 5;
        const int tc = lb % ntc, r = lb / ntc;
        t0 = tc * BT;
        p0 = (r % nsc) * 32;
        bn = r / nsc;
    }

    // ---- DMA roles: a wave-instruction fills 16 rows x 4 slots; lane -> (row l >> 2, slot l & 3), logical piece q (fixed per lane) ----
    // A wave counts the DMA instructions it issues in a step (wave-uniform) so that the wait at the end of the step can leave exactly
    // those in flight (s_waitcnt vmcnt(n), the immediate picked by a scalar switch).
    constexpr int WD = (WGROUPS + 3) / 4;
    constexpr int PA = MODE == 0 ? 1 : 3;  // halo groups per wave and step: 3x3: <= 8 over 8 of the 9 steps; (3,1,1): <= 6 over 2 of the 3
    const int q = (lane & 3) ^ ((lane >> 4) & 3);
    const int rsub = lane >> 2;
    const int ngroups = rows16 >> 4;
    const int per_wave = (ngroups + 3) >> 2;
    // source row of this lane in halo group g (element offset, or -1 = zeros): recomputed per issue (a handful of integer instructions
    // next to a DMA; a per-lane table of offsets spilled in the 160-channel instance)
    const int sfirst = MODE == 0 ? (int)(m0 - HS) : 0;  // host guarantees M * Cin_p < 2^31
    auto src_off = [&](int g) -> int {
        const int R = g * 16 + rsub;
        if (MODE == 0) {
            const int s = sfirst + R;
            return (R < BM + 2 * HS && (unsigned)s < (unsigned)M) ? s * Cin_p : -1;
        } else {
            const int ft = R >> 5, f = t0 - 1 + ft, sp = p0 + (R & 31);
            return (ft < BT + 2 && (unsigned)f < (unsigned)T && sp < FR) ? ((bn * T + f) * FR + sp) * Cin_p : -1;
        }
    };
    int woff[WD];  // element offset of this lane's weight row + its piece, -1 = zeros
    int wdst[WD];  // wave-uniform LDS offset of the group
#pragma unroll
    for (int i = 0; i < WD; ++i) {
        const int g = wave + 4 * i;
        const int n = n0 + g * 16 + rsub;
        woff[i] = n < d.w_rows ? n * Ktot + q * 8 : -1;
        wdst[i] = g < WGROUPS ? g * 1024 : -1;  // -1: this wave has no i-th group
    }
    // descriptors over the whole activation / weight tensors (the host guarantees both below 2^31 elements: 32-bit byte offsets)
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x), 0, (unsigned)min((long)M * Cin_p * 2, 0xffffffe0L), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(w), 0, (unsigned)min((long)d.w_rows * taps * kc * 2, 0xffffffe0L), 0x00020000);

    auto issue_a = [&](int cs, int i) -> int {  // this wave's i-th halo group of slice cs into stage cs & 1; returns the DMAs issued
        char* xb = smem + (cs & 1) * XBYTES;
        const int ci = cs * 32 + q * 8;
        const int g = wave + 4 * i;
        if (g >= ngroups) return 0;
        const int off = src_off(g);
        const bool ok = ci < Cin_p && off >= 0;
        // buffer addressing: a masked piece carries an out-of-range offset, the hardware then writes ZEROS to its LDS cell (no zero page, no
        // 64-bit pointer select per piece)
        const unsigned vo = ok ? (unsigned)(off + ci) * 2u : 0xfffffff0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(xb + g * 1024), 16, (int)vo, 0, 0, 0);
        return 1;
    };
    auto issue_w = [&](int cs, int tap, int stage) -> int {  // weight tile of step (cs, tap)
        char* wb = wsm + stage * WBYTES;
        const int koff = tap * kc + cs * 32;
        const bool kok = cs * 32 + q * 8 < kc;
        int n = 0;
#pragma unroll
        for (int i = 0; i < WD; ++i) {
            if (wdst[i] >= 0) {  // wave-uniform
                const bool ok = kok && woff[i] >= 0;
                const unsigned vo = ok ? (unsigned)(woff[i] + koff) * 2u : 0xfffffff0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(wb + wdst[i]), 16, (int)vo, 0, 0, 0);
                ++n;
            }
        }
        return n;
    };
    auto wait_all_but = [&](int n) {  // n wave-uniform: everything but the n most recent DMAs of this wave has landed
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;  // stricter than needed for n > 6, never weaker
        }
    };

    // ---- fragment roles ----
    int R0[MT];            // tile row of this lane's position for tap offset 0
    unsigned tmask[MT];    // MODE 0: taps of this lane's position that stay inside the image
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        if (MODE == 0) {
            const int local = wave * MT * 32 + j * 32 + c;
            R0[j] = local;
            tmask[j] = 0;
            const long m = m0 + local;
            if (m < M) {
                const int ow = (int)(m % d.Wo), oh = (int)((m / d.Wo) % d.Ho);
                int tp = 0;
                for (int b2 = 0; b2 < d.kh; ++b2)
                    for (int e = 0; e < d.kw; ++e, ++tp)
                        if ((unsigned)(oh + b2 - d.ph) < (unsigned)d.Hi && (unsigned)(ow + e - d.pw) < (unsigned)d.Wi) tmask[j] |= 1u << tp;
            }
        } else {
            R0[j] = (wave * MT + j) * 32 + c;
            tmask[j] = ~0u;
        }
    }

    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int swc = (c >> 2) & 3;
    const int wrow = c * 64;
    auto mma_step = [&](int cs, int tap, int tb, int te, int buf) {
        const char* xb = smem + (cs & 1) * XBYTES;
        const char* wb = wsm + buf * WBYTES;
        const int tapoff = MODE == 0 ? tb * d.Wi + te : tap * 32;
        const bool edge = MODE == 0 && (tb != d.ph || te != d.pw);  // uniform: the centre tap never leaves the image
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 b[MT], a[NT];
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int row = R0[j] + tapoff;
                const int slot = (2 * ks + h) ^ ((row >> 2) & 3);
                if (ABL & 2) {
                    int z = row + slot;
                    asm volatile("" : "+v"(z));
                    typedef int i32x4 __attribute__((ext_vector_type(4)));
                    b[j] = __builtin_bit_cast(bf16x8, i32x4{z, z, z, z});
                } else
                    b[j] = *reinterpret_cast<const bf16x8*>(xb + row * 64 + slot * 16);
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                if (ABL & 2) {
                    int z = wrow + i + buf;
                    asm volatile("" : "+v"(z));
                    typedef int i32x4 __attribute__((ext_vector_type(4)));
                    a[i] = __builtin_bit_cast(bf16x8, i32x4{z, z, z, z});
                } else
                    a[i] = *reinterpret_cast<const bf16x8*>(wb + wrow + i * 32 * 64 + ((2 * ks + h) ^ swc) * 16);
            }
            if (edge) {
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const bool ok = (tmask[j] >> tap) & 1u;
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 v = __builtin_bit_cast(u32x4, b[j]);
                    v.x = ok ? v.x : 0u;
                    v.y = ok ? v.y : 0u;
                    v.z = ok ? v.z : 0u;
                    v.w = ok ? v.w : 0u;
                    b[j] = __builtin_bit_cast(bf16x8, v);
                }
            }
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) mma32(acc[i][j], a[i], b[j]);
        }
        // machine-scheduler order of the step: the fragment reads of BOTH k-steps first, then the MFMAs back to back (the other wave of the
        // SIMD covers this wave's one LDS wait; left alone hipcc waits twice per step, once per k-step)
        if (PIPE) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MT + NT), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * MT * NT, 0);
        }
    };

    // ---- pipeline: three weight stages (the tile of step s + 2 is issued in step s), two halo stages (slice cs + 1 is spread over the first
    // taps - 1 steps of slice cs).  At the end of step s everything but the DMAs issued IN step s must have landed.
    float* const scb = reinterpret_cast<float*>(smem + scb_off);  // scale | bias of this block's channels, beyond tiles and epilogue image
    igemm_stage_scale_bias<BN>(scb, scale, bias, n0, d.w_rows, tid);
    if constexpr (SP && MODE == 1) {
        for (int i = 0; i < per_wave; ++i) issue_a(0, i);
        for (int tp = 0; tp < 3; ++tp) issue_w(0, tp, tp);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int cs = 0; cs < ncs; ++cs) {
            const int set = (cs & 1) * 3;
            if (cs + 1 < ncs) {  // stages of slice cs - 1: everyone left them at the barrier that ended it
                for (int i = 0; i < per_wave; ++i) issue_a(cs + 1, i);
                for (int tp = 0; tp < 3; ++tp) issue_w(cs + 1, tp, 3 - set + tp);
            }
#pragma unroll
            for (int tp = 0; tp < 3; ++tp) mma_step(cs, tp, tp, 0, set + tp);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else {
    for (int i = 0; i < per_wave; ++i) issue_a(0, i);
    issue_w(0, 0, 0);
    if (taps > 1) issue_w(0, 1, 1);
    else if (ncs > 1) issue_w(1, 0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int stage = 0;            // weight stage of the current step
    int cs2 = 0, tap2 = 2;    // (slice, tap) two steps ahead
    while (tap2 >= taps) {
        tap2 -= taps;
        ++cs2;
    }
    for (int cs = 0; cs < ((ABL & 16) ? 0 : ncs); ++cs) {
        int tap = 0;
        for (int tb = 0; tb < d.kt * d.kh; ++tb)      // MODE 0: tb = kh index; MODE 1: tb = kt index (kw == 1)
            for (int te = 0; te < d.kw; ++te, ++tap) {
                int st2 = stage + 2;
                st2 = st2 >= 3 ? st2 - 3 : st2;
                int issued = 0;
                if (!(ABL & 1) && cs + 1 < ncs && tap + 1 < taps) {
#pragma unroll
                    for (int u = 0; u < PA; ++u) issued += issue_a(cs + 1, tap * PA + u);  // its stage was last read in slice cs - 1
                }
                if (!(ABL & 1) && cs2 < ncs) issued += issue_w(cs2, tap2, st2);  // its stage was last read in step s - 1 (barrier since)
                if (++tap2 == taps) {
                    tap2 = 0;
                    ++cs2;
                }
                mma_step(cs, tap, tb, te, stage);
                // the DMAs of steps < s have landed (in-order completion), i.e. the weight tile of step s + 1 and, after the last tap (which
                // issues no halo group), the whole halo tile of slice cs + 1
                wait_all_but(issued);
                // barrier WITHOUT the fence of __syncthreads(): that fence is `s_waitcnt vmcnt(0)` and drained the DMAs just issued for step
                // s + 2 / slice cs + 1 at every step (pipeline depth 1 instead of 2); only LDS traffic is drained here
                if (!(ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everyone's have; nobody still reads this step's weight stage
                stage = stage == 2 ? 0 : stage + 1;
            }
    }
    }

    // ---- epilogue: scale / bias -> wave-private LDS image of 32 positions x BN channels -> residual + activation + whole-row stores ----
    if ((ABL & 8) && acc[0][0][0] != 1.2345f) return;
    const int width = min(BN, Cout_p - n0);  // channels of this block that exist (multiple of 8)
    const int cgs = width / 8;
    auto tile_rows = [&](int j, long& mbase, int& nvalid) {  // rows of tile j that are output positions
        if (MODE == 0) {
            mbase = m0 + wave * MT * 32 + j * 32;
            nvalid = (int)max(0L, min((long)32, M - mbase));
        } else {
            const int t = t0 + wave * MT + j;
            mbase = ((long)bn * T + t) * FR + p0;
            nvalid = t < T ? min(32, FR - p0) : 0;
        }
    };
    // (the LDS-image epilogue is no longer compiled into these kernels: a second epilogue body behind a run-time flag cost them ~2 %,
    // profiles/README entries 92 / 94; the MFMA stems still use it)
    igemm_epilogue_direct<NT, MT>(acc, scb, res, y, n0, cgs, d, lane, tile_rows);
}

// Geometry of the halo tile for this layer: tile rows (padded to 16) or 0 when the layer is not a stride-1 "same" (1,k,k) / (3,1,1) conv
// or the tile does not fit two blocks per CU.
int igemm_halo_mode(const pasn_conv_desc& d) {
    if (const char* e = tune("PASN_NO_HALO"))
        if (e[0] == '1') return 0;
    if (d.st != 1 || d.sh != 1 || d.sw != 1 || d.To != d.Ti || d.Ho != d.Hi || d.Wo != d.Wi) return 0;
    if (d.kt == 1 && d.pt == 0 && d.kh == 3 && d.kw == 3 && d.ph == 1 && d.pw == 1) return 1;
    if (d.kt == 3 && d.pt == 1 && d.kh == 1 && d.kw == 1 && d.ph == 0 && d.pw == 0) return 2;
    return 0;
}

static int halo_rows16(const pasn_conv_desc& d, int mode, int mt) {
    const int rows = mode == 1 ? 128 * mt + 2 * (d.ph * d.Wi + d.pw) : (4 * mt + 2) * 32;
    return (rows + 15) / 16 * 16;
}

// (3,1,1) layers: slice-granular pipeline (six weight stages) where it fits two blocks per CU; PASN_HALO_SP=0: the per-tap schedule everywhere
static bool halo_sp(int mode, int r16, int nt) {
    if (mode != 2) return false;
    if (const char* e = tune_dev("PASN_HALO_SP"))
        if (e[0] == '0') return false;
    return (size_t)2 * r16 * 64 + (size_t)6 * nt * 32 * 64 + (size_t)nt * 32 * 8 <= 80 * 1024;
}

bool igemm_halo_fits(const pasn_conv_desc& d, int mode, int nt, int mt) {
    const int r16 = halo_rows16(d, mode, mt);
    if (r16 > HG_MAX * 4 * 16) return false;
    if (ceil_div(r16 / 16, 4) > (mode == 1 ? 8 : 6)) return false;  // halo groups per wave the step schedule can place
    const size_t lds = (size_t)2 * r16 * 64 + (size_t)3 * nt * 32 * 64 + (size_t)nt * 32 * 8;
    return lds <= 80 * 1024;
}

int launch_igemm_halo(const void* x, const void* w, const float* scale, const float* bias, const void* res, void* y,
                      const pasn_conv_desc& d, int mode, int nt, int mt, hipStream_t s) {
    const int r16 = halo_rows16(d, mode, mt);
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const long boxes = mode == 1 ? ceil_div(M, 128L * mt) : (long)d.N * ceil_div(d.Ti, 4 * mt) * ceil_div(d.Hi * d.Wi, 32);
    const dim3 grid((unsigned)boxes, ceil_div(d.Cout_p, nt * 32)), block(256);
    const bool sp = halo_sp(mode, r16, nt);
    const size_t tiles = (size_t)2 * r16 * 64 + (size_t)(sp ? 6 : 3) * nt * 32 * 64, image = (size_t)4 * 32 * (nt * 32 + 8) * 2;
    const size_t scb_off = tiles > image ? tiles : image;
    const size_t lds = scb_off + (size_t)nt * 32 * 8;
#define PASN_IH(NT_, MT_, MODE_)                                                                                                  \
    if (nt == NT_ && mt == MT_ && mode == MODE_ + 1) {                                                                            \
        if (lds > 64 * 1024) PASN_MAX_LDS(96 * 1024, igemm_halo_kernel<NT_, MT_, MODE_>);                                         \
        hipLaunchKernelGGL((igemm_halo_kernel<NT_, MT_, MODE_>), grid, block, lds, s, (const __bf16*)x, (const __bf16*)w, scale,  \
                           bias, (const __bf16*)res, (__bf16*)y, d, r16, (int)scb_off);                                               \
        return check_launch("igemm_halo_kernel");                                                                                 \
    }
#ifdef PASN_TUNING  // (seven more instances of the 160-channel kernel: -DPASN_TUNING builds only, never in the product library)
    if (const char* e = tune_dev("PASN_HALO_ABL")) {  // timing-only builds of the 160-channel spatial instance
        const int abl = atoi(e);
#define PASN_IHA(A_)                                                                                                              \
        if (abl == A_ && nt == 5 && mt == 2 && mode == 1) {                                                                       \
            PASN_MAX_LDS(96 * 1024, igemm_halo_kernel<5, 2, 0, A_>);                                                              \
            hipLaunchKernelGGL((igemm_halo_kernel<5, 2, 0, A_>), grid, block, lds, s, (const __bf16*)x, (const __bf16*)w, scale,  \
                               bias, (const __bf16*)res, (__bf16*)y, d, r16, (int)scb_off);                                           \
            return check_launch("igemm_halo_kernel");                                                                             \
        }
        PASN_IHA(1) PASN_IHA(2) PASN_IHA(4) PASN_IHA(8) PASN_IHA(3) PASN_IHA(15) PASN_IHA(16)
#undef PASN_IHA
    }
#endif
#define PASN_IHS(NT_, MT_)                                                                                                        \
    if (sp && nt == NT_ && mt == MT_) {                                                                                           \
        if (lds > 64 * 1024) PASN_MAX_LDS(96 * 1024, igemm_halo_kernel<NT_, MT_, 1, 0, true>);                                    \
        hipLaunchKernelGGL((igemm_halo_kernel<NT_, MT_, 1, 0, true>), grid, block, lds, s, (const __bf16*)x, (const __bf16*)w, scale, \
                           bias, (const __bf16*)res, (__bf16*)y, d, r16, (int)scb_off);                   \
        return check_launch("igemm_halo_kernel (slice pipeline)");                                                                \
    }
    PASN_IHS(2, 2) PASN_IHS(2, 1) PASN_IHS(4, 1)
#undef PASN_IHS
    PASN_IH(2, 2, 0) PASN_IH(2, 1, 0) PASN_IH(4, 2, 0) PASN_IH(4, 1, 0) PASN_IH(5, 2, 0) PASN_IH(5, 1, 0)
    PASN_IH(2, 2, 1) PASN_IH(2, 1, 1) PASN_IH(4, 2, 1) PASN_IH(4, 1, 1) PASN_IH(5, 2, 1) PASN_IH(5, 1, 1)
#undef PASN_IH
    set_error("launch_igemm_halo: no such instance");
    return PASN_ERR_ARG;
}

}  // namespace pasn

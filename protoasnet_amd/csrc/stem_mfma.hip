// X3D stem on the matrix cores (bf16 activations out): (1,3,3) stride-(1,2,2) conv C -> 24, depthwise (5,1,1) temporal conv, BN, ReLU
// as ONE linear map with K = 5 frames x C x 3 rows x 4-wide window slots (the temporal and spatial weights multiplied together by the
// host: W'[co][kt][ci][kr][s] = wt[co][kt] * wxy[co][ci][kr][s]).
//
// The VALU stem (stem.hip) spends 648 fp32 FMAs + a 5-frame register ring per output position: 245 us at 32 x 3 x 16 x 224 x 224, 6 %
// of the step, for 462 MB of traffic (HBM floor ~100 us).  Here a block owns 4 output rows x 64 output columns of one clip and MARCHES
// ALONG T: every input frame's patch (C x 9 rows x 132 columns, bf16, normalised, zero outside the image / the clip) is staged ONCE
// into a ring of 6 LDS slots, and output frame t is 12 k-steps (C = 3; 4 for a grey clip) of v_mfma_f32_32x32x16_bf16 over the five
// slots t-2 .. t+2: per lane a fragment is two 4-byte-aligned 8-byte LDS reads (first_conv_mfma.hip's window-slot trick with a 4-wide
// window: patch columns 2 ow - 2 .. 2 ow + 1 hold the three taps of a stride-2 3-wide window in slots 1 .. 3), the weights sit
// in registers for the whole march.  The next frame's global loads fly under the current frame's MFMAs; one barrier per frame (the
// sixth slot is the one being refilled).  No intermediate tensor, no register ring, no VALU arithmetic except the epilogue.
// Not bit-identical to the two unfused launches (no rounding of the 24-channel intermediate, products of bf16-rounded combined weights):
// a bf16 tolerance path; fp32 activations keep stem.hip.
#include "common.h"
#include "igemm_epilogue.h"

namespace pasn {

// SM_PC: patch columns staged per row = 2 * 64 + 4 (the last window ends at patch column 131).  It was 192: 72.5 KB of LDS, two blocks per
// CU with every wave of a block in the same phase (stage / barrier / MFMA / epilogue) -- nothing above 40 % busy.  132: 53 KB, THREE
// blocks per CU (145-153 VGPRs), a third fewer staging loads: stem 191 -> ~150 us, 8.80 k -> 8.91 k clips/s end to end.
// Round 3: a block owns 56 of the 64 columns its two MFMA column tiles span (112 = 2 x 56: no 64 + 48 split, 116 staged patch columns
// instead of 132; the columns beyond 56 are computed from whatever follows in the ring and never stored): 152 -> 149 us.
constexpr int SM_ROWS = 4, SM_COLS = 56, SM_PC = 116, SM_PR = 2 * (SM_ROWS - 1) + 3, SM_RING = 6;
static_assert(SM_PC % 4 == 0 && SM_PC >= 2 * SM_COLS + 4, "whole 4-column staging units, every window inside the patch");
constexpr int SM_PF = 3;  // k-steps of fragment reads in flight ahead of the MFMAs

template <typename TIN>
struct SmRaw;  // 4 consecutive input values as loaded
template <>
struct SmRaw<float> {
    f32x4 v;
    __device__ void load(const float* p) { v = *reinterpret_cast<const f32x4*>(p); }
    __device__ float get(int e) const { return v[e]; }
};
template <>
struct SmRaw<__bf16> {
    bf16x4 v;
    __device__ void load(const __bf16* p) { v = *reinterpret_cast<const bf16x4*>(p); }
    __device__ float get(int e) const { return (float)v[e]; }
};
template <>
struct SmRaw<unsigned char> {
    unsigned v;
    __device__ void load(const unsigned char* p) { v = *reinterpret_cast<const unsigned*>(p); }
    __device__ float get(int e) const { return (float)((v >> (8 * e)) & 0xffu); }
};

template <typename TIN, int CIN>
__global__ __launch_bounds__(256, 2) void x3d_stem_mfma_kernel(const TIN* __restrict__ x, const __bf16* __restrict__ wq,
                                                               const float* __restrict__ scale, const float* __restrict__ bias,
                                                               __bf16* __restrict__ y, pasn_conv_desc d, float in_a, float in_b) {
    constexpr int NT = 1, MT = 2, BN = 32, OROW = BN + 8;
    constexpr int RPF = CIN * 3;               // patch rows (ci, kr) of the K axis per frame
    constexpr int KROWS = 5 * RPF;             // K rows (kt, ci, kr)
    constexpr int KS = (KROWS + 3) / 4;        // k-steps: four K rows of 4 window slots each (rows 4 s + 2 h, + 1 on lane half h)
    constexpr int SLOT = CIN * SM_PR * SM_PC;  // elements per ring slot
    constexpr int UNITS = SLOT / 4;            // 4-column staging units per frame
    constexpr int UPT = (UNITS + 255) / 256;   // units per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* const ring = reinterpret_cast<__bf16*>(smem);                                      // [SM_RING][CIN][SM_PR][SM_PC]
    __bf16* const imgs = reinterpret_cast<__bf16*>(smem + SM_RING * SLOT * 2);                 // [4 waves][32][OROW]
    float* const scb = reinterpret_cast<float*>(smem + SM_RING * SLOT * 2 + 4 * 32 * OROW * 2);  // [2][BN]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int T = d.Ti;

    const int ncol = (d.Wo + SM_COLS - 1) / SM_COLS, nrow = (d.Ho + SM_ROWS - 1) / SM_ROWS;
    int b = blockIdx.x;
    const int ct = b % ncol;
    b /= ncol;
    const int rt = b % nrow;
    const int n = b / nrow;
    const int oh0 = rt * SM_ROWS, ow0 = ct * SM_COLS;
    const int hi_base = oh0 * 2 - 1, wi_base = ow0 * 2 - 4;  // input coordinates of patch (row 0, column 0): window slot 3 <-> tap 0

    // ---- weights: this lane's A fragments of every k-step (K row 2 s + h, 32 channels x 8 slots per row) ----
    bf16x8 wa[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) wa[s] = *reinterpret_cast<const bf16x8*>(wq + ((size_t)(2 * s + h) * BN + c) * 8);  // row pair 2 s + h
    igemm_stage_scale_bias<BN>(scb, scale, bias, 0, d.Cout_p, tid);

    // ---- staging roles: this thread's units of every frame (same patch coordinates for all frames) ----
    const long plane = (long)d.Hi * d.Wi;
    int uoff[UPT];   // element offset inside a frame plane (channel ci folded in below), -1 = outside the image (zeros)
    int uci[UPT];
#pragma unroll
    for (int k = 0; k < UPT; ++k) {
        const int u = tid + 256 * k;
        const int pc4 = u % (SM_PC / 4), r = (u / (SM_PC / 4)) % SM_PR, ci = u / ((SM_PC / 4) * SM_PR);
        const int hi = hi_base + r, wi = wi_base + pc4 * 4;
        const bool ok = u < UNITS && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi;  // Wi % 4 == 0: whole units
        uoff[k] = ok ? hi * d.Wi + wi : -1;
        uci[k] = ci < CIN ? ci : CIN - 1;  // units past the patch (u >= UNITS) are never stored; keep their address inside the clip
    }
    SmRaw<TIN> raw[UPT];
    auto fetch = [&](int ti) {  // request frame ti's units (ti inside the clip)
#pragma unroll
        for (int k = 0; k < UPT; ++k)
            raw[k].load(x + (((long)n * CIN + uci[k]) * T + ti) * plane + (uoff[k] >= 0 ? uoff[k] : 0));
    };
    auto stage = [&](int slot, bool inside) {  // raw -> normalised bf16 -> ring slot (zeros outside the image / the clip)
        __bf16* dst = ring + slot * SLOT;
#pragma unroll
        for (int k = 0; k < UPT; ++k) {
            const int u = tid + 256 * k;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (inside && uoff[k] >= 0) ? fmaf(raw[k].get(e), in_a, in_b) : 0.0f;
            if (u < UNITS) store4(dst + u * 4, v);
        }
    };

    // ---- fragment roles ----
    // K rows R = kt * RPF + ci * 3 + kr; lane half h of k-step s holds rows 4 s + 2 h and 4 s + 2 h + 1, four window slots each.  In-slot offset of this lane for row (ci, kr), tile 0:
    // ((ci * SM_PR + 2 * wave + kr) * SM_PC + 2 * c); the slot of frame (to - 2 + kt) rotates with the output frame.
    const int lane_off = (2 * wave) * SM_PC + 2 * c + 2;  // window = patch columns 2 (ow - ow0) + 2 .. + 5: taps in slots 1 .. 3
    const int oh = oh0 + wave;
    const int cgs = d.Cout_p / 8;
    __bf16* const img = imgs + (size_t)wave * 32 * OROW;

    // Output frame `to` with its five input frames at ring element offsets sb[0..4] (wave-uniform, run time: the frame loop stays rolled --
    // unrolled by the ring length, hipcc hoisted 6 x 23 per-lane fragment addresses out of the loop and spilled).
    auto emit = [&](int to, const int (&sb)[5]) {
        f32x16 acc[NT][MT];
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // the four K rows of this step (rows past the end are pads: zero weights, any finite values)
            auto rowoff = [&](int R) {
                const int Rc = R < KROWS ? R : KROWS - 1;
                return sb[Rc / RPF] + (((Rc % RPF) / 3) * SM_PR + (Rc % RPF) % 3) * SM_PC;
            };
            const int oa = h ? rowoff(4 * s + 2) : rowoff(4 * s + 0);
            const int ob = h ? rowoff(4 * s + 3) : rowoff(4 * s + 1);
            const __bf16* srca = ring + lane_off + oa;
            const __bf16* srcb = ring + lane_off + ob;
            bf16x8 bfr[MT];
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const unsigned* pa = reinterpret_cast<const unsigned*>(srca + 64 * j);  // 4-byte aligned
                const unsigned* pb = reinterpret_cast<const unsigned*>(srcb + 64 * j);
                u32x4 v;
                v.x = pa[0];
                v.y = pa[1];
                v.z = pb[0];
                v.w = pb[1];
                bfr[j] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) mma32(acc[0][j], wa[s], bfr[j]);
        }
        // Order of the (single basic block) K loop for the machine scheduler: the fragment reads of three k-steps up front, then two MFMAs
        // per four reads -- a prefetch distance of three steps.  Left to itself hipcc issues the four ds_read2_b32 of a step right before
        // its two MFMAs and waits: 23 exposed LDS latencies per frame (235 us for the stem, no better than the VALU kernel).
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT * SM_PF, 0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);
            if (s + SM_PF < KS) __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT, 0);
        }
        igemm_epilogue<NT, MT>(acc, img, scb, nullptr, y, 0, cgs, d, lane, [&](int j, long& mbase, int& nvalid) {
            mbase = (((long)n * d.To + to) * d.Ho + oh) * d.Wo + ow0 + 32 * j;
            nvalid = oh < d.Ho ? min(min(32, SM_COLS - 32 * j), d.Wo - (ow0 + 32 * j)) : 0;  // a block owns SM_COLS columns of its 64-wide MFMA tile
        });
    };

    // ---- march: step f stages input frame ti = f - 2 into slot f % 6 and emits output frame to = f - 4 ----
    // (frames -2, -1, T, T+1 are zeros)
    const int nf = T + 4;
    int r = 0;  // f % SM_RING
#pragma unroll 1
    for (int f = 0; f < nf; ++f) {
        const int ti = f - 2;
        stage(r, ti >= 0 && ti < T);                 // slot r held frame f - 6: last read by output f - 6 (two barriers ago)
        if (ti + 1 >= 0 && ti + 1 < T) fetch(ti + 1);  // lands under this step's MFMAs
        __syncthreads();
        const int to = f - 4;
        if (to >= 0) {
            // frames to-2 .. to+2 were staged in steps f-4 .. f: slots (r + 2 .. r + 6) mod 6
            int sb[5];
#pragma unroll
            for (int kt = 0; kt < 5; ++kt) {
                int q = r + 2 + kt;
                q = q >= SM_RING ? q - SM_RING : q;
                q = q >= SM_RING ? q - SM_RING : q;
                sb[kt] = q * SLOT;
            }
            emit(to, sb);
        }
        r = r + 1 == SM_RING ? 0 : r + 1;
    }
}

int x3d_stem_mfma_supported(const pasn_conv_desc& d, int out_dtype) {
    if (const char* e = tune("PASN_NO_STEM_MFMA"))
        if (e[0] == '1') return 0;
    if (out_dtype != PASN_BF16 || (d.Cin != 1 && d.Cin != 3) || d.Cout_p > 32) return 0;
    if (d.kt != 1 || d.kh != 3 || d.kw != 3 || d.st != 1 || d.sh != 2 || d.sw != 2 || d.pt != 0 || d.ph != 1 || d.pw != 1) return 0;
    if (d.Wi % 4 != 0 || d.To != d.Ti) return 0;
    if ((long)d.N * d.Cin * d.Ti * d.Hi * d.Wi >= (1L << 31)) return 0;
    return 1;
}

template <typename TIN>
static int launch_stem_mfma_t(const void* x, const void* wq, const float* scale, const float* bias, void* y, const pasn_conv_desc& d,
                              float in_a, float in_b, hipStream_t s) {
    const size_t slot = (size_t)d.Cin * SM_PR * SM_PC * 2;
    const size_t lds = SM_RING * slot + (size_t)4 * 32 * (32 + 8) * 2 + 32 * 8;
    const dim3 grid((unsigned)((long)d.N * ceil_div(d.Ho, SM_ROWS) * ceil_div(d.Wo, SM_COLS))), block(256);
    if (d.Cin == 3) {
        PASN_MAX_LDS(96 * 1024, x3d_stem_mfma_kernel<TIN, 3>);
        hipLaunchKernelGGL((x3d_stem_mfma_kernel<TIN, 3>), grid, block, lds, s, (const TIN*)x, (const __bf16*)wq, scale, bias, (__bf16*)y, d,
                           in_a, in_b);
    } else {
        hipLaunchKernelGGL((x3d_stem_mfma_kernel<TIN, 1>), grid, block, lds, s, (const TIN*)x, (const __bf16*)wq, scale, bias, (__bf16*)y, d,
                           in_a, in_b);
    }
    return check_launch("x3d_stem_mfma_kernel");
}

int launch_x3d_stem_mfma(const void* x, const void* wq, const float* scale, const float* bias, void* y, const pasn_conv_desc& d, int in_dtype,
                         float in_a, float in_b, hipStream_t s) {
    if (in_dtype == PASN_F32) return launch_stem_mfma_t<float>(x, wq, scale, bias, y, d, in_a, in_b, s);
    if (in_dtype == PASN_BF16) return launch_stem_mfma_t<__bf16>(x, wq, scale, bias, y, d, in_a, in_b, s);
    if (in_dtype == PASN_U8) return launch_stem_mfma_t<unsigned char>(x, wq, scale, bias, y, d, in_a, in_b, s);
    set_error("x3d_stem_mfma: unknown input dtype");
    return PASN_ERR_ARG;
}

}  // namespace pasn

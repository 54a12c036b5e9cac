// First-layer conv on the matrix cores (bf16 activations out): planar (N,C,T,H,W) clip, C = 3 | 1, window (1,kh,kw) with kw <= 7, stride
// (1,2,2) -- the 7x7 stride-2 stems of R(2+1)D-18 (3 -> 45) and ResNet-18 (3 -> 64) (reference resnet_features.py:203-205, :316-320).
//
// The VALU kernel (conv.hip) spends 147 x Cout fp32 FMAs per output position: 248 us for 8 x 32 x 112 x 112 -> 45 channels, 9 % of the
// R(2+1)D trunk, for 91 MB of traffic (HBM floor ~20 us).  Here the window becomes the K axis of an MFMA:
//   * a block owns 4 output rows x 64 output columns of one frame (one wave per row, two 32-column tiles per wave) and stages the input
//     patch it needs once in LDS as bf16 (normalised: x*a + b; zeros outside the image): patch[c][row][192 columns], column pc <-> input
//     column 2*ow0 - pw - o + pc;
//   * K is ordered (c, kr, j) with j an 8-wide window slot: slot j <-> tap s = j - o, weight 0 outside 0 <= s < kw.  The 8 k values a lane
//     supplies for position ow are then 8 CONSECUTIVE patch columns starting at the even column 2*(ow - ow0): the B fragment is a plain
//     4-byte-aligned 16-byte LDS read, no gather, no packing.  A k-step (16) is two patch rows, one per lane half; the patch row stride
//     (192 columns = 96 dwords = 32 mod 64 banks) keeps the two halves on different banks;
//   * the weights come prepared by the host in exactly this order (bf16 [2*ksteps][64*NT][8]), 16 bytes per lane straight into the A
//     fragment registers, once per block;
//   * epilogue = the implicit-GEMM one (scale / bias / activation, wave-private LDS image, whole-row stores).
#include "common.h"
#include "igemm_epilogue.h"

namespace pasn {

constexpr int FC_ROWS = 4, FC_COLS = 64, FC_PC = 192;

template <typename TIN>
__device__ __forceinline__ void fc_load4(const TIN* p, float (&v)[4]);
template <>
__device__ __forceinline__ void fc_load4<float>(const float* p, float (&v)[4]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = a[j];
}
template <>
__device__ __forceinline__ void fc_load4<__bf16>(const __bf16* p, float (&v)[4]) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)a[j];
}
template <>
__device__ __forceinline__ void fc_load4<unsigned char>(const unsigned char* p, float (&v)[4]) {
    const unsigned a = *reinterpret_cast<const unsigned*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)((a >> (8 * j)) & 0xffu);
}

template <typename TIN, int NT, int KS>  // NT: 32-channel tiles (Cout_p <= 32 NT); KS: k-steps = ceil(C*kh / 2)
__global__ __launch_bounds__(256, 2) void first_conv_mfma_kernel(const TIN* __restrict__ x, const __bf16* __restrict__ wq,
                                                                 const float* __restrict__ scale, const float* __restrict__ bias,
                                                                 __bf16* __restrict__ y, pasn_conv_desc d, float in_a, float in_b, int o) {
    constexpr int MT = 2, BN = NT * 32, OROW = BN + 8;
    const int PR = 2 * (FC_ROWS - 1) + d.kh;  // patch rows per channel
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* const patch = reinterpret_cast<__bf16*>(smem);                                   // [C][PR][FC_PC]
    const int patch_bytes = d.Cin * PR * FC_PC * 2;
    __bf16* const imgs = reinterpret_cast<__bf16*>(smem + patch_bytes);                      // [4 waves][32][OROW]
    float* const scb = reinterpret_cast<float*>(smem + patch_bytes + 4 * 32 * OROW * 2);     // [2][BN]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;

    const int ncol = (d.Wo + FC_COLS - 1) / FC_COLS, nrow = (d.Ho + FC_ROWS - 1) / FC_ROWS;
    int b = blockIdx.x;
    const int ct = b % ncol;
    b /= ncol;
    const int rt = b % nrow;
    b /= nrow;
    const int t = b % d.To, n = b / d.To;
    const int oh0 = rt * FC_ROWS, ow0 = ct * FC_COLS;
    const int hi_base = oh0 * 2 - d.ph, wi_base = ow0 * 2 - d.pw - o;  // input coordinates of patch (row 0, column 0); wi_base % 4 == 0

    // ---- weights: this lane's A fragments of every k-step, 16 bytes each, straight from the prepared matrix ----
    bf16x8 wa[KS][NT];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int i = 0; i < NT; ++i) wa[s][i] = *reinterpret_cast<const bf16x8*>(wq + ((size_t)(2 * s + h) * BN + i * 32 + c) * 8);
    igemm_stage_scale_bias<BN>(scb, scale, bias, 0, d.Cout_p, tid);

    // ---- stage the patch: units of 4 columns; a unit is inside the image or outside it as a whole (Wi % 4 == 0, wi_base % 4 == 0) ----
    const long plane = (long)d.Hi * d.Wi;
    const int units = d.Cin * PR * (FC_PC / 4);
    for (int u0 = tid; u0 < units; u0 += 4 * 256) {
        float v[4][4];
        int dst[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = u0 + k * 256;
            const int pc4 = u % (FC_PC / 4), r = (u / (FC_PC / 4)) % PR, ci = u / ((FC_PC / 4) * PR);
            const int hi = hi_base + r, wi = wi_base + pc4 * 4;
            const bool ok = u < units && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi;
            dst[k] = u < units ? u * 4 : -1;
            if (ok) {
                fc_load4<TIN>(x + (((long)n * d.Cin + ci) * d.Ti + t) * plane + (long)hi * d.Wi + wi, v[k]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[k][e] = fmaf(v[k][e], in_a, in_b);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[k][e] = 0.0f;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (dst[k] >= 0) store4(patch + dst[k], v[k]);
    }
    __syncthreads();

    // ---- main loop: wave = output row oh0 + wave, tiles j = columns ow0 + 32 j .. + 31 ----
    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const __bf16* prow = patch + (2 * wave) * FC_PC + 2 * c;  // patch (row 2*ohl, column 2*(ow - ow0)) of tile 0 for tap row 0
    int qc = 0, qr = h;                               // this lane half's patch row (c, kr) of the current k-step
    while (qr >= d.kh) {
        qr -= d.kh;
        ++qc;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        // (the pad row of an odd C*kh has zero weights: it re-reads the last channel's rows, any finite values)
        const __bf16* src = prow + (min(qc, d.Cin - 1) * PR + qr) * FC_PC;
        bf16x8 bfr[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const unsigned* p32 = reinterpret_cast<const unsigned*>(src + 64 * j);  // 4-byte aligned
            u32x4 v;
            v.x = p32[0];
            v.y = p32[1];
            v.z = p32[2];
            v.w = p32[3];
            bfr[j] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) mma32(acc[i][j], wa[s][i], bfr[j]);
        qr += 2;
        if (qr >= d.kh) {
            qr -= d.kh;
            ++qc;
        }
    }

    // ---- epilogue ----
    const int oh = oh0 + wave;
    const int cgs = d.Cout_p / 8;
    igemm_epilogue<NT, MT>(acc, imgs + (size_t)wave * 32 * OROW, scb, nullptr, y, 0, cgs, d, lane, [&](int j, long& mbase, int& nvalid) {
        mbase = (((long)n * d.To + t) * d.Ho + oh) * d.Wo + ow0 + 32 * j;
        nvalid = oh < d.Ho ? min(32, d.Wo - (ow0 + 32 * j)) : 0;
    });
}

// Slot of tap 0 inside the 8-wide window, or -1 when this layer is not covered: (1,kh,kw) stride (1,2,2), slot + kw <= 8 with the window
// start on a multiple of 4 input columns (slot = (4 - pw) mod 4), Wi % 4 == 0, bf16 output, Cout_p <= 64, C*kh <= 22.
int first_conv_mfma_slot(const pasn_conv_desc& d, int out_dtype) {
    if (const char* e = tune("PASN_NO_FC_MFMA"))
        if (e[0] == '1') return -1;
    if (out_dtype != PASN_BF16 || d.kt != 1 || d.st != 1 || d.pt != 0 || d.sh != 2 || d.sw != 2) return -1;
    if (d.Cin != 1 && d.Cin != 3) return -1;
    if (d.Cout_p > 64 || d.Wi % 4 != 0 || d.kw > 7 || d.pw > 4 || d.Cin * d.kh > 22) return -1;
    const int o = (4 - d.pw % 4) % 4;  // the window starts on a multiple of 4 input columns (aligned staging units, even patch column)
    if (o + d.kw > 8) return -1;
    if (2 * (FC_COLS - 1) + 8 > FC_PC) return -1;
    return o;
}

template <typename TIN>
static int launch_fc_mfma_t(const void* x, const void* wq, const float* scale, const float* bias, void* y, const pasn_conv_desc& d, float in_a,
                            float in_b, int o, hipStream_t s) {
    const int nt = ceil_div(d.Cout_p, 32), ks = ceil_div(d.Cin * d.kh, 2);
    const int PR = 2 * (FC_ROWS - 1) + d.kh;
    const size_t lds = (size_t)d.Cin * PR * FC_PC * 2 + (size_t)4 * 32 * (nt * 32 + 8) * 2 + (size_t)nt * 32 * 8;
    const dim3 grid((unsigned)((long)d.N * d.To * ceil_div(d.Ho, FC_ROWS) * ceil_div(d.Wo, FC_COLS))), block(256);
#define PASN_FCM(NT_, KS_)                                                                                                       \
    if (nt == NT_ && ks == KS_) {                                                                                                \
        hipLaunchKernelGGL((first_conv_mfma_kernel<TIN, NT_, KS_>), grid, block, lds, s, (const TIN*)x, (const __bf16*)wq, scale, \
                           bias, (__bf16*)y, d, in_a, in_b, o);                                                                  \
        return check_launch("first_conv_mfma_kernel");                                                                           \
    }
    PASN_FCM(2, 11) PASN_FCM(1, 11) PASN_FCM(2, 4) PASN_FCM(1, 4)  // 3 x 7 rows (the 7x7 stems), 1 x 7 rows (grey)
    PASN_FCM(2, 5) PASN_FCM(1, 5) PASN_FCM(2, 2) PASN_FCM(1, 2)    // 3 x 3 rows (X3D conv_xy when the stem is not fused), 1 x 3 rows
#undef PASN_FCM
    set_error("first_conv_mfma: no such instance");
    return PASN_ERR_UNSUPPORTED;
}

int launch_first_conv_mfma(const void* x, const void* wq, const float* scale, const float* bias, void* y, const pasn_conv_desc& d,
                           int in_dtype, float in_a, float in_b, int o, hipStream_t s) {
    if (in_dtype == PASN_F32) return launch_fc_mfma_t<float>(x, wq, scale, bias, y, d, in_a, in_b, o, s);
    if (in_dtype == PASN_BF16) return launch_fc_mfma_t<__bf16>(x, wq, scale, bias, y, d, in_a, in_b, o, s);
    if (in_dtype == PASN_U8) return launch_fc_mfma_t<unsigned char>(x, wq, scale, bias, y, d, in_a, in_b, o, s);
    set_error("first_conv_mfma: unknown input dtype");
    return PASN_ERR_ARG;
}

}  // namespace pasn

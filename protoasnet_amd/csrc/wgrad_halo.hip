// Weight gradient of the stride-1 "same" windowed convs (bf16): (1,3,3) and (3,1,1) of R(2+1)D-18, 3x3 of ResNet-18
// (training path of resnet_features.py:49-66,307-327; `loss.backward()` in Video_XProtoNet_e2e.py:118-141).
//
//   dW[co][ci][tap] = sum_rows dy[row][co] * x[row + off(tap)][ci]        (zero where the tap leaves the image / the clip)
//
// pw_wgrad_bf16_kernel ran these as one pointwise gradient per tap (blockIdx.z): every tap re-read dy and x, every y-block staged all
// channels, every staged row paid three integer divisions for the window map, and the split-K atomics landed 36 bytes apart (tap is the
// fast index of dW).  R(2+1)D-18 at 8 x 32 x 112 x 112: 29.1 of the 41.3 ms training step, 139 GB/s; the 64 -> 144 (1,3,3) layer 2.5 ms
// (its forward: 0.2 ms).  Here:
//   * a block of 8 waves owns up to 3 output-channel tiles x 2 input-channel tiles x ALL taps (<= 54 32x32 tiles: a wave holds <= 3
//     (ci tile, tap) pairs x 3 co tiles in accumulators) and walks its row partition 128 rows at a time;
//   * per step dy is staged once (8 x 8 register transpose -> At[co][128 rows], as in the pointwise kernel) and x THREE times, each copy a
//     different row shift: MODE 0 (1,3,3): copy e holds x[row + e - 1] over the step's rows plus a halo of W rows either side, so tap (b, e)
//     is copy e read at the (even) offset (b - 1) W -- a 4-byte-aligned fragment read; rows whose column wraps over the image border are
//     zeroed while the copy is staged (copy 0: source column W - 1, copy 2: source column 0); the rows whose tap leaves the image
//     vertically (a few octets per frame) are masked in the fragment.  MODE 1 (3,1,1): copy a holds x[row + (a - 1) H W], tap a reads it
//     unshifted; frames that wrap over the clip are zeroed at staging;
//   * no atomics: a block stores its tiles to partial[partition][tap][co][ci] (contiguous 128-byte runs) and conv_wgrad_reduce_kernel sums
//     the partitions in index order into dW[co][ci][tap] -- deterministic, and the 36-byte scatter happens once on 83 k values instead of
//     once per partition.
#include "common.h"

namespace pasn {

constexpr int WH_KT = 128;

__device__ __forceinline__ void wh_transpose8x8(const uint4 (&in)[8], uint4 (&out)[8]) {
    const unsigned* I = reinterpret_cast<const unsigned*>(in);
    unsigned* O = reinterpret_cast<unsigned*>(out);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned lo = I[(2 * p) * 4 + q], hi = I[(2 * p + 1) * 4 + q];
            O[(2 * q) * 4 + p] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
            O[(2 * q + 1) * 4 + p] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);
        }
}

struct WhGeom {
    int mode;        // 0: (1,3,3) spatial taps, 1: (3,1,1) temporal taps
    int taps;        // 9 | 3
    int HAL;         // mode 0: halo rows either side of a copy (multiple of 64, >= W); mode 1: 0
    int L;           // rows per copy (WH_KT + 2 HAL)
    int pitchA, pitchB;
    int co_tiles, ci_tiles, co_groups, ci_groups;
    int rows_per_block, parts;
    int Cout_r, Cin_r;  // padded extents of the partial buffer
};

template <int COT, int PW>
__global__ __launch_bounds__(512) void conv_wgrad_halo_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ dy, float* __restrict__ partial,
                                                              pasn_conv_desc d, WhGeom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const At = lds;                                            // [COT*32][pitchA]
    unsigned char* const Bt = lds + (size_t)COT * 32 * g.pitchA;              // [3][64][pitchB]
    unsigned char* const mb = Bt + (size_t)3 * 64 * g.pitchB;                 // [2][WH_KT/8] row-validity bytes (mode 0)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    const int W = d.Wi, H = d.Hi, T = d.Ti, FR = d.Hi * d.Wi;
    const long r0 = (long)blockIdx.x * g.rows_per_block, r1 = min(M, r0 + g.rows_per_block);
    const int co0 = blockIdx.y * COT * 32, ci0 = blockIdx.z * 64;
    const int cit = min(2, g.ci_tiles - (int)blockIdx.z * 2);  // ci tiles of this block
    const int P = cit * g.taps;                                 // (ci tile, tap) pairs of this block

    // ---- this wave's pairs: p = wave + 8 i -> copy, LDS shift, row-mask selector ----
    int p_ci[PW], p_tap[PW], p_copy[PW], p_shift[PW], p_msel[PW];
    bool p_ok[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int p = wave + 8 * i;
        p_ok[i] = p < P;
        const int pc = p_ok[i] ? p : 0;
        p_ci[i] = pc / g.taps;
        p_tap[i] = pc % g.taps;
        if (g.mode == 0) {
            const int b = p_tap[i] / 3, e = p_tap[i] % 3;
            p_copy[i] = e;
            p_shift[i] = g.HAL + (b - 1) * W;
            p_msel[i] = b == 0 ? 0 : (b == 2 ? 1 : -1);
        } else {
            p_copy[i] = p_tap[i];
            p_shift[i] = 0;
            p_msel[i] = -1;
        }
    }
    f32x16 acc[PW][COT];
#pragma unroll
    for (int i = 0; i < PW; ++i)
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.0f;

    // ---- staging roles (the same every step): units = dy patches (octet, channel group), then the three x copies (copy, octet, group);
    // a thread owns units tid and tid + 512 ----
    const int dy_cgs = COT * 4, dy_units = (WH_KT / 8) * dy_cgs;
    const int x_oct = g.L / 8, x_units = 3 * x_oct * 8;
    const int units = dy_units + x_units;  // <= 1024 (host-checked)
    int u_dst[2];                 // LDS byte offset (an int: a pointer kept in a register array loses its address space and every
                                  // access through it becomes a flat_load / flat_store), -1 = no unit
    int u_pitch[2], u_cp[2];      // u_cp: -1 = dy patch, 0..2 = x copy
    int u_row[2];                 // source row of the patch's first row in the CURRENT step (32-bit: the host bounds rows * channels by 2^31)
    int u_ch[2];                  // first channel of the patch, -1 = zeros (channel group past the tensor / no unit)
    int u_rowlen[2];
    int u_coord[2], u_within[2];  // x copies: column (mode 0) / frame and in-frame position (mode 1) of that source row, kept incrementally
    const int Mi = (int)M, r0i = (int)r0, r1i = (int)r1;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int u = tid + v * 512;
        u_dst[v] = -1;
        u_ch[v] = -1;
        u_pitch[v] = 0;
        u_cp[v] = -1;
        u_row[v] = r0i;
        u_rowlen[v] = 0;
        u_coord[v] = u_within[v] = 0;
        if (u < dy_units) {
            const int cg = u % dy_cgs, oct = u / dy_cgs;
            const int ch = co0 + cg * 8;
            u_dst[v] = (cg * 8) * g.pitchA + oct * 16;
            u_pitch[v] = g.pitchA;
            u_row[v] = r0i + oct * 8;
            u_rowlen[v] = d.Cout_p;
            if (ch < d.Cout_p) u_ch[v] = ch;
        } else if (u < units) {
            const int q = u - dy_units;
            const int cg = q & 7, oct = (q >> 3) % x_oct, cp = (q >> 3) / x_oct;
            const int ch = ci0 + cg * 8;
            u_dst[v] = COT * 32 * g.pitchA + (cp * 64 + cg * 8) * g.pitchB + oct * 16;
            u_pitch[v] = g.pitchB;
            u_cp[v] = cp;
            u_row[v] = r0i + oct * 8 + (g.mode == 0 ? -g.HAL + (cp - 1) : (cp - 1) * FR);
            u_rowlen[v] = d.Cin_p;
            if (ch < d.Cin_p) u_ch[v] = ch;
            if (g.mode == 0) {
                u_coord[v] = ((u_row[v] % W) + W) % W;  // rows before the tensor are masked by their sign; the column arithmetic is modular
            } else {
                const int sp = u_row[v] + FR * T;        // one clip period up: non-negative, same frame index modulo T
                u_within[v] = sp % FR;
                u_coord[v] = (sp / FR) % T;
            }
        }
    }
    const int step_w = WH_KT % W, step_fr = WH_KT % FR, step_t = (WH_KT / FR) % T;  // coordinate advance per 128-row step

    for (long rb = r0; rb < r1; rb += WH_KT) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {  // one unit at a time: both in flight (64 staging registers) spilled the 9-tile instance
            uint4 pre[8];
            unsigned okbits = 0;
            const bool chok = u_ch[v] >= 0;
            const __bf16* src = (u_cp[v] < 0 ? dy : x) + (chok ? u_ch[v] : 0);
            const int s0 = u_row[v];
            if (u_cp[v] < 0) {  // dy rows of this partition
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool ok = chok && s0 + i < r1i;
                    pre[i] = *reinterpret_cast<const uint4*>(src + (ok ? s0 + i : r0i) * u_rowlen[v]);
                    okbits |= (ok ? 1u : 0u) << i;
                }
            } else {
                // x rows of copy cp; a row is zero when its column (mode 0) / frame (mode 1) is the one that only a wrapped tap would read
                const int period = g.mode == 0 ? W : T;
                const int bad = u_cp[v] == 0 ? period - 1 : (u_cp[v] == 2 ? 0 : -1);
                int coord = u_coord[v], within = u_within[v];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int sr = s0 + i;
                    const bool ok = chok && sr >= 0 && sr < Mi && coord != bad;
                    pre[i] = *reinterpret_cast<const uint4*>(src + (ok ? sr : 0) * u_rowlen[v]);
                    okbits |= (ok ? 1u : 0u) << i;
                    if (g.mode == 0) {
                        coord = coord + 1 == W ? 0 : coord + 1;
                    } else if (++within == FR) {
                        within = 0;
                        coord = coord + 1 == T ? 0 : coord + 1;
                    }
                }
                // advance to the next step
                if (g.mode == 0) {
                    u_coord[v] += step_w;
                    if (u_coord[v] >= W) u_coord[v] -= W;
                } else {
                    u_within[v] += step_fr;
                    int carry = step_t;
                    if (u_within[v] >= FR) {
                        u_within[v] -= FR;
                        ++carry;
                    }
                    u_coord[v] += carry;
                    while (u_coord[v] >= T) u_coord[v] -= T;
                }
            }
            u_row[v] += WH_KT;
            if (u_dst[v] >= 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (!((okbits >> i) & 1u)) pre[i] = make_uint4(0, 0, 0, 0);
                uint4 out[8];
                wh_transpose8x8(pre, out);
#pragma unroll
                for (int c = 0; c < 8; ++c) *reinterpret_cast<uint4*>(lds + u_dst[v] + c * u_pitch[v]) = out[c];
            }
        }
        if (g.mode == 0 && tid < WH_KT / 8) {  // rows whose tap row b = 0 / b = 2 leaves the image: bit i of byte [sel][octet] = row valid
            unsigned m0 = 0, m2 = 0;
            const int rr = (int)rb + tid * 8;
            int hh = (rr / W) % H, ww = rr % W;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                m0 |= (hh != 0 ? 1u : 0u) << i;
                m2 |= (hh != H - 1 ? 1u : 0u) << i;
                if (++ww == W) {
                    ww = 0;
                    hh = hh + 1 == H ? 0 : hh + 1;
                }
            }
            mb[tid] = (unsigned char)m0;
            mb[WH_KT / 8 + tid] = (unsigned char)m2;
        }
        __syncthreads();
#pragma unroll 2
        for (int kk = 0; kk < WH_KT / 16; ++kk) {
            bf16x8 a[COT];
#pragma unroll
            for (int c = 0; c < COT; ++c) a[c] = *reinterpret_cast<const bf16x8*>(At + (size_t)(c * 32 + m) * g.pitchA + (kk * 2 + h) * 16);
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                if (p_ok[i]) {  // wave-uniform
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const unsigned* p32 = reinterpret_cast<const unsigned*>(Bt + ((size_t)p_copy[i] * 64 + p_ci[i] * 32 + m) * g.pitchB +
                                                                            (size_t)(kk * 16 + 8 * h + p_shift[i]) * 2);  // 4-byte aligned
                    u32x4 v;
                    v.x = p32[0];
                    v.y = p32[1];
                    v.z = p32[2];
                    v.w = p32[3];
                    if (p_msel[i] >= 0) {
                        const unsigned mk = mb[p_msel[i] * (WH_KT / 8) + kk * 2 + h];
                        if (__builtin_amdgcn_ballot_w64(mk != 0xffu)) {  // rare: an octet with a row in the first / last image row
                            v.x &= ((mk & 1u) ? 0x0000ffffu : 0u) | ((mk & 2u) ? 0xffff0000u : 0u);
                            v.y &= ((mk & 4u) ? 0x0000ffffu : 0u) | ((mk & 8u) ? 0xffff0000u : 0u);
                            v.z &= ((mk & 16u) ? 0x0000ffffu : 0u) | ((mk & 32u) ? 0xffff0000u : 0u);
                            v.w &= ((mk & 64u) ? 0x0000ffffu : 0u) | ((mk & 128u) ? 0xffff0000u : 0u);
                        }
                    }
                    const bf16x8 b = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                    for (int c = 0; c < COT; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c], b, acc[i][c], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // ---- partial[part][tap][co][ci]: rows = co (accumulator row), lane = ci ----
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        if (!p_ok[i]) continue;
        const int ci = ci0 + p_ci[i] * 32 + m;
        float* base = partial + (((size_t)blockIdx.x * g.taps + p_tap[i]) * g.Cout_r) * g.Cin_r + ci;
#pragma unroll
        for (int c = 0; c < COT; ++c) {
            if ((blockIdx.y * COT + c) < g.co_tiles) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int co = co0 + c * 32 + acc_row(reg, h);
                    base[(size_t)co * g.Cin_r] = acc[i][c][reg];
                }
            }
        }
    }
}

// dW[co][ci][tap] += sum over partitions (index order) of partial[part][tap][co][ci]
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int parts, int taps, int Cout,
                                                                int Cin, int Cout_r, int Cin_r) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // (tap, co, ci), ci fastest: coalesced reads
    const long total = (long)taps * Cout * Cin;
    if (idx >= total) return;
    const int ci = (int)(idx % Cin), co = (int)((idx / Cin) % Cout), tap = (int)(idx / ((long)Cin * Cout));
    const size_t stride = (size_t)taps * Cout_r * Cin_r;
    const float* p = partial + ((size_t)tap * Cout_r + co) * Cin_r + ci;
    float s = 0.0f;
    int q = 0;
    for (; q + 8 <= parts; q += 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = p[(size_t)(q + k) * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += t[k];
    }
    for (; q < parts; ++q) s += p[(size_t)q * stride];
    dw[((size_t)co * Cin + ci) * taps + tap] += s;
}

// ---- the other windowed / strided convs (bf16): (1,3,3) stride (1,2,2), (3,1,1) stride (2,1,1), strided 1x1x1 shortcuts ----------------
// One tap per blockIdx.z, as a pointwise gradient whose x rows are GATHERED through the window map -- but tile-local (a block owns a 2 x 2
// group of 32 x 32 tiles and stages only their 64 + 64 channels, 128 output rows per step, one 8 x 8 patch per thread), with the window
// map advanced incrementally (three integer divisions per thread and step instead of three per row), and into the partial buffer
// (deterministic) instead of atomics 36 bytes apart.  pw_wgrad_bf16_kernel took 1.08 ms for the 64 -> 230 (1,3,3) stride-2 layer.
__global__ __launch_bounds__(256) void conv_wgrad_gather_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ dy, float* __restrict__ partial,
                                                                pasn_conv_desc d, int ci_pairs, int rows_per_block, int Cout_r, int Cin_r) {
    constexpr int PITCH = WH_KT * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const At = lds;                         // [64 co channels][PITCH]
    unsigned char* const Bt = lds + (size_t)64 * PITCH;    // [64 ci channels][PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int R = d.N * d.To * d.Ho * d.Wo;  // output rows (host: rows * channels < 2^31)
    const int r0 = blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
    const int cop = blockIdx.y / ci_pairs, cip = blockIdx.y % ci_pairs;
    const int co0 = cop * 64, ci0 = cip * 64;
    const int taps = d.kt * d.kh * d.kw, tap = blockIdx.z;
    const int ta = tap / (d.kh * d.kw), tb = (tap / d.kw) % d.kh, te = tap % d.kw;
    // staging role: one 8-row x 8-channel patch per thread and step -- 16 row octets x (8 co + 8 ci channel groups)
    const int gch = tid & 15, r8 = tid >> 4;
    const bool is_a = gch < 8;
    const int cg = is_a ? gch : gch - 8;
    const int cp = is_a ? d.Cout_p : d.Cin_p;
    const int ch = (is_a ? co0 : ci0) + cg * 8;
    const bool ch_ok = ch < cp;
    const __bf16* src = (is_a ? dy : x) + (ch_ok ? ch : 0);
    const int dst = (is_a ? 0 : 64 * PITCH) + (cg * 8) * PITCH + ((r8 + cg) & (WH_KT / 8 - 1)) * 16;  // slot rotation: see wgrad.hip (wg_slot)
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const int tco = wave >> 1, tci = wave & 1;
    for (int rb = r0; rb < r1; rb += WH_KT) {
        uint4 pre[8];
        unsigned okbits = 0;
        const int rr = rb + r8 * 8;
        if (is_a) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool ok = ch_ok && rr + i < r1;
                pre[i] = *reinterpret_cast<const uint4*>(src + (ok ? rr + i : r0) * cp);
                okbits |= (ok ? 1u : 0u) << i;
            }
        } else {
            // output position of the patch's first row, then one position per row
            const int rc = min(rr, R - 1);
            int wo = rc % d.Wo, q = rc / d.Wo;
            int ho = q % d.Ho;
            q /= d.Ho;
            int to = q % d.To, n = q / d.To;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ti = to * d.st - d.pt + ta, hi = ho * d.sh - d.ph + tb, wi = wo * d.sw - d.pw + te;
                const bool ok = ch_ok && rr + i < r1 && (unsigned)ti < (unsigned)d.Ti && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi;
                const int row = ((n * d.Ti + ti) * d.Hi + hi) * d.Wi + wi;
                pre[i] = *reinterpret_cast<const uint4*>(src + (ok ? row : 0) * cp);
                okbits |= (ok ? 1u : 0u) << i;
                if (++wo == d.Wo) {
                    wo = 0;
                    if (++ho == d.Ho) {
                        ho = 0;
                        if (++to == d.To) {
                            to = 0;
                            ++n;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (!((okbits >> i) & 1u)) pre[i] = make_uint4(0, 0, 0, 0);
        uint4 out[8];
        wh_transpose8x8(pre, out);
#pragma unroll
        for (int c = 0; c < 8; ++c) *reinterpret_cast<uint4*>(lds + dst + c * PITCH) = out[c];
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < WH_KT / 16; ++kk) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(At + (size_t)(tco * 32 + m) * PITCH + ((kk * 2 + h + tco * 4 + (m >> 3)) & (WH_KT / 8 - 1)) * 16);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bt + (size_t)(tci * 32 + m) * PITCH + ((kk * 2 + h + tci * 4 + (m >> 3)) & (WH_KT / 8 - 1)) * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // partial[part][tap][co][ci]
    const int ci = ci0 + tci * 32 + m;
    if (ci < Cin_r && co0 + tco * 32 < Cout_r) {
        float* base = partial + (((size_t)blockIdx.x * taps + tap) * Cout_r) * Cin_r + ci;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) base[(size_t)(co0 + tco * 32 + acc_row(reg, h)) * Cin_r] = acc[reg];
    }
}

struct WgGather {
    int taps, co_pairs, ci_pairs, rows_per_block, parts, Cout_r, Cin_r;
};

static bool wgrad_gather_geom(const pasn_conv_desc& d, int dtype, WgGather& g) {
    if (const char* e = tune("PASN_NO_WGRAD_GATHER"))
        if (e[0] == '1') return false;
    if (dtype != PASN_BF16 || d.Cin_p % 8 || d.Cout_p % 8) return false;
    g.taps = d.kt * d.kh * d.kw;
    const bool strided = d.st != 1 || d.sh != 1 || d.sw != 1;
    const bool det = tune("PASN_WGRAD_DET") ? atoi(tune("PASN_WGRAD_DET")) != 0 : false;
    if (g.taps == 1 && !strided && !det) return false;  // plain pointwise layers keep their (atomic) kernels unless asked
    if (g.taps > 27) return false;
    const long R = (long)d.N * d.To * d.Ho * d.Wo, Rin = (long)d.N * d.Ti * d.Hi * d.Wi;
    if (R * d.Cout_p >= (1L << 31) || Rin * d.Cin_p >= (1L << 31)) return false;
    const int co_tiles = ceil_div(d.Cout_p, 32), ci_tiles = ceil_div(d.Cin_p, 32);
    g.co_pairs = ceil_div(co_tiles, 2);
    g.ci_pairs = ceil_div(ci_tiles, 2);
    g.Cout_r = g.co_pairs * 64;
    g.Cin_r = g.ci_pairs * 64;
    const long gy = (long)g.co_pairs * g.ci_pairs * g.taps;
    // about 2048 blocks, at least two 128-row steps each, at most 48 MB of partials
    const long per_part = (long)g.taps * g.Cout_r * g.Cin_r * 4;
    long parts = std::max<long>(1, std::min<long>(std::min<long>((48L << 20) / per_part, 2048 / gy + 1), R / (2 * WH_KT)));
    const long rpb = (ceil_div(R, parts) + WH_KT - 1) / WH_KT * WH_KT;
    g.rows_per_block = (int)rpb;
    g.parts = (int)ceil_div(R, rpb);
    return true;
}

bool wgrad_halo_geom(const pasn_conv_desc& d, int dtype, WhGeom& g) {
    if (const char* e = tune("PASN_NO_WGRAD_HALO"))
        if (e[0] == '1') return false;
    if (dtype != PASN_BF16) return false;
    if (d.st != 1 || d.sh != 1 || d.sw != 1 || d.To != d.Ti || d.Ho != d.Hi || d.Wo != d.Wi) return false;
    if (d.kt == 1 && d.pt == 0 && d.kh == 3 && d.kw == 3 && d.ph == 1 && d.pw == 1 && d.Wi % 2 == 0) g.mode = 0;
    else if (d.kt == 3 && d.pt == 1 && d.kh == 1 && d.kw == 1 && d.ph == 0 && d.pw == 0) g.mode = 1;
    else return false;
    if (d.Cin_p % 8 || d.Cout_p % 8) return false;
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    if (M * d.Cin_p >= (1L << 31) || M * d.Cout_p >= (1L << 31)) return false;
    g.taps = g.mode == 0 ? 9 : 3;
    g.HAL = g.mode == 0 ? (d.Wi + 63) / 64 * 64 : 0;  // L a multiple of 128: copy pitch = 4 dwords mod 64 banks
    g.L = WH_KT + 2 * g.HAL;
    g.pitchA = WH_KT * 2 + 16;
    g.pitchB = g.L * 2 + 16;
    g.co_tiles = ceil_div(d.Cout_p, 32);
    g.ci_tiles = ceil_div(d.Cin_p, 32);
    g.Cout_r = g.co_tiles * 32;
    g.Cin_r = g.ci_tiles * 32;
    return true;
}

static int wh_cot(const WhGeom& g) { return g.co_tiles >= 3 ? 3 : g.co_tiles; }
static size_t wh_lds(const WhGeom& g, int cot) { return (size_t)cot * 32 * g.pitchA + (size_t)3 * 64 * g.pitchB + 2 * (WH_KT / 8); }

static void wh_partition(const pasn_conv_desc& d, WhGeom& g) {
    const int cot = wh_cot(g);
    g.co_groups = ceil_div(g.co_tiles, cot);
    g.ci_groups = ceil_div(g.ci_tiles, 2);
    const long M = (long)d.N * d.To * d.Ho * d.Wo;
    // one block per CU (LDS); about two rounds of blocks, at least two 128-row steps each, at most 256 partitions (partial buffer)
    long parts = std::max<long>(1, std::min<long>(std::min<long>(256, 512 / ((long)g.co_groups * g.ci_groups) + 1), M / (2 * WH_KT)));
    long rpb = (ceil_div(M, parts) + WH_KT - 1) / WH_KT * WH_KT;
    g.rows_per_block = (int)rpb;
    g.parts = (int)ceil_div(M, rpb);
}

size_t wgrad_halo_workspace_bytes(const pasn_conv_desc& d, int dtype) {
    WhGeom g{};
    if (!wgrad_halo_geom(d, dtype, g)) {
        WgGather q{};
        return wgrad_gather_geom(d, dtype, q) ? (size_t)q.parts * q.taps * q.Cout_r * q.Cin_r * sizeof(float) : 0;
    }
    if (wh_lds(g, wh_cot(g)) > 160 * 1024 || (WH_KT / 8) * wh_cot(g) * 4 + 3 * (g.L / 8) * 8 > 1024) return 0;
    wh_partition(d, g);
    return (size_t)g.parts * g.taps * g.Cout_r * g.Cin_r * sizeof(float);
}

bool wgrad_halo(const void* x, const void* dy, float* dw, void* ws, const pasn_conv_desc& d, int dtype, hipStream_t s) {
    WhGeom g{};
    if (!ws) return false;
    if (!wgrad_halo_geom(d, dtype, g)) {
        WgGather q{};
        if (!wgrad_gather_geom(d, dtype, q)) return false;
        const dim3 grid(q.parts, q.co_pairs * q.ci_pairs, q.taps);
        hipLaunchKernelGGL(conv_wgrad_gather_kernel, grid, dim3(256), (size_t)128 * (WH_KT * 2 + 16), s, (const __bf16*)x, (const __bf16*)dy, (float*)ws, d,
                           q.ci_pairs, q.rows_per_block, q.Cout_r, q.Cin_r);
        const long total = (long)q.taps * d.Cout * d.Cin;
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)ceil_div(total, 256L)), dim3(256), 0, s, (const float*)ws, dw, q.parts, q.taps, d.Cout,
                           d.Cin, q.Cout_r, q.Cin_r);
        return true;
    }
    const int cot = wh_cot(g);
    const size_t lds = wh_lds(g, cot);
    if (lds > 160 * 1024 || (WH_KT / 8) * cot * 4 + 3 * (g.L / 8) * 8 > 1024) return false;
    wh_partition(d, g);
    const int pairs = std::min(2, g.ci_tiles) * g.taps;
    const int pw = ceil_div(pairs, 8);
    const dim3 grid(g.parts, g.co_groups, g.ci_groups), block(512);
#define PASN_WH(COT_, PW_)                                                                                                        \
    if (cot == COT_ && pw == PW_) {                                                                                               \
        PASN_MAX_LDS(160 * 1024, conv_wgrad_halo_kernel<COT_, PW_>);                                                              \
        hipLaunchKernelGGL((conv_wgrad_halo_kernel<COT_, PW_>), grid, block, lds, s, (const __bf16*)x, (const __bf16*)dy, (float*)ws, d, g); \
    }
    PASN_WH(1, 1) PASN_WH(2, 1) PASN_WH(3, 1) PASN_WH(1, 2) PASN_WH(2, 2) PASN_WH(3, 2) PASN_WH(1, 3) PASN_WH(2, 3) PASN_WH(3, 3)
#undef PASN_WH
    const long total = (long)g.taps * d.Cout * d.Cin;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)ceil_div(total, 256L)), dim3(256), 0, s, (const float*)ws, dw, g.parts, g.taps, d.Cout,
                       d.Cin, g.Cout_r, g.Cin_r);
    return true;
}

}  // namespace pasn

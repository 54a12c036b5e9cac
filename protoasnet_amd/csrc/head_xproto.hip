// Head B (XProtoNet / Video_XProtoNet, "ProtoASNet"): add-on + occurrence module + occurrence-weighted
// pooling + cosine similarity + last layer.
//
// Round-1 structure ("Head B"): the five 1x1(x1) convs run on the MFMA implicit-GEMM kernel
// (conv.hip) over channels-last [N*S][C] rows; the pooling  F[n][p][d] = sum_s occ[n][s][p] * f[n][s][d]
// (the reference materialises the (N,P,D,S) product, Video_XProtoNet.py:87) is a split-S reduction into
// fixed-order slabs so results are bitwise reproducible; a finishing kernel sums the slabs and does
// cosine / (s+1)/2 / last layer.  Nothing of size N*P*D*S is ever formed.
#include "common.h"

namespace pasn {

constexpr int POOL_ST = 32;  // positions staged per step

// grid (dchunks*pchunks, G, N), 256 threads: thread = one feature column d, PC prototype accumulators.
template <typename T, int PC>
__global__ __launch_bounds__(256) void xproto_pool_kernel(const T* __restrict__ occ_cl, const T* __restrict__ f,
                                                          float* __restrict__ occ_planar, float* __restrict__ ws, int S,
                                                          int P, int Pp, int D, int Dp, int G, int dchunks, int do_pool) {
    __shared__ __attribute__((aligned(16))) float tile[POOL_ST * PC];
    const int dchunk = blockIdx.x % dchunks, pchunk = blockIdx.x / dchunks;
    const int g = blockIdx.y, n = blockIdx.z;
    const int SG = (S + G - 1) / G;
    const int sbeg = g * SG;
    const int send = min(S, sbeg + SG);
    const int d = dchunk * 256 + threadIdx.x;
    const bool dvalid = do_pool && d < D;
    const int pbase = pchunk * PC;
    float acc[PC];
#pragma unroll
    for (int i = 0; i < PC; ++i) acc[i] = 0.0f;

    for (int s0 = sbeg; s0 < send; s0 += POOL_ST) {
        for (int i = threadIdx.x; i < POOL_ST * PC; i += 256) {
            const int sl = i / PC, pl = i % PC;
            const int s = s0 + sl, p = pbase + pl;
            tile[i] = (s < send && p < P) ? (float)occ_cl[((long)n * S + s) * Pp + p] : 0.0f;
        }
        __syncthreads();
        if (occ_planar && dchunk == 0) {
            for (int i = threadIdx.x; i < POOL_ST * PC; i += 256) {
                const int pl = i / POOL_ST, sl = i % POOL_ST;
                const int s = s0 + sl, p = pbase + pl;
                if (s < send && p < P) occ_planar[((long)n * P + p) * S + s] = tile[sl * PC + pl];
            }
        }
        if (dvalid) {
            const int cnt = min(POOL_ST, send - s0);
            for (int sl = 0; sl < cnt; ++sl) {
                const float fv = (float)f[((long)n * S + s0 + sl) * Dp + d];
#pragma unroll
                for (int i = 0; i < PC; i += 4) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(tile + sl * PC + i);
                    acc[i + 0] = fmaf(o[0], fv, acc[i + 0]);
                    acc[i + 1] = fmaf(o[1], fv, acc[i + 1]);
                    acc[i + 2] = fmaf(o[2], fv, acc[i + 2]);
                    acc[i + 3] = fmaf(o[3], fv, acc[i + 3]);
                }
            }
        }
        __syncthreads();
    }
    if (dvalid) {
#pragma unroll
        for (int i = 0; i < PC; ++i) {
            const int p = pbase + i;
            if (p < P) ws[(((long)n * G + g) * P + p) * D + d] = acc[i];
        }
    }
}

// Occurrence-weighted pooling on the matrix cores: F[p][d] = sum_s occ[s][p] * f[s][d]  (the reference forms the (N,P,D,S) product,
// Video_XProtoNet.py:87-88).  The contraction index s is the SLOW index of both channels-last operands, which rules out the bf16
// MFMA's 8-consecutive-k fragments without a transpose; the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) takes ONE element per lane and
// operand -- lane (r, k) supplies A[p = r][s0 + k] and B[s0 + k][d = r] -- so both operands are plain loads of 32 consecutive channels
// of a row, converted to fp32 on the way (exact: fp32 products of bf16 values, fp32 accumulation in s order).  At fp32-MFMA rate the
// whole pooling of 32 clips x 1568 positions x 64 x 256 is ~10 us of matrix time (round 1's VALU kernel: 41-61 us).
// grid (G splits of S, groups of 128 feature columns, N), one wave per block; two prototype tiles x four column tiles in accumulators.
// The wave of column group 0 also writes the planar fp32 occurrence map (N,P,S) the callers get, through a [32 p][64 s] LDS tile.
template <typename T>
__global__ __launch_bounds__(64) void xproto_pool_mfma_kernel(const T* __restrict__ occ_cl, const T* __restrict__ f,
                                                              float* __restrict__ occ_planar, float* __restrict__ ws, int S, int P, int Pp,
                                                              int D, int Dp, int G, int pbase, int do_pool, int abs_in) {
    __shared__ float tile[2][32][65];
    const int lane = threadIdx.x, c = lane & 31, kq = lane >> 5;
    const int g = blockIdx.x, dg = blockIdx.y, n = blockIdx.z;
    const int SG = ((S + G - 1) / G + 63) / 64 * 64;  // whole 64-position tiles per split
    const int sbeg = g * SG, send = min(S, sbeg + SG);
    const int d0 = dg * 128;
    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    const bool want_map = occ_planar != nullptr && dg == 0;
    const int pcol[2] = {min(pbase + c, Pp - 1), min(pbase + 32 + c, Pp - 1)};
    const bool pok[2] = {pbase + c < P, pbase + 32 + c < P};
    int dcol[4];
    bool dok[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        dcol[b] = min(d0 + b * 32 + c, Dp - 1);
        dok[b] = do_pool && (d0 + b * 32 + c) < D;
    }
    // groups of 4 k-steps (8 positions): the 24 loads of group q + 1 are issued BEFORE the 32 MFMAs of group q (register double buffer;
    // with the loads behind the MFMAs every group paid a full memory round trip: 32 us per launch instead of ~16)
    auto load_group = [&](int sq, float (&av)[4][2], float (&bv)[4][4]) {  // raw, unconditional, clamped
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long row = (long)n * S + min(sq + 2 * u + kq, S - 1);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float o = (float)occ_cl[row * Pp + pcol[a]];
                av[u][a] = abs_in ? fabsf(o) : o;  // training tail: the occurrence module's output before its |.|
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[u][b] = (float)f[row * Dp + dcol[b]];
        }
    };
    auto use_group = [&](int sq, int s0, float (&av)[4][2], float (&bv)[4][4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = sq + 2 * u + kq;
            const bool sv = s < send;
#pragma unroll
            for (int a = 0; a < 2; ++a) av[u][a] = (sv && pok[a]) ? av[u][a] : 0.0f;
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[u][b] = (sv && dok[b]) ? bv[u][b] : 0.0f;
            if (want_map) {  // wave-uniform
                tile[0][c][s - s0] = av[u][0];
                tile[1][c][s - s0] = av[u][1];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
        }
    };
    auto flush_map = [&](int s0) {  // one wave: LDS ops of a wave complete in order, no barrier needed
        const int s = s0 + lane;
#pragma unroll
        for (int a = 0; a < 2; ++a)
            for (int pr = 0; pr < 32; ++pr) {
                const int p = pbase + a * 32 + pr;
                if (p < P && s < send) occ_planar[((long)n * P + p) * S + s] = tile[a][pr][lane];
            }
    };
    if (sbeg < send) {
        float avA[4][2], bvA[4][4], avB[4][2], bvB[4][4];
        load_group(sbeg, avA, bvA);
#pragma unroll 1
        for (int sq = sbeg; sq < send; sq += 16) {  // two groups per iteration: buffers A and B alternate by name
            const int s0a = sbeg + (sq - sbeg) / 64 * 64, s0b = sbeg + (sq + 8 - sbeg) / 64 * 64;
            load_group(sq + 8, avB, bvB);
            use_group(sq, s0a, avA, bvA);
            load_group(sq + 16, avA, bvA);
            if (sq + 8 < send) use_group(sq + 8, s0b, avB, bvB);
            if (want_map && ((sq + 16 - sbeg) % 64 == 0 || sq + 16 >= send)) flush_map(s0a);
        }
    }
    if (do_pool) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int d = d0 + b * 32 + c;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int p = pbase + a * 32 + acc_row(i, kq);
                    if (p < P && d < D) ws[(((long)n * G + g) * P + p) * D + d] = acc[a][b][i];
                }
            }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// grid N, 1024 threads (16 waves); a wave per prototype, a lane per 4 consecutive feature dims (D <= 1024).
// Pure latency kernel (32 clips x 30 prototypes x G slabs of 256 floats): the slab sum is unrolled x8 so eight 16-byte
// loads are in flight per lane (same summation order g = 0, 1, ... as a rolled loop), and the pooled feature stays in
// registers for the norm and the dot product instead of being re-read from global memory.  (The first version walked
// 8 prototypes per wave with rolled scalar loads: 216 us per launch.)
// XF_MAXC = 4-float chunks per lane: D <= 256 (1024 threads) or D <= 1024 (256 threads: the registers of 4 chunks)
template <int XF_MAXC, int THREADS>
__global__ __launch_bounds__(THREADS) void xproto_finish_kernel(const float* __restrict__ ws, const float* __restrict__ protos,
                                                             const float* __restrict__ fc_w, float* __restrict__ feat,
                                                             float* __restrict__ sim, float* __restrict__ logits, int G, int P,
                                                             int D, int K) {
    extern __shared__ __attribute__((aligned(16))) float sims[];  // [P]
    const int n = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int p = wave; p < P; p += nwaves) {
        float* fp = feat + ((long)n * P + p) * D;
        const float* pp = protos + (long)p * D;
        const float* wp = ws + ((long)n * G * P + p) * D;  // slab g at + g * P * D
        f32x4 fv[XF_MAXC], pv[XF_MAXC];
        float ssf = 0.0f, ssp = 0.0f;
#pragma unroll
        for (int ci = 0; ci < XF_MAXC; ++ci) {
            const int d0 = ci * 256 + lane * 4;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            pv[ci] = v;
            if (d0 < D) {  // D is a multiple of 4 (checked on the host)
                int g = 0;
                for (; g + 8 <= G; g += 8) {
                    f32x4 t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(wp + (long)(g + u) * P * D + d0);
#pragma unroll
                    for (int u = 0; u < 8; ++u) v += t[u];
                }
                for (; g < G; ++g) v += *reinterpret_cast<const f32x4*>(wp + (long)g * P * D + d0);
                *reinterpret_cast<f32x4*>(fp + d0) = v;
                pv[ci] = *reinterpret_cast<const f32x4*>(pp + d0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ssf = fmaf(v[j], v[j], ssf);
                    ssp = fmaf(pv[ci][j], pv[ci][j], ssp);
                }
            }
            fv[ci] = v;
        }
        ssf = wave_sum(ssf);
        ssp = wave_sum(ssp);
        const float nf = fmaxf(sqrtf(ssf), 1e-8f), np = fmaxf(sqrtf(ssp), 1e-8f);
        float dot = 0.0f;
#pragma unroll
        for (int ci = 0; ci < XF_MAXC; ++ci)
#pragma unroll
            for (int j = 0; j < 4; ++j) dot = fmaf(fv[ci][j] / nf, pv[ci][j] / np, dot);
        dot = wave_sum(dot);
        if (lane == 0) {
            const float sv = (dot + 1.0f) / 2.0f;
            sims[p] = sv;
            sim[(long)n * P + p] = sv;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float s = 0.0f;
        for (int p = 0; p < P; ++p) s = fmaf(sims[p], fc_w[(long)k * P + p], s);
        logits[(long)n * K + k] = s;
    }
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct XpLayout {
    size_t f1, f, o2, occ, slabs, total;
};
static XpLayout xp_layout(const pasn_xproto_desc& d, int dtype, int G) {
    const size_t es = dtype == PASN_BF16 ? 2 : 4;
    const size_t rows = (size_t)d.N * d.S;
    XpLayout L;
    size_t off = 0;
    L.f1 = off;
    off += align256(rows * d.Dp * es);
    L.f = off;
    off += align256(rows * d.Dp * es);
    L.o2 = off;
    off += align256(rows * d.Hp * es);
    L.occ = off;
    off += align256(rows * d.Pp * es);
    L.slabs = off;
    off += align256((size_t)d.N * G * d.P * d.D * sizeof(float));
    L.total = off;
    return L;
}

static pasn_conv_desc pointwise_desc(const pasn_xproto_desc& d, int cin, int cin_p, int cout, int cout_p, int act, int dtype) {
    const int kstep = dtype == PASN_BF16 ? 16 : 8;
    pasn_conv_desc c = {};
    c.N = d.N; c.Ti = 1; c.Hi = 1; c.Wi = d.S;
    c.Cin = cin; c.Cin_p = cin_p;
    c.To = 1; c.Ho = 1; c.Wo = d.S;
    c.Cout = cout; c.Cout_p = cout_p;
    c.kt = c.kh = c.kw = 1;
    c.st = c.sh = c.sw = 1;
    c.act = act;
    c.w_kc = (cin_p + kstep - 1) / kstep * kstep;
    c.w_rows = (cout_p + 127) / 128 * 128;
    return c;
}

}  // namespace pasn

using namespace pasn;

static bool xp_desc_ok(const pasn_xproto_desc* d) {
    return d && d->N > 0 && d->S > 0 && d->Cb > 0 && d->D > 0 && d->Hd > 0 && d->P > 0 && d->K > 0 && d->Cbp >= d->Cb &&
           d->Dp >= d->D && d->Hp >= d->Hd && d->Pp >= d->P && d->Cbp % 8 == 0 && d->Dp % 8 == 0 && d->Hp % 8 == 0 &&
           d->Pp % 8 == 0;
}

static bool xp_pool_mfma() {
    const char* e = tune("PASN_POOL_VALU");
    return !(e && e[0] == '1');
}

// Pooling on the matrix cores + finish, for the training tail (head_train.hip): r = occurrence-module output BEFORE its |.| (abs applied
// on load), z = add-on output (NULL: occurrence map only), slabs = [N][G][P][D] floats.  Same kernels as the inference head.
namespace pasn {
int xproto_tail_splits(const pasn_xproto_desc& d) {
    const long base = (long)d.N * ceil_div(d.D, 128) * ceil_div(d.P, 64);
    long G = (2048 + base - 1) / base;
    const long gmax = ceil_div(d.S, 64);
    return (int)std::max<long>(1, std::min(G, gmax));
}
int xproto_tail_pool_finish(const void* z, const void* r, const float* protos, const float* fc_w, float* occ, float* feat, float* sim,
                            float* logits, float* slabs, const pasn_xproto_desc& d, int dtype, hipStream_t s) {
    const int G = xproto_tail_splits(d);
    const bool full = z != nullptr;
    const dim3 grid(G, full ? ceil_div(d.D, 128) : 1, d.N), block(64);
    // occurrence map only: the kernel still issues its (unused) feature loads -- point them at r itself with r's row width
    const int D_ = full ? d.D : 0, Dp_ = full ? d.Dp : d.Pp;
    for (int pbase = 0; pbase < d.P; pbase += 64) {
        if (dtype == PASN_F32)
            hipLaunchKernelGGL((xproto_pool_mfma_kernel<float>), grid, block, 0, s, (const float*)r, (const float*)(full ? z : r), occ, slabs, d.S, d.P,
                               d.Pp, D_, Dp_, G, pbase, full ? 1 : 0, 1);
        else
            hipLaunchKernelGGL((xproto_pool_mfma_kernel<__bf16>), grid, block, 0, s, (const __bf16*)r, (const __bf16*)(full ? z : r), occ, slabs, d.S,
                               d.P, d.Pp, D_, Dp_, G, pbase, full ? 1 : 0, 1);
    }
    if (full) {
        if (d.D <= 256)
            hipLaunchKernelGGL((xproto_finish_kernel<1, 1024>), dim3(d.N), dim3(1024), (size_t)d.P * sizeof(float), s, slabs, protos, fc_w, feat, sim,
                               logits, G, d.P, d.D, d.K);
        else
            hipLaunchKernelGGL((xproto_finish_kernel<4, 256>), dim3(d.N), dim3(256), (size_t)d.P * sizeof(float), s, slabs, protos, fc_w, feat, sim,
                               logits, G, d.P, d.D, d.K);
    }
    return check_launch("xproto_tail_pool_finish");
}
}  // namespace pasn

extern "C" int pasn_xproto_head_splits(const pasn_xproto_desc* d) {
    if (!xp_desc_ok(d)) return 0;
    const int pc = d->P <= 32 ? 32 : 64;
    const bool mfma = xp_pool_mfma();
    const long base = mfma ? (long)d->N * ceil_div(d->D, 128) * ceil_div(d->P, 64) : (long)d->N * ceil_div(d->D, 256) * ceil_div(d->P, pc);
    long G = ((mfma ? 2048 : 1024) + base - 1) / base;  // aim at >= 1024 blocks of 256 threads / 2048 single-wave blocks
    const long gmax = ceil_div(d->S, mfma ? 64 : POOL_ST);
    if (G > gmax) G = gmax;
    if (G < 1) G = 1;
    return (int)G;
}

extern "C" size_t pasn_xproto_head_workspace_bytes(const pasn_xproto_desc* d, int dtype) {
    if (!xp_desc_ok(d)) return 0;
    return xp_layout(*d, dtype, pasn_xproto_head_splits(d)).total;
}

extern "C" int pasn_xproto_head_fwd(const void* x, const void* a1, const float* a1b, const void* a2, const float* a2b,
                                    const void* o1, const float* o1b, const void* o2, const float* o2b, const void* o3,
                                    const float* protos, const float* fc_w, float* occ, float* feat, float* sim,
                                    float* logits, void* ws, const pasn_xproto_desc* d, int dtype, void* stream) {
    PASN_REQUIRE(xp_desc_ok(d), "bad descriptor (channel strides must be multiples of 8)");
    PASN_REQUIRE(dtype == PASN_F32 || dtype == PASN_BF16, "unknown dtype");
    PASN_REQUIRE(x && o1 && o1b && o2 && o2b && o3 && occ && ws, "null pointer");
    PASN_REQUIRE(((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
    const bool full = d->mode == 0;
    if (full) PASN_REQUIRE(a1 && a1b && a2 && a2b && protos && fc_w && feat && sim && logits, "null pointer (full mode)");
    const int G = pasn_xproto_head_splits(d);
    const XpLayout L = xp_layout(*d, dtype, G);
    char* base = (char*)ws;
    void* b_f1 = base + L.f1;
    void* b_f = base + L.f;
    void* b_o2 = base + L.o2;
    void* b_occ = base + L.occ;
    float* b_slabs = (float*)(base + L.slabs);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (full) {
        pasn_conv_desc c1 = pointwise_desc(*d, d->Cb, d->Cbp, d->D, d->Dp, PASN_ACT_RELU, dtype);
        if ((rc = pasn_conv3d_fwd(x, a1, nullptr, a1b, nullptr, nullptr, b_f1, &c1, dtype, stream))) return rc;
        pasn_conv_desc c2 = pointwise_desc(*d, d->D, d->Dp, d->D, d->Dp, PASN_ACT_NONE, dtype);
        if ((rc = pasn_conv3d_fwd(b_f1, a2, nullptr, a2b, nullptr, nullptr, b_f, &c2, dtype, stream))) return rc;
    }
    pasn_conv_desc c3 = pointwise_desc(*d, d->Cb, d->Cbp, d->D, d->Dp, PASN_ACT_RELU, dtype);
    if ((rc = pasn_conv3d_fwd(x, o1, nullptr, o1b, nullptr, nullptr, b_f1, &c3, dtype, stream))) return rc;
    pasn_conv_desc c4 = pointwise_desc(*d, d->D, d->Dp, d->Hd, d->Hp, PASN_ACT_RELU, dtype);
    if ((rc = pasn_conv3d_fwd(b_f1, o2, nullptr, o2b, nullptr, nullptr, b_o2, &c4, dtype, stream))) return rc;
    pasn_conv_desc c5 = pointwise_desc(*d, d->Hd, d->Hp, d->P, d->Pp, PASN_ACT_ABS, dtype);  // occurrence_module.4 has no bias
    if ((rc = pasn_conv3d_fwd(b_o2, o3, nullptr, nullptr, nullptr, nullptr, b_occ, &c5, dtype, stream))) return rc;

    if (xp_pool_mfma()) {
        // matrix-core pooling: one wave per (S split, 128 feature columns, clip); prototype chunks of 64 are separate launches (P <= 64: one)
        const dim3 grid(G, full ? ceil_div(d->D, 128) : 1, d->N), block(64);
        for (int pbase = 0; pbase < d->P; pbase += 64) {
            if (dtype == PASN_F32)
                hipLaunchKernelGGL((xproto_pool_mfma_kernel<float>), grid, block, 0, s, (const float*)b_occ, (const float*)b_f, occ, b_slabs,
                                   d->S, d->P, d->Pp, d->D, d->Dp, G, pbase, full ? 1 : 0, 0);
            else
                hipLaunchKernelGGL((xproto_pool_mfma_kernel<__bf16>), grid, block, 0, s, (const __bf16*)b_occ, (const __bf16*)b_f, occ, b_slabs,
                                   d->S, d->P, d->Pp, d->D, d->Dp, G, pbase, full ? 1 : 0, 0);
        }
    } else {
    const int pc = d->P <= 32 ? 32 : 64;
    const int dchunks = full ? ceil_div(d->D, 256) : 1;
    const int pchunks = ceil_div(d->P, pc);
    const dim3 grid(dchunks * pchunks, G, d->N), block(256);
#define PASN_POOL(T, PC)                                                                                              \
    hipLaunchKernelGGL((xproto_pool_kernel<T, PC>), grid, block, 0, s, (const T*)b_occ, (const T*)b_f, occ, b_slabs, \
                       d->S, d->P, d->Pp, d->D, d->Dp, G, dchunks, full ? 1 : 0)
    if (dtype == PASN_F32) {
        if (pc == 32) PASN_POOL(float, 32); else PASN_POOL(float, 64);
    } else {
        if (pc == 32) PASN_POOL(__bf16, 32); else PASN_POOL(__bf16, 64);
    }
#undef PASN_POOL
    }
    if ((rc = check_launch("xproto_pool_kernel"))) return rc;
    if (full) {
        PASN_REQUIRE(d->D % 4 == 0 && d->D <= 1024, "prototype dimension must be a multiple of 4, at most 1024");
        if (d->D <= 256)
            hipLaunchKernelGGL((xproto_finish_kernel<1, 1024>), dim3(d->N), dim3(1024), (size_t)d->P * sizeof(float), s, b_slabs,
                               protos, fc_w, feat, sim, logits, G, d->P, d->D, d->K);
        else
            hipLaunchKernelGGL((xproto_finish_kernel<4, 256>), dim3(d->N), dim3(256), (size_t)d->P * sizeof(float), s, b_slabs,
                               protos, fc_w, feat, sim, logits, G, d->P, d->D, d->K);
        if ((rc = check_launch("xproto_finish_kernel"))) return rc;
    }
    return PASN_OK;
}

// ---- head B as two launches (head_chain.hip + the finish kernel): bf16, D = 256, trunk channel stride <= 192 ----------------------------
// Same contract as pasn_xproto_head_fwd except that the five conv weights are FRAGMENT-MAJOR ([rows / 32][kc / 16][64 lanes][8], the layout
// pasn_conv3d_fwd takes with w_frag = 1) and the workspace holds only the pooling slabs.
extern "C" int pasn_xproto_chain_supported(const pasn_xproto_desc* d, int dtype) {
    return xp_desc_ok(d) && (d->mode == 0 || d->mode == 1) && xproto_chain_supported(*d, dtype) ? 1 : 0;
}

extern "C" size_t pasn_xproto_chain_workspace_bytes(const pasn_xproto_desc* d) {
    if (!xp_desc_ok(d)) return 0;
    return align256((size_t)d->N * xproto_chain_tiles(*d) * d->P * d->D * sizeof(float));
}

extern "C" int pasn_xproto_chain_fwd(const void* x, const void* a1, const float* a1b, const void* a2, const float* a2b, const void* o1,
                                     const float* o1b, const void* o2, const float* o2b, const void* o3, const float* protos,
                                     const float* fc_w, float* occ, float* feat, float* sim, float* logits, void* ws,
                                     const pasn_xproto_desc* d, int dtype, void* stream) {
    PASN_REQUIRE(pasn_xproto_chain_supported(d, dtype), "shape / dtype outside the chained head (see pasn_xproto_chain_supported)");
    PASN_REQUIRE(x && o1 && o1b && o2 && o2b && o3 && occ, "null pointer");
    const bool full = d->mode == 0;
    if (full) {
        PASN_REQUIRE(a1 && a1b && a2 && a2b && protos && fc_w && feat && sim && logits && ws, "null pointer (full mode)");
        PASN_REQUIRE(((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
    }
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = launch_xproto_chain(x, a1, a1b, a2, a2b, o1, o1b, o2, o2b, o3, occ, (float*)ws, *d, s))) return rc;
    if (full) {
        hipLaunchKernelGGL((xproto_finish_kernel<1, 1024>), dim3(d->N), dim3(1024), (size_t)d->P * sizeof(float), s, (const float*)ws, protos, fc_w,
                           feat, sim, logits, xproto_chain_tiles(*d), d->P, d->D, d->K);
        if ((rc = check_launch("xproto_finish_kernel"))) return rc;
    }
    return PASN_OK;
}

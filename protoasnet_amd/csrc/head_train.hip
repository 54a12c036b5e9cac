// head_train.hip -- training-mode tail of head B (XProtoNet / Video_XProtoNet): occurrence-weighted pooling, cosine
// similarity, last layer, and their backward.  The 1x1x1 conv chains in front of it run on the conv kernels; what is left
// is small (N*P*D*S MACs) and HBM/latency-bound, so these are plain reduction kernels with a fixed summation order.
//
// Forward  (Video_XProtoNet.py:82-98, XProtoNet.py:51-67):
//   occ[n][p][s] = |r[n][s][p]|,  F[n][p][d] = sum_s occ[n][p][s] * z[n][s][d],
//   sim[n][p] = (cos(F[n][p], proto[p]) + 1) / 2   (each vector divided by max(norm, 1e-8)),  logits = sim . W^T
// Backward: autograd's derivative of the same expressions, given dlogits, an optional extra dsim (cluster / separation
// losses act on the similarities) and an optional extra docc (the occurrence-map losses).
#include "common.h"

namespace pasn {

constexpr float COS_EPS = 1e-8f;

__device__ __forceinline__ float block_sum(float v, float* red) {
    // 256 threads; result broadcast to every thread
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

template <typename T>
__global__ __launch_bounds__(256) void xproto_pool_fwd_kernel(const T* __restrict__ z, const T* __restrict__ r, float* __restrict__ occ,
                                                              float* __restrict__ F, int S, int D, int Dp, int P, int Pp) {
    const int n = blockIdx.x, p0 = blockIdx.y * 8;
    const T* zn = z + (size_t)n * S * Dp;
    const T* rn = r + (size_t)n * S * Pp;
    for (int s = threadIdx.x; s < S; s += 256) {
        float rv[8];
        load8(rn + (size_t)s * Pp + p0, rv);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (p0 + j < P) occ[((size_t)n * P + p0 + j) * S + s] = fabsf(rv[j]);
    }
    if (z == nullptr) return;  // occurrence map only (compute_occurence_map)
    for (int dd = threadIdx.x; dd < D; dd += 256) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
#pragma unroll 4
        for (int s = 0; s < S; ++s) {
            float rv[8];
            load8(rn + (size_t)s * Pp + p0, rv);
            const float zv = (float)zn[(size_t)s * Dp + dd];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(fabsf(rv[j]), zv, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (p0 + j < P) F[((size_t)n * P + p0 + j) * D + dd] = acc[j];
    }
}

__global__ __launch_bounds__(256) void xproto_tail_fwd_kernel(const float* __restrict__ F, const float* __restrict__ protos,
                                                              const float* __restrict__ fc_w, float* __restrict__ sim, float* __restrict__ logits,
                                                              int D, int P, int K) {
    __shared__ float red[4];
    __shared__ float sims[256];
    const int n = blockIdx.x;
    for (int p = 0; p < P; ++p) {
        float dot = 0.0f, nf = 0.0f, np = 0.0f;
        for (int dd = threadIdx.x; dd < D; dd += 256) {
            const float f = F[((size_t)n * P + p) * D + dd], q = protos[(size_t)p * D + dd];
            dot = fmaf(f, q, dot);
            nf = fmaf(f, f, nf);
            np = fmaf(q, q, np);
        }
        dot = block_sum(dot, red);
        nf = block_sum(nf, red);
        np = block_sum(np, red);
        if (threadIdx.x == 0) {
            const float c = dot / (fmaxf(sqrtf(nf), COS_EPS) * fmaxf(sqrtf(np), COS_EPS));
            const float sv = 0.5f * (c + 1.0f);
            sims[p] = sv;
            sim[(size_t)n * P + p] = sv;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) {
        float a = 0.0f;
        for (int p = 0; p < P; ++p) a = fmaf(sims[p], fc_w[(size_t)k * P + p], a);
        logits[(size_t)n * K + k] = a;
    }
}

// one block per prototype: dF[:, p, :], dproto[p, :], dfc_w[:, p].  A WAVE owns a clip (clips wave, wave + 4, ...): the two dot products
// are wave reductions (no block barrier), a lane holds D / 64 elements of the row, and the four clips of a batch are loaded before any
// is reduced.  (First version: the block walked the clips one by one with two block-wide sums each -- 32 x (a load round trip + 4
// barriers), 120 us of latency for 30 blocks.)  dproto: per-wave partial sums in clip order, combined in wave order (fixed).
__global__ __launch_bounds__(256) void xproto_tail_bwd_kernel(const float* __restrict__ F, const float* __restrict__ protos,
                                                              const float* __restrict__ fc_w, const float* __restrict__ sim,
                                                              const float* __restrict__ dlogits, const float* __restrict__ dsim,
                                                              float* __restrict__ dF, float* __restrict__ dprotos, float* __restrict__ dfc_w, int N,
                                                              int D, int P, int K) {
    constexpr int EL = 8;  // elements per lane: D <= 512
    __shared__ float dqs[4][64 * EL];
    const int p = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float q[EL], dq[EL];
    float np = 0.0f;
#pragma unroll
    for (int e = 0; e < EL; ++e) {
        const int dd = lane + 64 * e;
        q[e] = dd < D ? protos[(size_t)p * D + dd] : 0.0f;
        np = fmaf(q[e], q[e], np);
        dq[e] = 0.0f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) np += __shfl_xor(np, o, 64);
    const float nP = fmaxf(sqrtf(np), COS_EPS);
    const bool p_clamped = sqrtf(np) < COS_EPS;
    for (int n = wave; n < N; n += 4) {
        float f[EL];
        float dot = 0.0f, nf = 0.0f;
#pragma unroll
        for (int e = 0; e < EL; ++e) {
            const int dd = lane + 64 * e;
            f[e] = dd < D ? F[((size_t)n * P + p) * D + dd] : 0.0f;
        }
        float ds = dsim ? dsim[(size_t)n * P + p] : 0.0f;
        for (int k = 0; k < K; ++k) ds = fmaf(dlogits[(size_t)n * K + k], fc_w[(size_t)k * P + p], ds);
#pragma unroll
        for (int e = 0; e < EL; ++e) {
            dot = fmaf(f[e], q[e], dot);
            nf = fmaf(f[e], f[e], nf);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            dot += __shfl_xor(dot, o, 64);
            nf += __shfl_xor(nf, o, 64);
        }
        const float nF = fmaxf(sqrtf(nf), COS_EPS);
        const bool f_clamped = sqrtf(nf) < COS_EPS;
        const float dc = 0.5f * ds;
        const float inv = 1.0f / (nF * nP);
#pragma unroll
        for (int e = 0; e < EL; ++e) {
            const int dd = lane + 64 * e;
            // c = dot / (nF nP); a clamped norm is a constant
            const float gf = q[e] * inv - (f_clamped ? 0.0f : dot * f[e] * inv / (nF * nF));
            const float gq = f[e] * inv - (p_clamped ? 0.0f : dot * q[e] * inv / (nP * nP));
            if (dd < D) dF[((size_t)n * P + p) * D + dd] = dc * gf;
            dq[e] = fmaf(dc, gq, dq[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < EL; ++e) dqs[wave][lane + 64 * e] = dq[e];
    __syncthreads();
    for (int dd = threadIdx.x; dd < D; dd += 256) dprotos[(size_t)p * D + dd] = ((dqs[0][dd] + dqs[1][dd]) + dqs[2][dd]) + dqs[3][dd];
    for (int k = threadIdx.x; k < K; k += 256) {
        float a = 0.0f;
        for (int n = 0; n < N; ++n) a = fmaf(dlogits[(size_t)n * K + k], sim[(size_t)n * P + p], a);
        dfc_w[(size_t)k * P + p] = a;
    }
}

constexpr int PB_ROWS = 16;

// dz[n][s][d] = sum_p dF[n][p][d] |r[n][s][p]|;   dr[n][s][p] = sign(r) (sum_d dF[n][p][d] z[n][s][d] + docc[n][p][s])
template <typename T>
__global__ __launch_bounds__(256) void xproto_pool_bwd_kernel(const T* __restrict__ z, const T* __restrict__ r, const float* __restrict__ dF,
                                                              const float* __restrict__ docc, T* __restrict__ dz, T* __restrict__ dr, int S, int D,
                                                              int Dp, int P, int Pp) {
    extern __shared__ float sm[];
    float* zt = sm;                   // [PB_ROWS][D]
    float* rt = zt + PB_ROWS * D;     // [PB_ROWS][Pp]  (signed)
    const int n = blockIdx.x, s0 = blockIdx.y * PB_ROWS;
    const int rows = min(PB_ROWS, S - s0);
    const bool full = z != nullptr;  // false: occurrence map only, dr = sign(r) * docc
    if (full)
        for (int i = threadIdx.x; i < PB_ROWS * D; i += 256) {
            const int row = i / D, dd = i % D;
            zt[i] = row < rows ? (float)z[((size_t)n * S + s0 + row) * Dp + dd] : 0.0f;
        }
    for (int i = threadIdx.x; i < PB_ROWS * Pp; i += 256) {
        const int row = i / Pp, p = i % Pp;
        rt[i] = (row < rows && p < P) ? (float)r[((size_t)n * S + s0 + row) * Pp + p] : 0.0f;
    }
    __syncthreads();
    const float* dFn = dF + (size_t)n * P * D;
    for (int dd = threadIdx.x; full && dd < Dp; dd += 256) {
        float acc[PB_ROWS];
#pragma unroll
        for (int i = 0; i < PB_ROWS; ++i) acc[i] = 0.0f;
        if (dd < D) {
#pragma unroll 10
            for (int p = 0; p < P; ++p) {  // (unrolled: the loads of a batch of prototypes are in flight together)
                const float f = dFn[(size_t)p * D + dd];
#pragma unroll
                for (int i = 0; i < PB_ROWS; ++i) acc[i] = fmaf(f, fabsf(rt[i * Pp + p]), acc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < PB_ROWS; ++i)
            if (i < rows) dz[((size_t)n * S + s0 + i) * Dp + dd] = (T)acc[i];
    }
    const int row = threadIdx.x >> 4, pl = threadIdx.x & 15;
    for (int p = pl; p < Pp; p += 16) {
        float a = 0.0f;
        if (p < P && row < rows) {
            const float* f = dFn + (size_t)p * D;
#pragma unroll 16
            for (int dd = 0; full && dd < D; ++dd) a = fmaf(f[dd], zt[row * D + dd], a);
            if (docc) a += docc[((size_t)n * P + p) * S + s0 + row];
            const float rv = rt[row * Pp + p];
            a = rv > 0.0f ? a : (rv < 0.0f ? -a : 0.0f);
        }
        if (row < rows) dr[((size_t)n * S + s0 + row) * Pp + p] = (T)a;
    }
}

}  // namespace pasn

using namespace pasn;

extern "C" int pasn_xproto_tail_fwd(const void* z, const void* r, const float* protos, const float* fc_w, float* occ, float* feat,
                                    float* sim, float* logits, const pasn_xproto_desc* d, int dtype, void* stream) {
    PASN_REQUIRE(r && occ && d, "null pointer");
    PASN_REQUIRE(z == nullptr || (protos && fc_w && feat && sim && logits), "null pointer (only the occurrence-map mode, z == NULL, may omit them)");
    PASN_REQUIRE(d->Dp % 8 == 0 && d->Pp % 8 == 0 && d->P <= 256 && d->P <= d->Pp && d->D <= d->Dp, "bad head extents (P <= 256)");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(d->N, d->Pp / 8);
    if (dtype == PASN_BF16)
        hipLaunchKernelGGL(xproto_pool_fwd_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)z, (const __bf16*)r, occ, feat, d->S, d->D, d->Dp,
                           d->P, d->Pp);
    else
        hipLaunchKernelGGL(xproto_pool_fwd_kernel<float>, grid, dim3(256), 0, s, (const float*)z, (const float*)r, occ, feat, d->S, d->D, d->Dp, d->P,
                           d->Pp);
    if (z) hipLaunchKernelGGL(xproto_tail_fwd_kernel, dim3(d->N), dim3(256), 0, s, feat, protos, fc_w, sim, logits, d->D, d->P, d->K);
    return check_launch("xproto_tail_fwd");
}

namespace pasn {
int xproto_tail_splits(const pasn_xproto_desc& d);
int xproto_tail_pool_finish(const void* z, const void* r, const float* protos, const float* fc_w, float* occ, float* feat, float* sim,
                            float* logits, float* slabs, const pasn_xproto_desc& d, int dtype, hipStream_t s);
}  // namespace pasn

// The same forward with a workspace of pasn_xproto_tail_workspace_bytes(d) bytes: pooling on the matrix cores split over S (the kernels of
// the inference head) instead of N x P/8 blocks that each walk all S positions (R(2+1)D-18, 8 clips: 32 blocks, 438 us per step).
extern "C" size_t pasn_xproto_tail_workspace_bytes(const pasn_xproto_desc* d) {
    if (!d || d->N <= 0 || d->S <= 0 || d->P <= 0 || d->D <= 0) return 0;
    if (const char* e = tune("PASN_NO_TAIL_MFMA"))
        if (e[0] == '1') return 0;
    if (d->D % 4 != 0 || d->D > 1024) return 0;
    return (size_t)d->N * xproto_tail_splits(*d) * d->P * d->D * sizeof(float);
}

extern "C" int pasn_xproto_tail_fwd_ws(const void* z, const void* r, const float* protos, const float* fc_w, float* occ, float* feat,
                                       float* sim, float* logits, const pasn_xproto_desc* d, int dtype, void* ws, void* stream) {
    if (!ws || pasn_xproto_tail_workspace_bytes(d) == 0) return pasn_xproto_tail_fwd(z, r, protos, fc_w, occ, feat, sim, logits, d, dtype, stream);
    PASN_REQUIRE(r && occ && d, "null pointer");
    PASN_REQUIRE(z == nullptr || (protos && fc_w && feat && sim && logits), "null pointer (only the occurrence-map mode, z == NULL, may omit them)");
    PASN_REQUIRE(d->Dp % 8 == 0 && d->Pp % 8 == 0 && d->P <= 256 && d->P <= d->Pp && d->D <= d->Dp, "bad head extents (P <= 256)");
    return xproto_tail_pool_finish(z, r, protos, fc_w, occ, feat, sim, logits, (float*)ws, *d, dtype, (hipStream_t)stream);
}

extern "C" int pasn_xproto_tail_bwd(const void* z, const void* r, const float* protos, const float* fc_w, const float* feat, const float* sim,
                                    const float* dlogits, const float* dsim, const float* docc, float* dfeat, void* dz, void* dr,
                                    float* dprotos, float* dfc_w, const pasn_xproto_desc* d, int dtype, void* stream) {
    PASN_REQUIRE(r && dr && d, "null pointer");
    PASN_REQUIRE(z ? (protos && fc_w && feat && sim && dlogits && dfeat && dz && dprotos && dfc_w) : (docc != nullptr),
                 "null pointer (the occurrence-map mode, z == NULL, needs only r, docc, dr)");
    PASN_REQUIRE(d->Dp % 8 == 0 && d->Pp % 8 == 0 && d->P <= 256 && d->D <= 512, "bad head extents (P <= 256, D <= 512)");
    const size_t lds = ((size_t)PB_ROWS * d->D + (size_t)PB_ROWS * d->Pp) * sizeof(float);
    PASN_REQUIRE(lds <= 64 * 1024, "head tile above the LDS budget");
    hipStream_t s = (hipStream_t)stream;
    if (z)
        hipLaunchKernelGGL(xproto_tail_bwd_kernel, dim3(d->P), dim3(256), 0, s, feat, protos, fc_w, sim, dlogits, dsim, dfeat, dprotos, dfc_w, d->N, d->D,
                       d->P, d->K);
    const dim3 grid(d->N, ceil_div(d->S, PB_ROWS));
    if (dtype == PASN_BF16)
        hipLaunchKernelGGL(xproto_pool_bwd_kernel<__bf16>, grid, dim3(256), lds, s, (const __bf16*)z, (const __bf16*)r, dfeat, docc, (__bf16*)dz,
                           (__bf16*)dr, d->S, d->D, d->Dp, d->P, d->Pp);
    else
        hipLaunchKernelGGL(xproto_pool_bwd_kernel<float>, grid, dim3(256), lds, s, (const float*)z, (const float*)r, dfeat, docc, (float*)dz,
                           (float*)dr, d->S, d->D, d->Dp, d->P, d->Pp);
    return check_launch("xproto_tail_bwd");
}

// ---- head A (ProtoPNet) backward ------------------------------------------------------------------------------------------
// Forward (pasn_l2_head_fwd; ProtoPNet.py:189-243): dist = relu(|z|^2 - 2 z.p + |p|^2), min over positions (argmin s*),
// sim = log((d+1)/(d+eps)) or -d, logits = sim . W^T.  Only the arg-min position of each (clip, prototype) carries gradient:
//   g[n][p] = dmin[n][p] + (sum_k dlogits[n][k] W[k][p]) * sim'(d),   dz[n][s*][:] += 2 g (z - p),   dp[p][:] += 2 g (p - z)
namespace pasn {

__device__ __forceinline__ float l2_sim(float d, int activation, float eps) { return activation == 0 ? logf((d + 1.0f) / (d + eps)) : -d; }
__device__ __forceinline__ float l2_dsim(float d, int activation, float eps) { return activation == 0 ? 1.0f / (d + 1.0f) - 1.0f / (d + eps) : -1.0f; }

// one block per clip: zero dz[n], then walk the prototypes in order (a thread owns its channels, so two prototypes sharing an
// arg-min position accumulate without a race and in a fixed order)
template <typename T>
__global__ __launch_bounds__(256) void l2_head_bwd_z_kernel(const T* __restrict__ z, const float* __restrict__ protos, const float* __restrict__ fc_w,
                                                            const float* __restrict__ min_dist, const int32_t* __restrict__ argmin,
                                                            const float* __restrict__ dlogits, const float* __restrict__ dmin, T* __restrict__ dz,
                                                            float* __restrict__ coef, int S, int D, int Dp, int P, int K, int activation,
                                                            float eps) {
    const int n = blockIdx.x;
    T* dzn = dz + (size_t)n * S * Dp;
    const T* zn = z + (size_t)n * S * Dp;
    for (size_t i = threadIdx.x; i < (size_t)S * Dp; i += 256) dzn[i] = (T)0.0f;
    __syncthreads();
    for (int p = 0; p < P; ++p) {
        const float d = min_dist[(size_t)n * P + p];
        float ds = 0.0f;
        for (int k = 0; k < K; ++k) ds = fmaf(dlogits[(size_t)n * K + k], fc_w[(size_t)k * P + p], ds);
        const float g = (dmin ? dmin[(size_t)n * P + p] : 0.0f) + ds * l2_dsim(d, activation, eps);
        if (threadIdx.x == 0) coef[(size_t)n * P + p] = g;
        const int s = argmin[(size_t)n * P + p];
        for (int dd = threadIdx.x; dd < D; dd += 256) {
            const size_t o = (size_t)s * Dp + dd;
            dzn[o] = (T)((float)dzn[o] + 2.0f * g * ((float)zn[o] - protos[(size_t)p * D + dd]));
        }
    }
}

// one block per prototype: dprotos[p][:], dfc_w[:][p]  (fixed order over the clips)
template <typename T>
__global__ __launch_bounds__(256) void l2_head_bwd_p_kernel(const T* __restrict__ z, const float* __restrict__ protos, const float* __restrict__ min_dist,
                                                            const int32_t* __restrict__ argmin, const float* __restrict__ dlogits,
                                                            const float* __restrict__ coef, float* __restrict__ dprotos, float* __restrict__ dfc_w,
                                                            int N, int S, int D, int Dp, int P, int K, int activation, float eps) {
    const int p = blockIdx.x;
    for (int dd = threadIdx.x; dd < D; dd += 256) {
        const float q = protos[(size_t)p * D + dd];
        float a = 0.0f;
        for (int n = 0; n < N; ++n) {
            const int s = argmin[(size_t)n * P + p];
            a = fmaf(2.0f * coef[(size_t)n * P + p], q - (float)z[((size_t)n * S + s) * Dp + dd], a);
        }
        dprotos[(size_t)p * D + dd] = a;
    }
    for (int k = threadIdx.x; k < K; k += 256) {
        float a = 0.0f;
        for (int n = 0; n < N; ++n) a = fmaf(dlogits[(size_t)n * K + k], l2_sim(min_dist[(size_t)n * P + p], activation, eps), a);
        dfc_w[(size_t)k * P + p] = a;
    }
}

}  // namespace pasn

extern "C" int pasn_l2_head_bwd(const void* z, const float* protos, const float* fc_w, const float* min_dist, const int32_t* argmin,
                                const float* dlogits, const float* dmin, void* dz, float* coef, float* dprotos, float* dfc_w, int N, int S,
                                int D, int Dp, int P, int K, int dtype, int activation, float eps, void* stream) {
    PASN_REQUIRE(z && protos && fc_w && min_dist && argmin && dlogits && dz && coef && dprotos && dfc_w, "null pointer");
    PASN_REQUIRE(N > 0 && S > 0 && D > 0 && P > 0 && K > 0 && Dp >= D && Dp % 8 == 0, "bad head extents");
    PASN_REQUIRE(activation == 0 || activation == 1, "activation must be 0 (log) or 1 (linear)");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASN_BF16) {
        hipLaunchKernelGGL(l2_head_bwd_z_kernel<__bf16>, dim3(N), dim3(256), 0, s, (const __bf16*)z, protos, fc_w, min_dist, argmin, dlogits, dmin,
                           (__bf16*)dz, coef, S, D, Dp, P, K, activation, eps);
        hipLaunchKernelGGL(l2_head_bwd_p_kernel<__bf16>, dim3(P), dim3(256), 0, s, (const __bf16*)z, protos, min_dist, argmin, dlogits, coef, dprotos,
                           dfc_w, N, S, D, Dp, P, K, activation, eps);
    } else {
        hipLaunchKernelGGL(l2_head_bwd_z_kernel<float>, dim3(N), dim3(256), 0, s, (const float*)z, protos, fc_w, min_dist, argmin, dlogits, dmin,
                           (float*)dz, coef, S, D, Dp, P, K, activation, eps);
        hipLaunchKernelGGL(l2_head_bwd_p_kernel<float>, dim3(P), dim3(256), 0, s, (const float*)z, protos, min_dist, argmin, dlogits, coef, dprotos, dfc_w,
                           N, S, D, Dp, P, K, activation, eps);
    }
    return check_launch("l2_head_bwd");
}

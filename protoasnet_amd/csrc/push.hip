// Prototype push sweeps: per-prototype running (min distance, source index, source vector) kept on the device.
// Only P*(D+2) words ever leave the GPU at the end of a sweep, instead of the per-batch host copies of
// features / distances / occurrence maps / images the reference makes (push_abs_revision.py:278-285).
#include "common.h"

namespace pasn {

// One wave per prototype.  Tie rules are the reference's:
//   inside a batch  : first index attaining the batch minimum (np.argmin, push_abs_revision.py:300)
//   across batches  : accepted when batch_min <= best (push_abs_revision.py:299) -- a later batch wins ties.
__global__ __launch_bounds__(64) void push_xproto_kernel(const float* __restrict__ proto_dist, const float* __restrict__ feat,
                                                         const int64_t* __restrict__ labels, const int32_t* __restrict__ proto_class,
                                                         const int32_t* __restrict__ class_mask, float* __restrict__ best_dist,
                                                         int64_t* __restrict__ best_index, float* __restrict__ best_feat, int B,
                                                         int P, int D, int64_t index_base) {
    const int j = blockIdx.x;
    const int lane = threadIdx.x;
    const bool masked = class_mask[j] != 0;
    const int64_t cls = proto_class[j];
    float v = INFINITY;
    int idx = 0x7fffffff;
    for (int b = lane; b < B; b += 64) {
        if (masked && labels[b] != cls) continue;
        const float dv = proto_dist[(long)b * P + j];
        if (dv < v || idx == 0x7fffffff) {  // b grows per lane: strict '<' keeps the first; the first eligible clip always enters
            v = dv;
            idx = b;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(idx, off);
        if (oi != 0x7fffffff && (idx == 0x7fffffff || ov < v || (ov == v && oi < idx))) {
            v = ov;
            idx = oi;
        }
    }
    if (idx == 0x7fffffff) return;          // no clip of this prototype's class in the batch (mask.all(): continue)
    if (!(v <= best_dist[j])) return;       // '<=': later batch wins a tie
    if (lane == 0) {
        best_dist[j] = v;
        best_index[j] = index_base + idx;
    }
    const float* src = feat + ((long)idx * P + j) * D;
    for (int d = lane; d < D; d += 64) best_feat[(long)j * D + d] = src[d];
}

// PPNet rule (push_ProtoPNet.py:198-235): argmin over the flattened (n_c, h, w) of the class-filtered distance
// map -- images keep their batch order, so "first flattened index" = smallest (b, s) lexicographically --
// accepted on strict '<' (the first batch wins ties).  One block (256 threads) per prototype.
template <typename T>
__global__ __launch_bounds__(256) void push_ppnet_kernel(const float* __restrict__ dist, const T* __restrict__ z,
                                                         const int64_t* __restrict__ labels, const int32_t* __restrict__ proto_class,
                                                         int class_specific, float* __restrict__ best_dist,
                                                         int64_t* __restrict__ best_index, float* __restrict__ best_patch, int B,
                                                         int P, int S, int D, int Dp, int64_t index_base) {
    __shared__ float sv[4];
    __shared__ long si[4];
    const int j = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t cls = proto_class[j];
    float v = INFINITY;
    long idx = -1;  // flattened b*S + s
    const long total = (long)B * S;
    for (long i = threadIdx.x; i < total; i += 256) {
        const int b = (int)(i / S);
        if (class_specific && labels[b] != cls) continue;
        const int s = (int)(i % S);
        const float dv = dist[((long)b * P + j) * S + s];
        if (idx < 0 || dv < v) {
            v = dv;
            idx = i;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off);
        const long oi = __shfl_xor(idx, off);
        if (oi >= 0 && (idx < 0 || ov < v || (ov == v && oi < idx))) {
            v = ov;
            idx = oi;
        }
    }
    if (lane == 0) {
        sv[wave] = v;
        si[wave] = idx;
    }
    __syncthreads();
    v = sv[0];
    idx = si[0];
    for (int q = 1; q < 4; ++q) {
        if (si[q] >= 0 && (idx < 0 || sv[q] < v || (sv[q] == v && si[q] < idx))) {
            v = sv[q];
            idx = si[q];
        }
    }
    if (idx < 0) return;                   // no image of the target class in this batch
    if (!(v < best_dist[j])) return;       // strict '<': first batch wins a tie
    __syncthreads();                       // every thread has read best_dist[j] before it is overwritten
    const int b = (int)(idx / S), s = (int)(idx % S);
    if (threadIdx.x == 0) {
        best_dist[j] = v;
        best_index[2 * j + 0] = index_base + b;
        best_index[2 * j + 1] = s;
    }
    const T* src = z + ((long)b * S + s) * Dp;
    for (int d = threadIdx.x; d < D; d += 256) best_patch[(long)j * D + d] = (float)src[d];
}

}  // namespace pasn

using namespace pasn;

extern "C" int pasn_push_xproto_update(const float* proto_dist, const float* feat, const int64_t* labels,
                                       const int32_t* proto_class, const int32_t* class_mask, float* best_dist,
                                       int64_t* best_index, float* best_feat, int B, int P, int D, int64_t index_base,
                                       void* stream) {
    PASN_REQUIRE(proto_dist && feat && labels && proto_class && class_mask && best_dist && best_index && best_feat, "null pointer");
    PASN_REQUIRE(B > 0 && P > 0 && D > 0, "empty problem");
    hipLaunchKernelGGL(push_xproto_kernel, dim3(P), dim3(64), 0, (hipStream_t)stream, proto_dist, feat, labels, proto_class,
                       class_mask, best_dist, best_index, best_feat, B, P, D, index_base);
    return check_launch("push_xproto_kernel");
}

extern "C" int pasn_push_ppnet_update(const float* dist, const void* z, const int64_t* labels, const int32_t* proto_class,
                                      int class_specific, float* best_dist, int64_t* best_index, float* best_patch, int B,
                                      int P, int S, int D, int Dp, int dtype, int64_t index_base, void* stream) {
    PASN_REQUIRE(dist && z && labels && proto_class && best_dist && best_index && best_patch, "null pointer");
    PASN_REQUIRE(B > 0 && P > 0 && S > 0 && D > 0 && Dp >= D, "empty problem");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASN_F32)
        hipLaunchKernelGGL((push_ppnet_kernel<float>), dim3(P), dim3(256), 0, s, dist, (const float*)z, labels, proto_class,
                           class_specific, best_dist, best_index, best_patch, B, P, S, D, Dp, index_base);
    else if (dtype == PASN_BF16)
        hipLaunchKernelGGL((push_ppnet_kernel<__bf16>), dim3(P), dim3(256), 0, s, dist, (const __bf16*)z, labels, proto_class,
                           class_specific, best_dist, best_index, best_patch, B, P, S, D, Dp, index_base);
    else {
        set_error("pasn_push_ppnet_update: unknown dtype");
        return PASN_ERR_ARG;
    }
    return check_launch("push_ppnet_kernel");
}

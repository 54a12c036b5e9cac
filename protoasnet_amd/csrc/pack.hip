// Training: every conv weight of a step re-packed from the live fp32 parameters in ONE launch.
//
// The training pass launches the forward kernels of the inference path, which read weights in their own layouts ([row][tap][k] padded,
// fragment-major for the x-tile pointwise kernels, [tap][Cp] for the depthwise stencils; transposed with reversed taps for the input
// gradients).  Parameters change every optimizer step, so the packing is per step: as torch expressions that was ~300 launches of 2-5 us
// per X3D-S step (slice-assign with cast, permuted copy, flip: 1.2 ms of a 27.7 ms step, and ~3 ms of host time).  Here a table of jobs
// in device memory (built once per plan: the parameters' storage is stable across optimizer steps) drives one kernel; every element of
// every destination is written (zero where the layout pads), so the destinations need no clearing.
#include "common.h"

namespace pasn {

constexpr int PACK_CHUNK = 2048;  // destination elements per block (256 threads x 8)

__global__ __launch_bounds__(256) void pack_weights_kernel(const pasn_pack_job* __restrict__ jobs, const int* __restrict__ block_job,
                                                           const int* __restrict__ block_chunk) {
    const pasn_pack_job j = jobs[block_job[blockIdx.x]];
    const long i0 = (long)block_chunk[blockIdx.x] * PACK_CHUNK;
#pragma unroll
    for (int e = 0; e < PACK_CHUNK / 256; ++e) {
        const long i = i0 + e * 256 + threadIdx.x;
        if (i >= j.n) break;
        bool valid;
        long sidx;
        if (j.mode >= 2) {  // depthwise [c][taps] -> [tap][Cp] (mode 3: taps reversed = the stencil of the input gradient)
            const int c = (int)(i % j.kc), tap = (int)(i / j.kc);
            valid = c < j.cout;
            sidx = (long)c * j.taps + (j.mode == 3 ? j.taps - 1 - tap : tap);
        } else {
            int row, k, tap = 0;
            if (j.frag) {  // [row tile][k-step][lane half][32 rows][ch]: 64 lanes x 16 bytes per (tile, step), pointwise only
                long t = i;
                const int c = (int)(t % j.ch);
                t /= j.ch;
                const int r32 = (int)(t % 32);
                t /= 32;
                const int h = (int)(t % 2);
                t /= 2;
                const int nks = j.kc / j.kstep;
                const int ks = (int)(t % nks);
                row = (int)(t / nks) * 32 + r32;
                k = ks * j.kstep + h * j.ch + c;
            } else {  // [row][tap][k]
                k = (int)(i % j.kc);
                const long t = i / j.kc;
                tap = (int)(t % j.taps);
                row = (int)(t / j.taps);
            }
            if (j.mode == 0) {  // forward: row = output channel, k = input channel
                valid = row < j.cout && k < j.cin;
                sidx = ((long)row * j.cin + k) * j.taps + tap;
            } else {  // input gradient: row = the conv's input channel, k = its output channel, taps reversed
                valid = row < j.cin && k < j.cout;
                sidx = ((long)k * j.cin + row) * j.taps + (j.taps - 1 - tap);
            }
        }
        const float v = valid ? j.src[sidx] : 0.0f;
        if (j.bf16) reinterpret_cast<__bf16*>(j.dst)[i] = (__bf16)v;
        else reinterpret_cast<float*>(j.dst)[i] = v;
    }
}

}  // namespace pasn

extern "C" int pasn_pack_chunk(void) { return pasn::PACK_CHUNK; }

extern "C" int pasn_pack_weights(const pasn_pack_job* jobs, const int* block_job, const int* block_chunk, int nblocks, void* stream) {
    using namespace pasn;
    if (nblocks <= 0) return PASN_OK;
    PASN_REQUIRE(jobs && block_job && block_chunk, "pasn_pack_weights: null table");
    hipLaunchKernelGGL(pack_weights_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, jobs, block_job, block_chunk);
    return check_launch("pack_weights_kernel");
}

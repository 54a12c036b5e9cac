// VALU issue-rate probe for gfx950 (optimisation tool, not part of the product): cycles per wave-instruction of the candidate
// inner-loop instructions of the depthwise stencil, at 1 / 2 / 4 waves per SIMD, every CU busy.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;

constexpr int NACC = 24;   // independent chains, like the stencil's WT x 8 accumulators
constexpr int ITERS = 512;

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, const unsigned* in, unsigned long long* cyc) {
    float acc[NACC];
    unsigned a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = in[threadIdx.x + 64 * i];
        b[i] = in[threadIdx.x + 64 * i + 512];
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            const unsigned x = a[i & 7], y = b[i & 7];
            if constexpr (MODE == 0) {  // v_fma_f32
                acc[i] = __builtin_fmaf(__uint_as_float(x), __uint_as_float(y), acc[i]);
            } else if constexpr (MODE == 1) {  // v_pk_fma_f32 (two chains per instruction)
                if (i % 2 == 0) {
                    f32x2 c = {acc[i], acc[i + 1]};
                    f32x2 u = {__uint_as_float(a[i & 7]), __uint_as_float(a[(i + 1) & 7])};
                    f32x2 v = {__uint_as_float(b[i & 7]), __uint_as_float(b[(i + 1) & 7])};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(u), "v"(v));
                    acc[i] = c[0];
                    acc[i + 1] = c[1];
                }
            } else if constexpr (MODE == 2) {  // v_dot2c_f32_bf16
                asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
            } else if constexpr (MODE == 3) {  // v_perm_b32
                unsigned r;
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(__float_as_uint(acc[i])), "v"(y));
                acc[i] = __uint_as_float(r);
            } else if constexpr (MODE == 4) {  // v_lshlrev_b32 (the bf16 -> fp32 conversion)
                unsigned r;
                asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(r) : "v"(__float_as_uint(acc[i]) + x));
                acc[i] = __uint_as_float(r);
            } else if constexpr (MODE == 5) {  // v_pk_fma_f16
                unsigned r = __float_as_uint(acc[i]);
                asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(r) : "v"(x), "v"(y));
                acc[i] = __uint_as_float(r);
            } else if constexpr (MODE == 6) {  // v_cndmask_b32
                unsigned r;
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r) : "v"(x), "v"(__float_as_uint(acc[i])));
                acc[i] = __uint_as_float(r);
            } else if constexpr (MODE == 7) {  // v_dot2_f32_bf16 VOP3P form if the assembler has it
                asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

template <int MODE>
void run(const char* name, int per_instr_div, float* out, unsigned* in, unsigned long long* cyc) {
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        probe<MODE><<<blocks, 256>>>(out, in, cyc);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        probe<MODE><<<blocks, 256>>>(out, in, cyc);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2048];
        CK(hipMemcpy(h, cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double mean = 0;
        for (int i = 0; i < blocks; ++i) mean += (double)h[i];
        mean /= blocks;
        const double instr = (double)ITERS * NACC / per_instr_div;  // wave-instructions per wave
        // memtime ticks at 100 MHz on gfx9; convert through the wall clock as a cross-check
        printf("%-18s waves/SIMD=%d  wall %.1f us  memtime/instr %.3f  => wall ns per instr per SIMD-slot %.3f\n", name, wps, ms * 1e3,
               mean / instr, ms * 1e6 / (instr * wps));
    }
}

int main() {
    float* out;
    unsigned* in;
    unsigned long long* cyc;
    CK(hipMalloc(&out, 2048 * 256 * 4));
    CK(hipMalloc(&in, 4096 * 4));
    CK(hipMalloc(&cyc, 2048 * 8));
    unsigned h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = 0x3f803f80u + (i & 7);
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    run<0>("v_fma_f32", 1, out, in, cyc);
    run<1>("v_pk_fma_f32", 2, out, in, cyc);
    run<2>("v_dot2c_f32_bf16", 1, out, in, cyc);
    run<3>("v_perm_b32", 1, out, in, cyc);
    run<4>("v_lshlrev+add", 1, out, in, cyc);
    run<5>("v_pk_fma_f16", 1, out, in, cyc);
    run<6>("v_cndmask_b32", 1, out, in, cyc);
    return 0;
}

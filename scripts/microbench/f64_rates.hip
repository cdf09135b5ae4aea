// Issue rates of the f64 instructions the B-spline transform is made of, on gfx950: v_fma_f64, v_cvt_f64_f32, v_cvt_f32_f64,
// v_fma_f32 for scale.  One block of 64 lanes per SIMD slot x 4, long dependent-free chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int MODE> __global__ __launch_bounds__(256) void k(int iters, const float *in, double *out)
{
    float f[8]; double d[8];
    for (int j = 0; j < 8; j++) { f[j] = in[(threadIdx.x + j) & 63]; d[j] = (double)f[j]; }
    const double m = d[0] * 1e-9 + 1.0;
    const float mf = (float)m;
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 0) d[j] = fma(d[j], m, 1e-3);                                   // v_fma_f64
            else if (MODE == 1) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[j]) : "v"(f[j])); f[j] += 1.0f; }     // cvt + v_add_f32
            else if (MODE == 2) f[j] = fmaf(f[j], mf, 1e-3f);                           // v_fma_f32
            else if (MODE == 3) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[j]) : "v"(d[j])); d[j] += 1.0; }      // cvt + v_add_f64
            else if (MODE == 4) { f[j] += 1.0f; }                                        // v_add_f32 alone
            else if (MODE == 5) { d[j] += 1.0; }                                         // v_add_f64 alone
            else if (MODE == 6) { double t; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(t) : "v"(f[j])); d[j] = fma(t, m, d[j]); f[j] += 1.0f; }  // the transform's triple: cvt, fma, (+ add f32 to keep f live)
        }
    }
    double s = 0;
    for (int j = 0; j < 8; j++) s += d[j] + (double)f[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, double per_iter)
{
    float *in; double *out; CK(hipMalloc(&in, 256)); CK(hipMemset(in, 0, 256)); CK(hipMalloc(&out, 8192 * 256 * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = 8192, iters = 2048;
    k<MODE><<<blocks, 256>>>(16, in, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k<MODE><<<blocks, 256>>>(iters, in, out);
    CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double wave_instr = (double)blocks * 4 * iters * 8 * per_iter;          // wavefront instructions issued
    // cycles per wavefront instruction and SIMD at 2.4 GHz (1024 SIMDs)
    printf("%-44s %8.3f ms  %6.2f cycles per wave64 instruction per SIMD (assuming 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / wave_instr);
    CK(hipFree(in)); CK(hipFree(out));
}
int main()
{
    run<0>("v_fma_f64", 1);
    run<2>("v_fma_f32", 1);
    run<4>("v_add_f32", 1);
    run<5>("v_add_f64", 1);
    run<1>("v_cvt_f64_f32 + v_add_f32 (2 instr)", 2);
    run<3>("v_cvt_f32_f64 + v_add_f64 (2 instr)", 2);
    run<6>("v_cvt_f64_f32 + v_fma_f64 + v_add_f32 (3 instr)", 3);
    return 0;
}

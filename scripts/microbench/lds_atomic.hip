// LDS accumulate throughput on gfx950: float atomic vs integer atomic vs plain RMW.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int N = 8192;   // floats of LDS per block
template <int MODE> __global__ __launch_bounds__(256) void k(int iters, unsigned stride, float *out)
{
    __shared__ float buf[N];
    for (int i = threadIdx.x; i < N; i += 256) buf[i] = 0.f;
    __syncthreads();
    unsigned idx = (threadIdx.x * stride) % N;
    float v = 1.0f + threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) atomicAdd(&buf[idx], v);                                   // ds_add_f32
        else if (MODE == 1) atomicAdd((unsigned *)&buf[idx], (unsigned)it);       // ds_add_u32
        else if (MODE == 2) { buf[idx] += v; }                                     // read + write
        else if (MODE == 3) atomicAdd((unsigned long long *)&buf[idx & ~1u], (unsigned long long)it); // ds_add_u64
        else if (MODE == 4) { float4 *p = (float4 *)&buf[idx & ~3u]; float4 t = *p; t.x += v; t.y += v; t.z += v; t.w += v; *p = t; }
        idx = (idx + 64 * stride + 1) % N;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = buf[1] + buf[N - 1];
}
template <int MODE> void run(const char *name, unsigned stride)
{
    float *out; CK(hipMalloc(&out, 4096 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = 2048, iters = 4096;
    k<MODE><<<blocks, 256>>>(16, stride, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k<MODE><<<blocks, 256>>>(iters, stride, out);
    CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double lane_ops = (double)blocks * 256 * iters;
    printf("%-28s stride %2u: %8.3f ms  %7.1f G lane-ops/s  (%.2f lane-ops/clk/CU at 2.4 GHz, 256 CUs)\n", name, stride, ms,
           lane_ops / ms / 1e6, lane_ops / (ms * 1e-3) / 256 / 2.4e9);
    CK(hipFree(out));
}
int main()
{
    for (unsigned s : {1u, 4u}) {
        run<0>("ds_add_f32 (atomicAdd float)", s);
        run<1>("ds_add_u32 (atomicAdd uint)", s);
        run<2>("plain f32 read+add+write", s);
        run<3>("ds_add_u64", s);
        run<4>("float4 read+add+write", s);
    }
    return 0;
}

// chain.hip -- forward evaluation and Jacobian of a FROG transform chain (include/frog_chain.h).
// One thread per point, f64; the links live in device memory in application order.  The lattices
// are small (<= a few 10^4 control points) and every thread of a wavefront reads nearby taps, so
// the coefficient loads are L1/L2 hits; the kernel is f64-ALU work (4^3 taps x 12 products per link).
#include <hip/hip_runtime.h>

#include "frog_chain.h"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace frog { void set_last_error(const std::string &s); }

namespace {

struct DevLink {
    int type;
    double m[12];                   // linear: 3 rows of 4
    int dims[3];
    double origin[3], spacing[3];
    const float *coeffs;
};

__device__ __forceinline__ void basis(double f, double F[4], double G[4])
{
    F[3] = f * f * f / 6;
    F[0] = (f * f - f) / 2 - F[3] + 1.0 / 6;
    F[2] = f + F[0] - F[3] * 2;
    F[1] = 1 - F[0] - F[2] - F[3];
    G[0] = -(1 - f) * (1 - f) / 2;
    G[1] = 1.5 * f * f - 2 * f;
    G[2] = -1.5 * f * f + f + 0.5;
    G[3] = f * f / 2;
}

template <bool JAC>
__device__ void chain_point(const DevLink *links, int n_links, double p[3], double A[3][3])
{
    if (JAC) { A[0][0] = A[1][1] = A[2][2] = 1; A[0][1] = A[0][2] = A[1][0] = A[1][2] = A[2][0] = A[2][1] = 0; }
    for (int l = 0; l < n_links; l++) {
        const DevLink &t = links[l];
        double q[3], J[3][3];
        if (t.type == FROG_T_LINEAR) {
            for (int r = 0; r < 3; r++) {
                q[r] = t.m[4 * r] * p[0] + t.m[4 * r + 1] * p[1] + t.m[4 * r + 2] * p[2] + t.m[4 * r + 3];
                if (JAC) for (int c = 0; c < 3; c++) J[r][c] = t.m[4 * r + c];
            }
        } else {
            double F[3][4], G[3][4];
            int i0[3];
            for (int k = 0; k < 3; k++) {
                const double u = (p[k] - t.origin[k]) / t.spacing[k];
                const double fl = floor(u);
                i0[k] = (int)fl - 1;
                basis(u - fl, F[k], G[k]);
            }
            double d[3] = { 0, 0, 0 }, dd[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
            for (int k = 0; k < 4; k++) {
                const int z = i0[2] + k;
                if (z < 0 || z >= t.dims[2]) continue;
                for (int j = 0; j < 4; j++) {
                    const int y = i0[1] + j;
                    if (y < 0 || y >= t.dims[1]) continue;
                    for (int i = 0; i < 4; i++) {
                        const int x = i0[0] + i;
                        if (x < 0 || x >= t.dims[0]) continue;
                        const float *c = t.coeffs + 3 * ((size_t)x + (size_t)t.dims[0] * ((size_t)y + (size_t)t.dims[1] * (size_t)z));
                        const double w = F[0][i] * F[1][j] * F[2][k];
                        const double c0 = c[0], c1 = c[1], c2 = c[2];
                        d[0] += w * c0; d[1] += w * c1; d[2] += w * c2;
                        if (JAC) {
                            const double wx = G[0][i] * F[1][j] * F[2][k], wy = F[0][i] * G[1][j] * F[2][k], wz = F[0][i] * F[1][j] * G[2][k];
                            dd[0][0] += wx * c0; dd[0][1] += wy * c0; dd[0][2] += wz * c0;
                            dd[1][0] += wx * c1; dd[1][1] += wy * c1; dd[1][2] += wz * c1;
                            dd[2][0] += wx * c2; dd[2][1] += wy * c2; dd[2][2] += wz * c2;
                        }
                    }
                }
            }
            for (int r = 0; r < 3; r++) {
                q[r] = p[r] + d[r];
                if (JAC) for (int c = 0; c < 3; c++) J[r][c] = (r == c ? 1.0 : 0.0) + dd[r][c] / t.spacing[c];
            }
        }
        if (JAC) {
            double B[3][3];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) B[r][c] = J[r][0] * A[0][c] + J[r][1] * A[1][c] + J[r][2] * A[2][c];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) A[r][c] = B[r][c];
        }
        p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
    }
}

__global__ __launch_bounds__(256) void chain_apply_kernel(const DevLink *links, int n_links, const double *in, double *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double p[3] = { in[3 * i], in[3 * i + 1], in[3 * i + 2] }, A[3][3];
    chain_point<false>(links, n_links, p, A);
    out[3 * i] = p[0]; out[3 * i + 1] = p[1]; out[3 * i + 2] = p[2];
}

// one thread per grid node; block-level reduction of (negative count, minimum determinant)
__global__ __launch_bounds__(256) void chain_check_kernel(const DevLink *links, int n_links, double ox, double oy, double oz,
                                                          double sx, double sy, double sz, uint32_t nx, uint32_t ny, uint32_t nz,
                                                          unsigned long long *n_negative, double *block_min)
{
    __shared__ double mins[256];
    __shared__ unsigned int negs[256];
    const size_t total = (size_t)nx * ny * nz;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double det = INFINITY;
    unsigned int neg = 0;
    if (idx < total) {
        const uint32_t i = (uint32_t)(idx % nx), j = (uint32_t)((idx / nx) % ny), k = (uint32_t)(idx / ((size_t)nx * ny));
        double p[3] = { ox + i * sx, oy + j * sy, oz + k * sz }, A[3][3];
        chain_point<true>(links, n_links, p, A);
        det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0])
            + A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
        neg = det < 0 ? 1u : 0u;
    }
    mins[threadIdx.x] = det; negs[threadIdx.x] = neg;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) { mins[threadIdx.x] = fmin(mins[threadIdx.x], mins[threadIdx.x + h]); negs[threadIdx.x] += negs[threadIdx.x + h]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        block_min[blockIdx.x] = mins[0];
        if (negs[0]) atomicAdd(n_negative, (unsigned long long)negs[0]);
    }
}

int fail(int code, const std::string &msg) { frog::set_last_error(msg); return code; }

#define KCHECK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail(FROG_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

} // namespace

struct frog_chain {
    int device = 0;
    std::vector<DevLink> h_links;
    std::vector<float *> d_coeffs;
    DevLink *d_links = nullptr;
};

extern "C" {

void frog_chain_destroy(frog_chain *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (float *p : c->d_coeffs) if (p) (void)hipFree(p);
    if (c->d_links) (void)hipFree(c->d_links);
    delete c;
}

uint32_t frog_chain_num_links(const frog_chain *c) { return c ? (uint32_t)c->h_links.size() : 0; }

int frog_chain_create(const frog_chain_link *links, uint32_t n_links, int device, frog_chain **out)
{
    if (!out || (n_links && !links)) return fail(FROG_E_INVALID, "bad arguments to frog_chain_create");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(FROG_E_NODEVICE, "no HIP device: no CPU fallback");
    if (device < 0 || device >= count) return fail(FROG_E_INVALID, "bad device index");
    KCHECK(hipSetDevice(device));
    frog_chain *c = new (std::nothrow) frog_chain;
    if (!c) return fail(FROG_E_NOMEM, "out of host memory");
    c->device = device;
    for (uint32_t l = 0; l < n_links; l++) {
        const frog_chain_link &t = links[l];
        DevLink d;
        std::memset(&d, 0, sizeof d);
        d.type = t.type;
        if (t.type == FROG_T_LINEAR) {
            for (int k = 0; k < 12; k++) d.m[k] = t.matrix[k];
            c->d_coeffs.push_back(nullptr);
        } else if (t.type == FROG_T_BSPLINE) {
            const size_t G = (size_t)t.dims[0] * t.dims[1] * t.dims[2];
            if (!G || !t.coeffs) { frog_chain_destroy(c); return fail(FROG_E_INVALID, "empty lattice"); }
            for (int k = 0; k < 3; k++) {
                if (!(t.spacing[k] > 0)) { frog_chain_destroy(c); return fail(FROG_E_INVALID, "lattice spacing must be positive"); }
                d.dims[k] = (int)t.dims[k]; d.origin[k] = t.origin[k]; d.spacing[k] = t.spacing[k];
            }
            float *p = nullptr;
            if (hipMalloc((void **)&p, 3 * G * sizeof(float)) != hipSuccess
                || hipMemcpy(p, t.coeffs, 3 * G * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
                if (p) (void)hipFree(p);
                frog_chain_destroy(c);
                return fail(FROG_E_HIP, "cannot copy lattice coefficients to the device");
            }
            d.coeffs = p;
            c->d_coeffs.push_back(p);
        } else {
            frog_chain_destroy(c);
            return fail(FROG_E_INVALID, "unknown transform type");
        }
        c->h_links.push_back(d);
    }
    if (n_links) {
        if (hipMalloc((void **)&c->d_links, n_links * sizeof(DevLink)) != hipSuccess
            || hipMemcpy(c->d_links, c->h_links.data(), n_links * sizeof(DevLink), hipMemcpyHostToDevice) != hipSuccess) {
            frog_chain_destroy(c);
            return fail(FROG_E_HIP, "cannot copy the chain to the device");
        }
    }
    *out = c;
    return FROG_OK;
}

int frog_chain_apply(frog_chain *c, const double *in, double *out, size_t n)
{
    if (!c || (n && (!in || !out))) return fail(FROG_E_INVALID, "bad arguments to frog_chain_apply");
    if (!n) return FROG_OK;
    KCHECK(hipSetDevice(c->device));
    double *d_in = nullptr, *d_out = nullptr;
    KCHECK(hipMalloc((void **)&d_in, 3 * n * sizeof(double)));
    if (hipMalloc((void **)&d_out, 3 * n * sizeof(double)) != hipSuccess) { (void)hipFree(d_in); return fail(FROG_E_HIP, "hipMalloc"); }
    hipError_t e = hipMemcpy(d_in, in, 3 * n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        chain_apply_kernel<<<(unsigned)((n + 255) / 256), 256>>>(c->d_links, (int)c->h_links.size(), d_in, d_out, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out, 3 * n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_in); (void)hipFree(d_out);
    if (e != hipSuccess) return fail(FROG_E_HIP, std::string("frog_chain_apply: ") + hipGetErrorString(e));
    return FROG_OK;
}

int frog_chain_check(frog_chain *c, const double origin[3], const double spacing[3], const uint32_t dims[3],
                     uint64_t *n_negative, double *min_determinant)
{
    if (!c || !origin || !spacing || !dims || !n_negative) return fail(FROG_E_INVALID, "bad arguments to frog_chain_check");
    const size_t total = (size_t)dims[0] * dims[1] * dims[2];
    *n_negative = 0;
    if (min_determinant) *min_determinant = INFINITY;
    if (!total) return FROG_OK;
    if (total > ((size_t)1 << 40)) return fail(FROG_E_INVALID, "grid too large");
    KCHECK(hipSetDevice(c->device));
    const size_t blocks = (total + 255) / 256;
    unsigned long long *d_neg = nullptr;
    double *d_min = nullptr;
    KCHECK(hipMalloc((void **)&d_neg, sizeof(unsigned long long)));
    if (hipMalloc((void **)&d_min, blocks * sizeof(double)) != hipSuccess) { (void)hipFree(d_neg); return fail(FROG_E_HIP, "hipMalloc"); }
    hipError_t e = hipMemset(d_neg, 0, sizeof(unsigned long long));
    if (e == hipSuccess) {
        chain_check_kernel<<<(unsigned)blocks, 256>>>(c->d_links, (int)c->h_links.size(), origin[0], origin[1], origin[2],
                                                      spacing[0], spacing[1], spacing[2], dims[0], dims[1], dims[2], d_neg, d_min);
        e = hipGetLastError();
    }
    unsigned long long neg = 0;
    std::vector<double> mins(blocks);
    if (e == hipSuccess) e = hipMemcpy(&neg, d_neg, sizeof neg, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(mins.data(), d_min, blocks * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_neg); (void)hipFree(d_min);
    if (e != hipSuccess) return fail(FROG_E_HIP, std::string("frog_chain_check: ") + hipGetErrorString(e));
    *n_negative = neg;
    if (min_determinant) { double m = INFINITY; for (double v : mins) m = std::fmin(m, v); *min_determinant = m; }
    return FROG_OK;
}

}

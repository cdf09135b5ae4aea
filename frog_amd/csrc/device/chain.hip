// chain.hip -- forward evaluation and Jacobian of a FROG transform chain (include/frog_chain.h).
// One thread per point, f64; the links live in device memory in application order.  The lattices
// are small (<= a few 10^4 control points) and every thread of a wavefront reads nearby taps, so
// the coefficient loads are L1/L2 hits; the kernel is f64-ALU work (4^3 taps x 12 products per link).
#include <hip/hip_runtime.h>

#include "frog_chain.h"

#include <cmath>
#include <cstring>
#include <limits>
#include <type_traits>
#include <string>
#include <vector>

namespace frog { void set_last_error(const std::string &s); }

namespace {

constexpr double INVERSE_TOLERANCE = 1e-3;      // vtkWarpTransform::InverseTolerance default
constexpr int INVERSE_ITERATIONS = 500;         // vtkWarpTransform::InverseIterations default

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

// one link, forward: q = T(p), J = dT/dp
template <bool JAC>
__device__ __forceinline__ void link_forward(const DevLink &t, const double p[3], double q[3], double J[3][3])
{
    if (t.type == FROG_T_LINEAR) {
        for (int r = 0; r < 3; r++) {
            q[r] = t.m[4 * r] * p[0] + t.m[4 * r + 1] * p[1] + t.m[4 * r + 2] * p[2] + t.m[4 * r + 3];
            if (JAC) for (int c = 0; c < 3; c++) J[r][c] = t.m[4 * r + c];
        }
        return;
    }
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

// delta = J^-1 r  (Cramer; the lattices frog writes with -gd 1 have det J > 0)
__device__ __forceinline__ void solve3(const double J[3][3], const double r[3], double delta[3])
{
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2],
                 c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
    const double inv = 1.0 / det;
    delta[0] = (c00 * r[0] + (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * r[1] + (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * r[2]) * inv;
    delta[1] = (c01 * r[0] + (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * r[1] + (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * r[2]) * inv;
    delta[2] = (c02 * r[0] + (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * r[1] + (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * r[2]) * inv;
}

// Inverse of a B-spline link: the x with T(x) = p, by Newton's method with the step-halving safeguard
// vtkWarpTransform uses for its inverses (first guess x = p - d(p); a step that increases |T(x) - p|^2 is
// shortened by a factor from the parabola through the last two values, clamped to [0.1, 0.5]; stop when
// both the step and the residual are below the tolerance, VTK's default 1e-3, or after 500 iterations,
// then fall back to the best point seen).  q = x, J = (dT/dx)^-1 at x.
template <bool JAC>
__device__ void bspline_inverse(const DevLink &t, const double p[3], double q[3], double Jinv[3][3])
{
    DevLink f = t;
    f.type = FROG_T_BSPLINE;
    const double tol2 = INVERSE_TOLERANCE * INVERSE_TOLERANCE;
    double x[3], fx[3], J[3][3], r[3], delta[3] = { 0, 0, 0 }, last_x[3], last_f = 0, fderiv = 0, frac = 1;
    link_forward<false>(f, p, fx, J);
    for (int k = 0; k < 3; k++) { x[k] = p[k] - (fx[k] - p[k]); last_x[k] = x[k]; }
    int it = 0;
    for (; it < INVERSE_ITERATIONS; it++) {
        link_forward<true>(f, x, fx, J);
        for (int k = 0; k < 3; k++) r[k] = fx[k] - p[k];
        const double fval = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
        if (it == 0 || fval < last_f) {
            solve3(J, r, delta);
            const double err2 = delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2];
            if (err2 < tol2 && fval < tol2) break;
            for (int k = 0; k < 3; k++) last_x[k] = x[k];
            last_f = fval;
            // derivative of |T(x) - p|^2 along -delta at the last point: -2 r.(J delta) = -2 |r|^2
            fderiv = -2.0 * fval;
            for (int k = 0; k < 3; k++) x[k] -= delta[k];
            frac = 1.0;
            continue;
        }
        double a = -fderiv / (2.0 * (fval - last_f - fderiv));
        a = a < 0.1 ? 0.1 : (a > 0.5 ? 0.5 : a);
        frac *= a;
        for (int k = 0; k < 3; k++) x[k] = last_x[k] - frac * delta[k];
    }
    if (it >= INVERSE_ITERATIONS) for (int k = 0; k < 3; k++) x[k] = last_x[k];      // did not converge: best point seen
    for (int k = 0; k < 3; k++) q[k] = x[k];
    if (JAC) {
        link_forward<true>(f, x, fx, J);
        for (int c = 0; c < 3; c++) {
            const double e[3] = { c == 0 ? 1.0 : 0.0, c == 1 ? 1.0 : 0.0, c == 2 ? 1.0 : 0.0 };
            double col[3];
            solve3(J, e, col);
            for (int rr = 0; rr < 3; rr++) Jinv[rr][c] = col[rr];
        }
    }
}

template <bool JAC>
__device__ void chain_point(const DevLink *links, int n_links, double p[3], double A[3][3])
{
    if (JAC) { A[0][0] = A[1][1] = A[2][2] = 1; A[0][1] = A[0][2] = A[1][0] = A[1][2] = A[2][0] = A[2][1] = 0; }
    for (int l = 0; l < n_links; l++) {
        const DevLink &t = links[l];
        double q[3], J[3][3];
        if (t.type == FROG_T_BSPLINE_INVERSE) bspline_inverse<JAC>(t, p, q, J);
        else link_forward<JAC>(t, p, q, J);
        if (JAC) {
            double B[3][3];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) B[r][c] = J[r][0] * A[0][c] + J[r][1] * A[1][c] + J[r][2] * A[2][c];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) A[r][c] = B[r][c];
        }
        p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
    }
}

// ---- vtkImageReslice as tools/VolumeTransform.cxx:119-136 configures it ------------------------------------
// One thread per output voxel: its position in the reference volume's frame goes through the chain (which
// maps output space to source space), the source is sampled there in its own scalar type, the arithmetic is
// f64, the result goes back to that type (integers: rounded half up and clamped, as vtkImageReslice does).
// Inside test with VTK's default
// half-voxel border: a sample up to half a voxel outside the first/last voxel centre still reads the
// edge value (indices clamped); further out it is `background`.
template <class S>
__device__ __forceinline__ S to_voxel(double v)
{
    if constexpr (std::is_integral<S>::value) {
        const double lo = (double)std::numeric_limits<S>::lowest(), hi = (double)std::numeric_limits<S>::max();
        double x = floor(v + 0.5);
        x = x < lo ? lo : (x > hi ? hi : x);
        return (S)x;
    } else {
        return (S)v;
    }
}

template <class S>
__global__ __launch_bounds__(256) void reslice_kernel(const DevLink *links, int n_links, const S *__restrict__ src,
                                                      int sx, int sy, int sz, double so0, double so1, double so2,
                                                      double ss0, double ss1, double ss2,
                                                      uint32_t nx, uint32_t ny, uint32_t nz, double oo0, double oo1, double oo2,
                                                      double os0, double os1, double os2, int linear, double background,
                                                      S *__restrict__ out)
{
    const size_t total = (size_t)nx * ny * nz;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const uint32_t i = (uint32_t)(idx % nx), j = (uint32_t)((idx / nx) % ny), k = (uint32_t)(idx / ((size_t)nx * ny));
    double p[3] = { oo0 + i * os0, oo1 + j * os1, oo2 + k * os2 }, A[3][3];
    chain_point<false>(links, n_links, p, A);
    const double c[3] = { (p[0] - so0) / ss0, (p[1] - so1) / ss1, (p[2] - so2) / ss2 };
    const int dims[3] = { sx, sy, sz };
    double v = background;
    bool inside = true;
    for (int a = 0; a < 3; a++) inside = inside && c[a] >= -0.5 && c[a] <= (double)dims[a] - 0.5;
    if (inside) {
        auto at = [&](int x, int y, int z) -> double {
            x = x < 0 ? 0 : (x >= sx ? sx - 1 : x);
            y = y < 0 ? 0 : (y >= sy ? sy - 1 : y);
            z = z < 0 ? 0 : (z >= sz ? sz - 1 : z);
            return (double)src[(size_t)x + (size_t)sx * ((size_t)y + (size_t)sy * (size_t)z)];
        };
        if (!linear) {
            v = at((int)floor(c[0] + 0.5), (int)floor(c[1] + 0.5), (int)floor(c[2] + 0.5));
        } else {
            const double f0 = floor(c[0]), f1 = floor(c[1]), f2 = floor(c[2]);
            const int x0 = (int)f0, y0 = (int)f1, z0 = (int)f2;
            const double fx = c[0] - f0, fy = c[1] - f1, fz = c[2] - f2;
            const double rx = 1 - fx, ry = 1 - fy, rz = 1 - fz;
            v = rz * (ry * (rx * at(x0, y0, z0) + fx * at(x0 + 1, y0, z0)) + fy * (rx * at(x0, y0 + 1, z0) + fx * at(x0 + 1, y0 + 1, z0)))
              + fz * (ry * (rx * at(x0, y0, z0 + 1) + fx * at(x0 + 1, y0, z0 + 1)) + fy * (rx * at(x0, y0 + 1, z0 + 1) + fx * at(x0 + 1, y0 + 1, z0 + 1)));
        }
    }
    out[idx] = to_voxel<S>(v);
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
        } else if (t.type == FROG_T_BSPLINE || t.type == FROG_T_BSPLINE_INVERSE) {
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


int frog_chain_invert_links(const frog_chain_link *in, uint32_t n, frog_chain_link *out)
{
    if (n && (!in || !out)) return fail(FROG_E_INVALID, "bad arguments to frog_chain_invert_links");
    for (uint32_t l = 0; l < n; l++) {
        frog_chain_link t = in[n - 1 - l];
        if (t.type == FROG_T_BSPLINE) t.type = FROG_T_BSPLINE_INVERSE;
        else if (t.type == FROG_T_BSPLINE_INVERSE) t.type = FROG_T_BSPLINE;
        else if (t.type == FROG_T_LINEAR) {
            // affine inverse: [A b; 0 1]^-1 = [A^-1  -A^-1 b; 0 1]
            const double *m = in[n - 1 - l].matrix;
            const double a00 = m[0], a01 = m[1], a02 = m[2], a10 = m[4], a11 = m[5], a12 = m[6], a20 = m[8], a21 = m[9], a22 = m[10];
            const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
            const double det = a00 * c00 + a01 * c01 + a02 * c02;
            if (det == 0.0 || !std::isfinite(det)) return fail(FROG_E_INVALID, "singular matrix in the chain");
            double inv[3][3] = {
                { c00 / det, (a02 * a21 - a01 * a22) / det, (a01 * a12 - a02 * a11) / det },
                { c01 / det, (a00 * a22 - a02 * a20) / det, (a02 * a10 - a00 * a12) / det },
                { c02 / det, (a01 * a20 - a00 * a21) / det, (a00 * a11 - a01 * a10) / det } };
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) t.matrix[4 * r + c] = inv[r][c];
                t.matrix[4 * r + 3] = -(inv[r][0] * m[3] + inv[r][1] * m[7] + inv[r][2] * m[11]);
            }
            t.matrix[12] = t.matrix[13] = t.matrix[14] = 0.0; t.matrix[15] = 1.0;
        } else {
            return fail(FROG_E_INVALID, "unknown transform type");
        }
        out[l] = t;
    }
    return FROG_OK;
}

extern "C++" {
namespace {

template <class S>
int reslice_typed(frog_chain *c, const frog_volume *src, frog_volume *out, int interpolation, double background)
{
    const size_t n_src = (size_t)src->dims[0] * src->dims[1] * src->dims[2], n_out = (size_t)out->dims[0] * out->dims[1] * out->dims[2];
    S *d_src = nullptr, *d_out = nullptr;
    KCHECK(hipMalloc((void **)&d_src, n_src * sizeof(S)));
    if (hipMalloc((void **)&d_out, n_out * sizeof(S)) != hipSuccess) { (void)hipFree(d_src); return fail(FROG_E_NOMEM, "hipMalloc (output volume)"); }
    hipError_t e = hipMemcpy(d_src, src->data, n_src * sizeof(S), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        reslice_kernel<S><<<(unsigned)((n_out + 255) / 256), 256>>>(
            c->d_links, (int)c->h_links.size(), d_src, (int)src->dims[0], (int)src->dims[1], (int)src->dims[2],
            src->origin[0], src->origin[1], src->origin[2], src->spacing[0], src->spacing[1], src->spacing[2],
            out->dims[0], out->dims[1], out->dims[2], out->origin[0], out->origin[1], out->origin[2],
            out->spacing[0], out->spacing[1], out->spacing[2], interpolation != 0, background, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out->data, d_out, n_out * sizeof(S), hipMemcpyDeviceToHost);
    (void)hipFree(d_src); (void)hipFree(d_out);
    if (e != hipSuccess) return fail(FROG_E_HIP, std::string("frog_chain_reslice: ") + hipGetErrorString(e));
    return FROG_OK;
}

} // namespace
} // extern "C++"

int frog_chain_reslice(frog_chain *c, const frog_volume *src, frog_volume *out, int interpolation, double background)
{
    if (!c || !src || !out || !src->data || !out->data) return fail(FROG_E_INVALID, "bad arguments to frog_chain_reslice");
    if (out->dtype != src->dtype || !frog_volume_voxel_bytes(src->dtype)) return fail(FROG_E_INVALID, "output and source scalar types must match");
    const size_t n_src = (size_t)src->dims[0] * src->dims[1] * src->dims[2], n_out = (size_t)out->dims[0] * out->dims[1] * out->dims[2];
    if (!n_src || !n_out) return fail(FROG_E_INVALID, "empty volume");
    for (int k = 0; k < 3; k++)
        if (src->dims[k] > 0x7FFFFFFFu || !(src->spacing[k] != 0.0)) return fail(FROG_E_INVALID, "bad source geometry");
    KCHECK(hipSetDevice(c->device));
    switch (src->dtype) {
    case FROG_V_U8: return reslice_typed<uint8_t>(c, src, out, interpolation, background);
    case FROG_V_I8: return reslice_typed<int8_t>(c, src, out, interpolation, background);
    case FROG_V_U16: return reslice_typed<uint16_t>(c, src, out, interpolation, background);
    case FROG_V_I16: return reslice_typed<int16_t>(c, src, out, interpolation, background);
    case FROG_V_U32: return reslice_typed<uint32_t>(c, src, out, interpolation, background);
    case FROG_V_I32: return reslice_typed<int32_t>(c, src, out, interpolation, background);
    case FROG_V_F32: return reslice_typed<float>(c, src, out, interpolation, background);
    default: return reslice_typed<double>(c, src, out, interpolation, background);
    }
}

}

// k_grid.hip.h -- point transforms (K5, K11, K12), lattice set-up helpers (K13) and
// the B-spline gradient scatter / control-point update (K7-K10) of SURVEY.md.
//
// Lattice data per owned image: coeff float4[G] (xyz + pad) and grad float4[G]
// (sum w*sDisp xyz, sum w*sWeight), control point (i,j,k) at i + dx*(j + dy*k) as
// in vtkImageData.  Points are binned once per lattice into bricks of B^3 cells
// (B = 8 or 4); a brick touches (B+3)^3 control points, which fit in LDS, so the
// 64-tap scatter of a point runs in LDS (one lane per tap, wave-private tile,
// plain read-add-write) and only the brick's sums go to HBM, as float atomics
// onto the gradient lattice gradf[image][cp] (float4: sum w*sDisp xyz, sum w*sWeight).
#pragma once

#include "ctx.h"

namespace frog {

struct GeomDev {
    int dims[3];
    int n_cp;
    double origin[3];
    double spacing[3];
    int brick;
    int nbricks[3];
    int n_bricks;
};

inline GeomDev to_dev(const GridGeom &g)
{
    GeomDev d;
    for (int k = 0; k < 3; k++) { d.dims[k] = g.dims[k]; d.origin[k] = g.origin[k]; d.spacing[k] = g.spacing[k]; d.nbricks[k] = g.nbricks[k]; }
    d.n_cp = g.n_cp; d.brick = g.brick; d.n_bricks = g.n_bricks;
    return d;
}

// imageGroup.cxx:221-232
__device__ __forceinline__ void bspline_weights(double F[4], double f)
{
    const double sixth = 1.0 / 6.0;
    const double half = 0.5;
    const double f2 = f * f;
    F[3] = f2 * f * sixth;
    F[0] = (f2 - f) * half - F[3] + sixth;
    F[2] = f + F[0] - F[3] * 2;
    F[1] = 1 - F[0] - F[2] - F[3];
}

// ---- K5: linear transform (vtkLinearTransformPoint, f64 row products -> f32) ----
__global__ __launch_bounds__(256) void transform_linear_kernel(float4 *pos, float4 *pos2, const double *mat,
                                                               uint32_t pt_begin, uint32_t pt_end, int apply)
{
    uint32_t p = pt_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pt_end) return;
    float4 v = pos[p];
    const double *M = mat + (size_t)__float_as_int(v.w) * 16;
    float4 o;
    o.x = (float)(M[0] * v.x + M[1] * v.y + M[2] * v.z + M[3]);
    o.y = (float)(M[4] * v.x + M[5] * v.y + M[6] * v.z + M[7]);
    o.z = (float)(M[8] * v.x + M[9] * v.y + M[10] * v.z + M[11]);
    o.w = v.w;
    pos2[p] = o;
    if (apply) pos[p] = o;
}

// ---- K11: cubic B-spline forward transform (vtkBSplineTransform, BorderModeZero) --
// Thread per point in brick order (perm), so a wavefront's taps fall into a few
// neighbouring cells and hit L1/L2.
__global__ __launch_bounds__(256) void transform_bspline_kernel(float4 *pos, float4 *pos2, const float4 *coeff,
                                                                const uint32_t *perm, uint32_t n_points,
                                                                uint32_t image_begin, const GeomDev g, int apply)
{
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_points) return;
    const uint32_t p = perm[s];
    const float4 v = pos[p];
    const float4 *cf = coeff + (size_t)(__float_as_int(v.w) - (int)image_begin) * g.n_cp;
    const float in[3] = { v.x, v.y, v.z };
    double F[3][4];
    int i0[3];
    #pragma unroll
    for (int k = 0; k < 3; k++) {
        double q = ((double)in[k] - g.origin[k]) / g.spacing[k];
        double fl = floor(q);
        i0[k] = (int)fl - 1;
        bspline_weights(F[k], q - fl);
    }
    double disp[3] = { 0, 0, 0 };
    const int dx = g.dims[0], dy = g.dims[1], dz = g.dims[2];
    for (int k = 0; k < 4; k++) {
        const int z = i0[2] + k;
        if (z < 0 || z >= dz) continue;
        double vz[3] = { 0, 0, 0 };
        for (int j = 0; j < 4; j++) {
            const int y = i0[1] + j;
            if (y < 0 || y >= dy) continue;
            double vy[3] = { 0, 0, 0 };
            const float4 *row = cf + (size_t)dx * ((size_t)y + (size_t)dy * z);
            #pragma unroll
            for (int i = 0; i < 4; i++) {
                const int x = i0[0] + i;
                if (x < 0 || x >= dx) continue;
                const float4 c = row[x];
                const double f = F[0][i];
                vy[0] += c.x * f; vy[1] += c.y * f; vy[2] += c.z * f;
            }
            const double f = F[1][j];
            vz[0] += vy[0] * f; vz[1] += vy[1] * f; vz[2] += vy[2] * f;
        }
        const double f = F[2][k];
        disp[0] += vz[0] * f; disp[1] += vz[1] * f; disp[2] += vz[2] * f;
    }
    float4 o;
    o.x = (float)((double)in[0] + disp[0] * 1.0);
    o.y = (float)((double)in[1] + disp[1] * 1.0);
    o.z = (float)((double)in[2] + disp[2] * 1.0);
    o.w = v.w;
    pos2[p] = o;
    if (apply) pos[p] = o;
}

// ---- K13: bounding box of the owned xyz (getBoundingBox, imageGroup.cxx:1513) ----
// doubles of floats are exact, min/max are order independent -> deterministic.
__global__ __launch_bounds__(256) void bounds_kernel(const float4 *pos, uint32_t pt_begin, uint32_t pt_end,
                                                     float *block_minmax /*[gridDim.x][6]*/)
{
    __shared__ float sh[6][256];
    float mn[3] = { 3.402823466e38f, 3.402823466e38f, 3.402823466e38f };
    float mx[3] = { -3.402823466e38f, -3.402823466e38f, -3.402823466e38f };
    for (uint32_t p = pt_begin + blockIdx.x * blockDim.x + threadIdx.x; p < pt_end; p += gridDim.x * blockDim.x) {
        float4 v = pos[p];
        mn[0] = fminf(mn[0], v.x); mn[1] = fminf(mn[1], v.y); mn[2] = fminf(mn[2], v.z);
        mx[0] = fmaxf(mx[0], v.x); mx[1] = fmaxf(mx[1], v.y); mx[2] = fmaxf(mx[2], v.z);
    }
    for (int k = 0; k < 3; k++) { sh[k][threadIdx.x] = mn[k]; sh[3 + k][threadIdx.x] = mx[k]; }
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h)
            for (int k = 0; k < 3; k++) {
                sh[k][threadIdx.x] = fminf(sh[k][threadIdx.x], sh[k][threadIdx.x + h]);
                sh[3 + k][threadIdx.x] = fmaxf(sh[3 + k][threadIdx.x], sh[3 + k][threadIdx.x + h]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) block_minmax[blockIdx.x * 6 + threadIdx.x] = sh[threadIdx.x][0];
}

// Cell of a point as the scatter computes it (imageGroup.cxx:303-310): the
// lattice coordinate is rounded to f32 before floor().
__device__ __forceinline__ void scatter_cell(const float in[3], const GeomDev &g, int ic[3], float frac[3])
{
    #pragma unroll
    for (int k = 0; k < 3; k++) {
        const float coord = (float)(((double)in[k] - g.origin[k]) / g.spacing[k]);
        const int c = (int)floorf(coord);
        ic[k] = c;
        frac[k] = coord - (float)c;
    }
}

__device__ __forceinline__ uint32_t brick_of(const int ic[3], const GeomDev &g)
{
    // cells are 1-based (origin = box min - spacing); clamp keeps stray points legal
    int b[3];
    #pragma unroll
    for (int k = 0; k < 3; k++) {
        int c = ic[k] - 1;
        c = c < 0 ? 0 : c;
        int bk = c / g.brick;
        b[k] = bk >= g.nbricks[k] ? g.nbricks[k] - 1 : bk;
    }
    return (uint32_t)(b[0] + g.nbricks[0] * (b[1] + g.nbricks[1] * b[2]));
}

// brick key count / placement (two launches around a scan).  grid = (chunks of
// BRICK_BLOCK_POINTS points, owned image): a block sees one image only, so its
// keys are that image's bricks and are first aggregated in an LDS histogram
// (integer LDS atomics are full rate); HBM sees one atomic per (block, brick)
// instead of one per point -- at the coarsest lattice an image has ONE brick.
constexpr int BRICK_BLOCK_POINTS = 1024;
constexpr int BRICK_LDS_KEYS = 8192;

__device__ __forceinline__ uint32_t point_brick(const float4 v, const GeomDev &g)
{
    const float in[3] = { v.x, v.y, v.z };
    int ic[3]; float fr[3];
    scatter_cell(in, g, ic, fr);
    return brick_of(ic, g);
}

__global__ __launch_bounds__(256) void brick_count_kernel(const float4 *pos, const uint32_t *poff,
                                                          uint32_t image_begin, const GeomDev g, uint32_t *counts)
{
    __shared__ uint32_t h[BRICK_LDS_KEYS];
    const uint32_t img = blockIdx.y;
    const uint32_t p0 = poff[image_begin + img] + blockIdx.x * BRICK_BLOCK_POINTS;
    const uint32_t pe = poff[image_begin + img + 1];
    if (p0 >= pe) return;
    const uint32_t p1 = min(p0 + (uint32_t)BRICK_BLOCK_POINTS, pe);
    const bool lds = g.n_bricks <= BRICK_LDS_KEYS;
    uint32_t *gc = counts + (size_t)img * g.n_bricks;
    if (lds) {
        for (int k = threadIdx.x; k < g.n_bricks; k += 256) h[k] = 0u;
        __syncthreads();
    }
    for (uint32_t p = p0 + threadIdx.x; p < p1; p += 256) {
        const uint32_t b = point_brick(pos[p], g);
        if (lds) atomicAdd(&h[b], 1u); else atomicAdd(&gc[b], 1u);
    }
    if (lds) {
        __syncthreads();
        for (int k = threadIdx.x; k < g.n_bricks; k += 256)
            if (h[k]) atomicAdd(&gc[k], h[k]);
    }
}

__global__ __launch_bounds__(256) void brick_place_kernel(const float4 *pos, const uint32_t *poff,
                                                          uint32_t image_begin, const GeomDev g,
                                                          uint32_t *cursor, uint32_t *perm)
{
    __shared__ uint32_t h[BRICK_LDS_KEYS];
    const uint32_t img = blockIdx.y;
    const uint32_t p0 = poff[image_begin + img] + blockIdx.x * BRICK_BLOCK_POINTS;
    const uint32_t pe = poff[image_begin + img + 1];
    if (p0 >= pe) return;
    const uint32_t p1 = min(p0 + (uint32_t)BRICK_BLOCK_POINTS, pe);
    const bool lds = g.n_bricks <= BRICK_LDS_KEYS;
    uint32_t *gcur = cursor + (size_t)img * g.n_bricks;
    constexpr int PER = BRICK_BLOCK_POINTS / 256;
    uint32_t key[PER], rank[PER];
    if (lds) {
        for (int k = threadIdx.x; k < g.n_bricks; k += 256) h[k] = 0u;
        __syncthreads();
    }
    #pragma unroll
    for (int m = 0; m < PER; m++) {
        const uint32_t p = p0 + threadIdx.x + 256 * m;
        key[m] = 0xFFFFFFFFu; rank[m] = 0;
        if (p < p1) {
            key[m] = point_brick(pos[p], g);
            if (lds) rank[m] = atomicAdd(&h[key[m]], 1u);           // rank inside this block
            else perm[atomicAdd(&gcur[key[m]], 1u)] = p;
        }
    }
    if (!lds) return;
    __syncthreads();
    for (int k = threadIdx.x; k < g.n_bricks; k += 256)
        if (h[k]) h[k] = atomicAdd(&gcur[k], h[k]);                 // block's base slot in the brick
    __syncthreads();
    #pragma unroll
    for (int m = 0; m < PER; m++) {
        const uint32_t p = p0 + threadIdx.x + 256 * m;
        if (p < p1) perm[h[key[m]] + rank[m]] = p;
    }
}

// exclusive scan of n counts -> ptr[0..n], single block; also max count
__global__ __launch_bounds__(1024) void brick_scan_kernel(const uint32_t *counts, uint32_t n, uint32_t *ptr,
                                                          uint32_t *cursor, uint32_t *max_count)
{
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t carry;
    __shared__ uint32_t mx;
    if (threadIdx.x == 0) { carry = 0; mx = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < n ? counts[i] : 0;
        atomicMax(&mx, v);
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            uint32_t t = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        uint32_t excl = sh[threadIdx.x] - v + carry;
        if (i < n) { ptr[i] = excl; cursor[i] = excl; }
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) { ptr[n] = carry; *max_count = mx; }
}

// ---- K7: scatter of the per-point sums onto the gradient lattice ----------------
// grid = (n_keys, max chunks per key); block = ONE wavefront with a private LDS
// tile of the brick's (B+3)^3 control points, float, component-major.
// Batches of 64 points.  Phase 1, lane = point: cell and the 12 cubic weights
// (f64, imageGroup.cxx:303-310) go to an LDS scratch.  Phase 2, lane = tap
// (i + 4j + 16k): w = wx[i]*wy[j]*wz[k] in f64 (imageGroup.cxx:322), then a plain
// LDS read-add-write per component: the 64 taps of a point are 64 distinct
// control points and the tile belongs to this wavefront alone, so no atomic is
// needed.  (ds_add_f32 runs ~25x slower than read+add+write on gfx950 and integer
// fixed point cannot hold the dynamic range of the tensor weights: control points
// on the shell of the cloud have gw ~ 1e-10 and still need g/gw to f32 relative
// precision -- scripts/microbench/lds_atomic.hip)
// The tile is flushed to HBM with float atomics (memory-side, ~1.3 TB/s).
constexpr int SCATTER_CHUNK = 512;
constexpr int BRICK_CP_MAX = 11;            // brick 8 -> 11^3 control points

struct ScatterScratch {
    double wts[12][64];     // [axis*4 + tap][point]
    float4 sums[64];
    int4 cell[64];          // ic xyz, w = 1 if the point contributes
};

__global__ __launch_bounds__(64) void scatter_kernel(const float4 *pos, const float4 *point_sums,
                                                     const uint32_t *perm, const uint32_t *brick_ptr,
                                                     float4 *gradf, const GeomDev g)
{
    __shared__ float tile[4 * BRICK_CP_MAX * BRICK_CP_MAX * BRICK_CP_MAX];
    __shared__ ScatterScratch sc;
    const uint32_t key = blockIdx.x;
    const uint32_t begin = brick_ptr[key] + blockIdx.y * SCATTER_CHUNK;
    const uint32_t end_all = brick_ptr[key + 1];
    if (begin >= end_all) return;
    const uint32_t end = min(begin + (uint32_t)SCATTER_CHUNK, end_all);

    const int lane = threadIdx.x;
    const int E = g.brick + 3;                  // control points per brick edge
    const int n_tile = E * E * E;
    for (int k = lane; k < n_tile * 4; k += 64) tile[k] = 0.f;

    const uint32_t img = key / g.n_bricks;
    uint32_t bidx = key - img * g.n_bricks;
    const int bx = bidx % g.nbricks[0]; bidx /= g.nbricks[0];
    const int by = bidx % g.nbricks[1];
    const int bz = bidx / g.nbricks[1];
    // first control point of the brick: cell c (1-based) uses control points c-1..c+2
    const int cp0[3] = { bx * g.brick, by * g.brick, bz * g.brick };
    float *gimg = reinterpret_cast<float *>(gradf + (size_t)img * g.n_cp);

    const int ti = lane & 3, tj = (lane >> 2) & 3, tk = lane >> 4;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t batch = begin; batch < end; batch += 64) {
        // phase 1: lane = point
        const uint32_t s = batch + lane;
        int4 cell = make_int4(0, 0, 0, 0);
        if (s < end) {
            const uint32_t p = perm[s];
            const float4 sm = point_sums[p];
            if (sm.w != 0.f) {                   // imageGroup.cxx:299
                const float4 v = pos[p];
                const float in[3] = { v.x, v.y, v.z };
                int ic[3]; float fr[3];
                scatter_cell(in, g, ic, fr);
                double F[4];
                #pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    bspline_weights(F, (double)fr[ax]);
                    #pragma unroll
                    for (int m = 0; m < 4; m++) sc.wts[ax * 4 + m][lane] = F[m];
                }
                sc.sums[lane] = sm;
                cell = make_int4(ic[0], ic[1], ic[2], 1);
            }
        }
        sc.cell[lane] = cell;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // phase 2: lane = tap
        const int n = (int)min(64u, end - batch);
        for (int q = 0; q < n; q++) {
            const int4 c = sc.cell[q];
            if (!c.w) continue;
            const double w = sc.wts[ti][q] * sc.wts[4 + tj][q] * sc.wts[8 + tk][q];
            const float4 sm = sc.sums[q];
            const float a0 = (float)(w * (double)sm.x), a1 = (float)(w * (double)sm.y);
            const float a2 = (float)(w * (double)sm.z), a3 = (float)(w * (double)sm.w);
            const int lx = c.x - 1 - cp0[0] + ti, ly = c.y - 1 - cp0[1] + tj, lz = c.z - 1 - cp0[2] + tk;
            if (lx >= 0 && ly >= 0 && lz >= 0 && lx < E && ly < E && lz < E) {
                float *dst = tile + (lx + E * (ly + E * lz));
                dst[0] += a0; dst[n_tile] += a1; dst[2 * n_tile] += a2; dst[3 * n_tile] += a3;
            } else {
                // stray point clamped into this brick (outside the scaled box): straight to HBM
                const int gx = c.x - 1 + ti, gy = c.y - 1 + tj, gz = c.z - 1 + tk;
                if (gx >= 0 && gy >= 0 && gz >= 0 && gx < g.dims[0] && gy < g.dims[1] && gz < g.dims[2]) {
                    float *dst = gimg + 4 * ((size_t)gx + (size_t)g.dims[0] * ((size_t)gy + (size_t)g.dims[1] * gz));
                    atomicAdd(dst + 0, a0); atomicAdd(dst + 1, a1); atomicAdd(dst + 2, a2); atomicAdd(dst + 3, a3);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // flush: consecutive lanes walk components, then x -> contiguous 16-byte control points
    for (int k = lane; k < n_tile * 4; k += 64) {
        const int c = k & 3;
        int q = k >> 2;
        const float val = tile[c * n_tile + q];
        if (val == 0.f) continue;
        const int lx = q % E; q /= E;
        const int ly = q % E;
        const int lz = q / E;
        const int gx = cp0[0] + lx, gy = cp0[1] + ly, gz = cp0[2] + lz;
        if (gx >= g.dims[0] || gy >= g.dims[1] || gz >= g.dims[2]) continue;
        atomicAdd(gimg + 4 * ((size_t)gx + (size_t)g.dims[0] * ((size_t)gy + (size_t)g.dims[1] * gz)) + c, val);
    }
}

// ---- K8 + first half of K9: control-point step and sum over owned images --------
// thread per control point; images in ascending order (imageGroup.cxx:346-375, :411-415)
__global__ __launch_bounds__(256) void cp_propose_kernel(const float4 *coeff, const float4 *gradf,
                                                         float4 *grad, uint32_t n_owned,
                                                         int n_cp, float alpha, double *gridsum)
{
    const int cp = blockIdx.x * blockDim.x + threadIdx.x;
    if (cp >= n_cp) return;
    double sx = 0, sy = 0, sz = 0;
    for (uint32_t i = 0; i < n_owned; i++) {
        const size_t o = (size_t)i * n_cp + cp;
        const float4 g4 = gradf[o];
        const float4 c4 = coeff[o];
        float4 n4;
        if (g4.w > 0) {
            n4.x = c4.x + alpha * g4.x / g4.w;
            n4.y = c4.y + alpha * g4.y / g4.w;
            n4.z = c4.z + alpha * g4.z / g4.w;
        } else {
            n4.x = c4.x; n4.y = c4.y; n4.z = c4.z;
        }
        n4.w = g4.w;
        grad[o] = n4;
        sx += n4.x; sy += n4.y; sz += n4.z;
    }
    gridsum[3 * (size_t)cp] = sx; gridsum[3 * (size_t)cp + 1] = sy; gridsum[3 * (size_t)cp + 2] = sz;
}

// ---- second half of K9: subtract the group mean, count oversize coefficients -----
// (imageGroup.cxx:417-428); gridsum holds the sum over ALL images.
__global__ __launch_bounds__(256) void cp_center_kernel(float4 *grad, uint32_t n_owned, int n_cp, uint32_t n_images,
                                                        const double *gridsum, double lim_x, double lim_y, double lim_z,
                                                        unsigned long long *n_big)
{
    __shared__ unsigned int sh[256];
    const int cp = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int cnt = 0;
    if (cp < n_cp) {
        const double mx = gridsum[3 * (size_t)cp] / n_images;
        const double my = gridsum[3 * (size_t)cp + 1] / n_images;
        const double mz = gridsum[3 * (size_t)cp + 2] / n_images;
        for (uint32_t i = 0; i < n_owned; i++) {
            const size_t o = (size_t)i * n_cp + cp;
            float4 v = grad[o];
            v.x = (float)((double)v.x - mx);
            v.y = (float)((double)v.y - my);
            v.z = (float)((double)v.z - mz);
            grad[o] = v;
            cnt += ((double)fabsf(v.x) > lim_x) + ((double)fabsf(v.y) > lim_y) + ((double)fabsf(v.z) > lim_z);
        }
    }
    sh[threadIdx.x] = cnt;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) sh[threadIdx.x] += sh[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0 && sh[0]) atomicAdd(n_big, (unsigned long long)sh[0]);
}

// counter -> double slot of the energy buffer (so one f64 all-reduce carries it)
__global__ void nbig_publish_kernel(const unsigned long long *n_big, double *energy)
{
    energy[2] = (double)*n_big;
}

// ---- K10: commit (imageGroup.cxx:441-468) -------------------------------------------
__global__ __launch_bounds__(256) void cp_commit_kernel(float4 *coeff, const float4 *grad, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = grad[i];
    v.w = 0.f;
    coeff[i] = v;
}

} // namespace frog

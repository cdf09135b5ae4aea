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
    double inv_spacing[3];      // 1 / spacing: the f32 form of K11 multiplies (a B-spline is continuous across a cell face)
    int brick;
    int nbricks[3];
    int n_bricks;
    // lattice layout (ctx.h GridGeom::blocked): entry of (image, node) = (node >> lat_sh) * lat_blk + image * lat_img + (node & lat_mask)
    int lat_sh;
    uint32_t lat_mask;
    size_t lat_blk, lat_img;
    uint32_t mask_words;        // > 0: the lattice keeps entries of ACTIVE (image, node) pairs only (ctx.h GridGeom::sparse): words per IMAGE
};

// Sparse lattices (ctx.h GridGeom::sparse): is (image, node) an active pair?  mask = [owned image][mask_words]: a bit per node.
__device__ __forceinline__ bool lat_active(const uint32_t *mask, const GeomDev &g, uint32_t img, uint32_t node)
{
    return (mask[(size_t)img * g.mask_words + (node >> 5)] >> (node & 31u)) & 1u;
}

// entry of control point `node` of owned image `img` in coeff / grad / gradf / grad_spare
__host__ __device__ __forceinline__ size_t lat(const GeomDev &g, uint32_t img, uint32_t node)
{
    // (one arithmetic form for both layouts, no branch: with a test on g.lat_blk in front -- uniform, a scalar branch -- the
    // thread-per-point transform no longer issued the 16 loads of a z-slab together: 2.31 -> 3.05 ms on cfg 5's level 4)
    return (size_t)(node >> g.lat_sh) * g.lat_blk + (size_t)img * g.lat_img + (size_t)(node & g.lat_mask);
}

inline GeomDev to_dev(const GridGeom &g)
{
    GeomDev d;
    if (g.blocked) { d.lat_sh = 4; d.lat_mask = 15u; d.lat_blk = (size_t)16 * g.lat_images; d.lat_img = 16; }
    else { d.lat_sh = 31; d.lat_mask = 0x7FFFFFFFu; d.lat_blk = 0; d.lat_img = (size_t)g.n_cp; }
    d.mask_words = g.sparse ? g.mask_words : 0u;
    for (int k = 0; k < 3; k++) { d.dims[k] = g.dims[k]; d.origin[k] = g.origin[k]; d.spacing[k] = g.spacing[k]; d.inv_spacing[k] = 1.0 / g.spacing[k]; d.nbricks[k] = g.nbricks[k]; }
    d.n_cp = g.n_cp; d.brick = g.brick; d.n_bricks = g.n_bricks;
    return d;
}

// imageGroup.cxx:221-232
template <typename T>
__device__ __forceinline__ void bspline_weights(T F[4], T f)
{
    const T sixth = (T)(1.0 / 6.0);
    const T half = (T)0.5;
    const T f2 = f * f;
    F[3] = f2 * f * sixth;
    F[0] = (f2 - f) * half - F[3] + sixth;
    F[2] = f + F[0] - F[3] * 2;
    F[1] = 1 - F[0] - F[2] - F[3];
}

// K11's cell and weights of one axis.  T = double: vtkBSplineTransform's arithmetic as VTK defines it (f64 quotient,
// floor, f64 weights).  T = float (the product path since round 5): the lattice coordinate still in f64 -- a point at 300 mm
// is 1e7 f32 ulps from the origin and the fraction must not lose them -- but through the reciprocal of the spacing, and
// the fraction and the four weights in f32: their 6e-8 moves a displacement of a few mm by 1e-7 mm, against an ulp of the
// result of 3e-5 mm at 300 mm.  Which side of a cell face the quotient's last bit puts a point on does not matter to a C2
// function (it does matter to the scatter, which therefore keeps the reference's f32-rounded f64 quotient: scatter_cell).
template <typename T>
__device__ __forceinline__ int bspline_axis(T F[4], float x, double origin, double spacing, double inv_spacing)
{
    double q;
    if constexpr (sizeof(T) == 8) q = ((double)x - origin) / spacing;
    else q = ((double)x - origin) * inv_spacing;
    const double fl = floor(q);
    bspline_weights<T>(F, (T)(q - fl));
    return (int)fl - 1;
}

// Four f32 lanes at a time: a stored control point is a float4 (x, y, z, pad).  Read and multiplied as a whole it is one
// ds_read_b128 (4 LDS cycles per wavefront; the three components alone are a ds_read_b96: 8 cycles) and two v_pk_fma_f32
// instead of three v_fma_f32.  The f32 form of K11 therefore carries the pad along: its sum is +0.0 when the staged pad is
// (transform_bspline_tile_kernel stages it as 0), and it is ADDED to every component at the end -- x + (+0.0f) = x for every x
// the sums can produce but -0.0 -- which keeps the compiler from narrowing the reads; the thread-per-point form adds the same
// +0.0f, so both forms have the same bits.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 fma4(f32x4 c, float f, f32x4 acc) { return __builtin_elementwise_fma(c, (f32x4)(f), acc); }
__device__ __forceinline__ f32x4 as_f32x4(const float4 v) { return f32x4{ v.x, v.y, v.z, v.w }; }

// The four scalars a deformable step hands to the host (energy sums, oversize count, list flag), written straight into
// pinned host memory by the first thread of the transform that is queued behind the step -- the kernel that starts once
// the scalars are final -- followed by the step's sequence number, which the host spins on.  A copy kernel + event between
// the lattice step and the transform cost every iteration 4 us of copy, a dependent-launch gap and the event's wake-up.
__device__ __forceinline__ void publish_step_scalars(const double *energy, double *host, double seq)
{
    if (host && blockIdx.x == 0 && threadIdx.x == 0) {
        #pragma unroll
        for (int k = 0; k < 4; k++) __hip_atomic_store(&host[k], energy[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __hip_atomic_store(&host[7], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Sharded contexts, two collectives per iteration (include/frog_hip.h frog_transform_points_slab): the transform writes the rank's rows
// straight into its slot of the slab the coordinate all-gather moves, and its first thread appends the step's four scalars as
// they stand on this rank -- energy sums, the rank's OWN oversize count, list flag -- as the slot's trailer: what used to be an
// all-reduce of its own between the step and the transform travels with the coordinates (frog_comm_unpack_slab_step adds the
// ranks' trailers up on the other side).
__device__ __forceinline__ void write_slab_trailer(const double *energy, double *trailer)
{
    if (trailer && blockIdx.x == 0 && threadIdx.x == 0) {
        #pragma unroll
        for (int k = 0; k < 4; k++) trailer[k] = energy[k];
    }
}

// ---- K5: linear transform (vtkLinearTransformPoint, f64 row products -> f32) ----
// `snap` (null: no list): the kernel also measures how far its block's points are from the culling list's snapshot and
// compares with the list's allowance, as the B-spline transforms do (k_cull.hip.h); `host_scalars`: the launch queued behind a
// linear step hands the step's scalars to the host (publish_step_scalars).
__global__ __launch_bounds__(256) void transform_linear_kernel(float4 *pos, P3 *pos2, const double *mat,
                                                               uint32_t pt_begin, uint32_t pt_end, int apply,
                                                               const P3 *snap, uint32_t *disp_part, const float *disp_allow,
                                                               uint32_t *cull_state, const double *energy, double *host_scalars, double seq,
                                                               double *trailer)
{
    publish_step_scalars(energy, host_scalars, seq);
    write_slab_trailer(energy, trailer);
    __shared__ uint32_t dmax_s[4];
    const uint32_t p = pt_begin + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t dmax = 0;
    if (p < pt_end) {
        float4 v = pos[p];
        const double *M = mat + (size_t)__float_as_int(v.w) * 16;
        float4 o;
        o.x = (float)(M[0] * v.x + M[1] * v.y + M[2] * v.z + M[3]);
        o.y = (float)(M[4] * v.x + M[5] * v.y + M[6] * v.z + M[7]);
        o.z = (float)(M[8] * v.x + M[9] * v.y + M[10] * v.z + M[11]);
        o.w = v.w;
        pos2[p] = P3{ o.x, o.y, o.z };
        if (apply) pos[p] = o;
        if (snap) {
            const P3 q = snap[p];
            const float ex = o.x - q.x, ey = o.y - q.y, ez = o.z - q.z;
            dmax = __float_as_uint(__builtin_sqrtf(ex * ex + ey * ey + ez * ez)) & 0x7FFFFFFFu;
        }
    }
    if (snap) {                                 // block-uniform
        #pragma unroll
        for (int off = 32; off > 0; off >>= 1) dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, off, 64));
        if ((threadIdx.x & 63) == 0) dmax_s[threadIdx.x >> 6] = dmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t m = max(max(dmax_s[0], dmax_s[1]), max(dmax_s[2], dmax_s[3]));
            disp_part[blockIdx.x] = m;
            if (!(__uint_as_float(m) <= *disp_allow)) atomicOr(cull_state, 1u);      // NaN included: see transform_bspline_kernel
        }
    }
}

// ---- K11: cubic B-spline forward transform (vtkBSplineTransform, BorderModeZero) --
// Thread per point in brick order (perm), so a wavefront's taps fall into a few
// neighbouring cells and hit L1/L2.
// `proposal` / `energy` (both null in ordinary calls): the launch that follows a deformable step is queued before
// the host knows whether the diffeomorphism guard accepted it (imageGroup.cxx:434-439); it reads the proposal lattice
// when the device-side oversize count says "accepted" and the standing coefficients otherwise.  The host then swaps
// the two buffers instead of copying one onto the other (the commit of :441-468 is a pointer exchange).
// T: the type the weights and the 64-tap sums are formed in (bspline_axis): float on the product path, double behind
// FROG_K11_F64=1 (the form of rounds 1-4, kept to measure the difference; the reference's bits live in k_reforder.hip.h).
// WAVES_MAX: resident wavefronts per SIMD the launch may use.  On the sparse fine lattices of a large group (cfg 5 levels 3-4: 1 to
// 7.7 GB of coefficients, bricks of 8^3 cells) every wavefront in flight is 64 x 64 scattered 16-byte reads, and six of them per
// SIMD (what the f32 form's 76 registers allow) evict each other's lines: capped at two, 0.96 -> 0.56 ms (level 3) and 2.95 ->
// 1.96 ms (level 4); the f64 form's 220 registers had capped it by accident (0.68 / 2.18).  Small contexts on dense lattices (a
// rank of eight of cfg 3: 16 us) are a few per cent faster uncapped.
template <typename T, int WAVES_MAX = 8>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, WAVES_MAX)))
void transform_bspline_kernel(float4 *pos, const float4 *pos_b, P3 *pos2, const float4 *coeff,
                                                                const uint32_t *perm, uint32_t n_points,
                                                                uint32_t image_begin, const GeomDev g, int apply,
                                                                const P3 *snap, uint32_t *disp_part,
                                                                const float4 *proposal, const double *energy, int guarantee,
                                                                const float *disp_allow, uint32_t *cull_state,
                                                                double *host_scalars, double seq, double *trailer, uint32_t by_xcd)
{
    publish_step_scalars(energy, host_scalars, seq);
    write_slab_trailer(energy, trailer);
    if (proposal && !(guarantee && energy[2] > 0.0)) coeff = proposal;
    // by_xcd = the number of 256-point blocks (0: off; the grid is then that number rounded up to a multiple of 8): XCD x =
    // blockIdx % 8 works through the x-th eighth of the points in their (image, brick, cell) order, so that the blocks resident on an
    // XCD are neighbouring bricks of a few images and their stencils meet in that XCD's L2 (section 8 row 42 of DESIGN.md)
    uint32_t vb = blockIdx.x;
    if (by_xcd) {
        const uint32_t per = (by_xcd + 7u) / 8u;
        vb = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
        if (vb >= by_xcd) return;
    }
    // every lane computes (the tail of the last block on the last point again, without storing): the displacement
    // reduction at the end is wave-wide
    const uint32_t s_raw = vb * blockDim.x + threadIdx.x;
    const bool valid = s_raw < n_points;
    const uint32_t s = valid ? s_raw : n_points - 1;
    const uint32_t p = perm[s];
    const float4 v = pos_b[s];                  // = pos[p], as the lattice's set-up gathered it (coalesced here)
    const uint32_t img = (uint32_t)(__float_as_int(v.w) - (int)image_begin);
    const float in[3] = { v.x, v.y, v.z };
    T F[3][4];
    int i0[3];
    #pragma unroll
    for (int k = 0; k < 3; k++) i0[k] = bspline_axis<T>(F[k], in[k], g.origin[k], g.spacing[k], g.inv_spacing[k]);
    T disp[3] = { 0, 0, 0 };
    const int dx = g.dims[0], dy = g.dims[1], dz = g.dims[2];
    // explicit fma below: fusing only removes one rounding before the result is rounded to f32 (VTK itself is unpinned)
    if (i0[0] >= 0 && i0[1] >= 0 && i0[2] >= 0 && i0[0] + 3 < dx && i0[1] + 3 < dy && i0[2] + 3 < dz) {
        // all 64 taps exist -- always the case for the group's own points (the lattice covers 1.2x
        // their bounding box): no per-tap test, so the 16 loads of a z-slab are issued together
        // instead of one row at a time behind a branch (the kernel waits on memory 79 % of the time)
        const uint32_t base = (uint32_t)i0[0] + (uint32_t)dx * ((uint32_t)i0[1] + (uint32_t)dy * (uint32_t)i0[2]);
        if constexpr (sizeof(T) == 4) {
            f32x4 acc = { 0.f, 0.f, 0.f, 0.f };
            #pragma unroll
            for (int k = 0; k < 4; k++) {
                float4 c[4][4];
                #pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t row = base + (uint32_t)dx * ((uint32_t)j + (uint32_t)dy * (uint32_t)k);
                    #pragma unroll
                    for (int i = 0; i < 4; i++) c[j][i] = coeff[lat(g, img, row + i)];
                }
                f32x4 vz = { 0.f, 0.f, 0.f, 0.f };
                #pragma unroll
                for (int j = 0; j < 4; j++) {
                    f32x4 vy = { 0.f, 0.f, 0.f, 0.f };
                    #pragma unroll
                    for (int i = 0; i < 4; i++) vy = fma4(as_f32x4(c[j][i]), F[0][i], vy);
                    vz = fma4(vy, F[1][j], vz);
                }
                acc = fma4(vz, F[2][k], acc);
            }
            disp[0] = acc.x + 0.0f; disp[1] = acc.y + 0.0f; disp[2] = acc.z + 0.0f;       // the tiled form's pad (fma4 above)
        } else
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            float4 c[4][4];
            #pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t row = base + (uint32_t)dx * ((uint32_t)j + (uint32_t)dy * (uint32_t)k);
                #pragma unroll
                for (int i = 0; i < 4; i++) c[j][i] = coeff[lat(g, img, row + i)];
            }
            T vz[3] = { 0, 0, 0 };
            #pragma unroll
            for (int j = 0; j < 4; j++) {
                T vy[3] = { 0, 0, 0 };
                #pragma unroll
                for (int i = 0; i < 4; i++) {
                    const T f = F[0][i];
                    vy[0] = fma((T)c[j][i].x, f, vy[0]); vy[1] = fma((T)c[j][i].y, f, vy[1]); vy[2] = fma((T)c[j][i].z, f, vy[2]);
                }
                const T f = F[1][j];
                vz[0] = fma(vy[0], f, vz[0]); vz[1] = fma(vy[1], f, vz[1]); vz[2] = fma(vy[2], f, vz[2]);
            }
            const T f = F[2][k];
            disp[0] = fma(vz[0], f, disp[0]); disp[1] = fma(vz[1], f, disp[1]); disp[2] = fma(vz[2], f, disp[2]);
        }
    } else {
        for (int k = 0; k < 4; k++) {                   // BorderModeZero: taps outside the lattice contribute nothing
            const int z = i0[2] + k;
            if (z < 0 || z >= dz) continue;
            T vz[3] = { 0, 0, 0 };
            for (int j = 0; j < 4; j++) {
                const int y = i0[1] + j;
                if (y < 0 || y >= dy) continue;
                T vy[3] = { 0, 0, 0 };
                const uint32_t row = (uint32_t)dx * ((uint32_t)y + (uint32_t)dy * (uint32_t)z);
                for (int i = 0; i < 4; i++) {
                    const int x = i0[0] + i;
                    if (x < 0 || x >= dx) continue;
                    const float4 c = coeff[lat(g, img, row + (uint32_t)x)];
                    const T f = F[0][i];
                    vy[0] = fma((T)c.x, f, vy[0]); vy[1] = fma((T)c.y, f, vy[1]); vy[2] = fma((T)c.z, f, vy[2]);
                }
                const T f = F[1][j];
                vz[0] = fma(vy[0], f, vz[0]); vz[1] = fma(vy[1], f, vz[1]); vz[2] = fma(vy[2], f, vz[2]);
            }
            const T f = F[2][k];
            disp[0] = fma(vz[0], f, disp[0]); disp[1] = fma(vz[1], f, disp[1]); disp[2] = fma(vz[2], f, disp[2]);
        }
        if constexpr (sizeof(T) == 4) { disp[0] += 0.0f; disp[1] += 0.0f; disp[2] += 0.0f; }      // as the branch above
    }
    float4 o;
    o.x = (float)((double)in[0] + (double)disp[0] * 1.0);
    o.y = (float)((double)in[1] + (double)disp[1] * 1.0);
    o.z = (float)((double)in[2] + (double)disp[2] * 1.0);
    o.w = v.w;
    if (valid) {
        pos2[p] = P3{ o.x, o.y, o.z };
        if (apply) pos[p] = o;
    }
    if (snap) {
        // largest distance of a point of this block from where it was when the outlier-culling list was built
        // (k_cull.hip.h: what cull_disp_kernel computes, here for the price of one 12-byte read): one slot per block
        __shared__ uint32_t sh[4];
        const P3 q = snap[p];
        const float dx = o.x - q.x, dy = o.y - q.y, dz = o.z - q.z;
        uint32_t m = valid ? (__float_as_uint(__builtin_sqrtf(dx * dx + dy * dy + dz * dz)) & 0x7FFFFFFFu) : 0u;
        #pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t mb = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
            disp_part[vb] = mb;
            // beyond what the culling list allows (k_cull.hip.h cull_allow_kernel; NaN compares false): the next sweep
            // walks every record and the host rebuilds the list.  Hardly ever taken: no contention.
            if (!(__uint_as_float(mb) <= *disp_allow)) atomicOr(cull_state, 1u);
        }
    }
}

// ---- K11 for a lattice that is still all zeros (every level and every regrid starts with one: setupDeformableTransforms
// is followed by transformPoints, imageGroup.cxx:79-80, :107-108): the displacement is +0.0 in every component and the
// result (float)((double) x + 0.0) = x + 0.0f -- not x: -0.0 becomes +0.0, as through the lattice.  A copy instead of 64
// taps per point (62-88 us, three to seven times per run).
__global__ __launch_bounds__(256) void transform_zero_lattice_kernel(float4 *pos, P3 *pos2, uint32_t pt_begin, uint32_t pt_end, int apply)
{
    const uint32_t p = pt_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pt_end) return;
    float4 v = pos[p];
    v.x += 0.0f; v.y += 0.0f; v.z += 0.0f;
    pos2[p] = P3{ v.x, v.y, v.z };
    if (apply) pos[p] = v;
}

// ---- K11, tiled form: one wavefront per scatter block (image, brick, <= SCATTER_CHUNK points) -----------------
// The thread-per-point form above issues 64 float4 tap loads per point (texture path busy 67 % of the kernel, 222
// registers, 2 wavefronts per SIMD).  Here the brick's (B+3)^3 coefficients are loaded ONCE per block and kept in LDS;
// a point's taps are LDS reads.  Same operations in the same order on the same f64 values as the form above: identical bits
// (tests/test_gpu_round2.py::test_tiled_transform_equals_pointwise).  A point whose f64 cell (vtkBSplineTransform floors
// the f64 lattice coordinate) is not the f32-rounded cell it was sorted by (imageGroup.cxx:303-310 rounds the coordinate
// to f32 first: they differ for points within one f32 ulp of a cell face) may need taps outside the tile and reads them
// from memory.  ScatterBlock is declared further down, with the table's construction.
struct ScatterBlock;
template <typename T>
__global__ __launch_bounds__(64) void transform_bspline_tile_kernel(float4 *pos, const float4 *pos_b, P3 *pos2, const float4 *coeff,
                                                                    const uint32_t *perm, const ScatterBlock *blocks,
                                                                    const uint32_t *n_blocks, const GeomDev g, int apply,
                                                                    const P3 *snap, uint32_t *disp_part,
                                                                    const float4 *proposal, const double *energy, int guarantee,
                                                                    const float *disp_allow, uint32_t *cull_state,
                                                                    double *host_scalars, double seq, double *trailer, uint32_t by_xcd);

// Zeroes up to ZERO_MAX device buffers in ONE launch (a lattice set-up clears six: every hipMemsetAsync is a launch
// of its own with a few microseconds of idle stream in front of it).  A block clears ZERO_BLOCK_BYTES of one buffer with
// 16-byte stores (the buffers are hipMalloc'ed: aligned); sizes are multiples of 4 bytes, the last words of a buffer go
// one by one.
constexpr int ZERO_MAX = 12;
constexpr unsigned ZERO_BLOCK_BYTES = 256 * 16 * 8;
struct ZeroList {
    uint32_t *p[ZERO_MAX];
    unsigned long long bytes[ZERO_MAX];
    unsigned first_block[ZERO_MAX + 1];     // blocks [first_block[k], first_block[k + 1]) clear buffer k
    int n;
};
__global__ __launch_bounds__(256) void zero_buffers_kernel(const ZeroList z)
{
    int k = 0;
    while (k + 1 < z.n && blockIdx.x >= z.first_block[k + 1]) k++;
    const unsigned long long begin = (unsigned long long)(blockIdx.x - z.first_block[k]) * ZERO_BLOCK_BYTES;
    const unsigned long long end = min(z.bytes[k], begin + ZERO_BLOCK_BYTES);
    unsigned char *base = reinterpret_cast<unsigned char *>(z.p[k]);
    const unsigned long long vec_end = begin + (end - begin) / 16 * 16;
    for (unsigned long long o = begin + 16ull * threadIdx.x; o < vec_end; o += 16ull * 256)
        *reinterpret_cast<uint4 *>(base + o) = make_uint4(0u, 0u, 0u, 0u);
    for (unsigned long long o = vec_end + 4ull * threadIdx.x; o < end; o += 4ull * 256)
        *reinterpret_cast<uint32_t *>(base + o) = 0u;
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

// the per-block boxes folded into one (6 floats: min xyz, max xyz); one block
__global__ __launch_bounds__(256) void bounds_final_kernel(const float *block_minmax, int n_blocks, float *out6)
{
    __shared__ float sh[6][256];
    float mn[3] = { 3.402823466e38f, 3.402823466e38f, 3.402823466e38f };
    float mx[3] = { -3.402823466e38f, -3.402823466e38f, -3.402823466e38f };
    for (int b = threadIdx.x; b < n_blocks; b += 256)
        for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], block_minmax[b * 6 + k]); mx[k] = fmaxf(mx[k], block_minmax[b * 6 + 3 + k]); }
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
    if (threadIdx.x < 6) out6[threadIdx.x] = sh[threadIdx.x][0];
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

// Sort key of a point: (image, brick, cell inside the brick), i.e. points are grouped
// by brick (the LDS tile of the scatter) and, inside a brick, by cell: consecutive
// points of one cell share all 64 tap addresses, which the scatter exploits.
// Count / placement are two launches around a scan.  grid = (chunks of
// BRICK_BLOCK_POINTS points, owned image): a block sees one image only, so when the
// image has few keys they are first aggregated in an LDS histogram (integer LDS
// atomics are full rate) and HBM sees one atomic per (block, key) instead of one per
// point -- the coarsest lattice has fewer than 100 cells per image.
constexpr int BRICK_BLOCK_POINTS = 1024;
constexpr int BRICK_LDS_KEYS = 8192;

__device__ __forceinline__ uint32_t point_key(const float4 v, const GeomDev &g)
{
    const float in[3] = { v.x, v.y, v.z };
    int ic[3]; float fr[3];
    scatter_cell(in, g, ic, fr);
    int b[3], l[3];
    #pragma unroll
    for (int k = 0; k < 3; k++) {
        int c = ic[k] - 1;                       // 0-based cell; clamp keeps stray points legal
        c = c < 0 ? 0 : c;
        int bk = c / g.brick;
        bk = bk >= g.nbricks[k] ? g.nbricks[k] - 1 : bk;
        int lc = c - bk * g.brick;
        lc = lc >= g.brick ? g.brick - 1 : lc;
        b[k] = bk; l[k] = lc;
    }
    const uint32_t brick = (uint32_t)(b[0] + g.nbricks[0] * (b[1] + g.nbricks[1] * b[2]));
    const uint32_t local = (uint32_t)(l[0] + g.brick * (l[1] + g.brick * l[2]));
    return brick * (uint32_t)(g.brick * g.brick * g.brick) + local;
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
    const int nk = g.n_bricks * g.brick * g.brick * g.brick;      // keys per image
    const bool lds = nk <= BRICK_LDS_KEYS;
    uint32_t *gc = counts + (size_t)img * nk;
    if (lds) {
        for (int k = threadIdx.x; k < nk; k += 256) h[k] = 0u;
        __syncthreads();
    }
    for (uint32_t p = p0 + threadIdx.x; p < p1; p += 256) {
        const uint32_t b = point_key(pos[p], g);
        if (lds) atomicAdd(&h[b], 1u); else atomicAdd(&gc[b], 1u);
    }
    if (lds) {
        __syncthreads();
        for (int k = threadIdx.x; k < nk; k += 256)
            if (h[k]) atomicAdd(&gc[k], h[k]);
    }
}

__global__ __launch_bounds__(256) void brick_place_kernel(const float4 *pos, const uint32_t *poff,
                                                          uint32_t image_begin, const GeomDev g,
                                                          uint32_t *cursor, uint32_t *perm, uint32_t *perm_key)
{
    __shared__ uint32_t h[BRICK_LDS_KEYS];
    const uint32_t img = blockIdx.y;
    const uint32_t p0 = poff[image_begin + img] + blockIdx.x * BRICK_BLOCK_POINTS;
    const uint32_t pe = poff[image_begin + img + 1];
    if (p0 >= pe) return;
    const uint32_t p1 = min(p0 + (uint32_t)BRICK_BLOCK_POINTS, pe);
    const int nk = g.n_bricks * g.brick * g.brick * g.brick;
    const bool lds = nk <= BRICK_LDS_KEYS;
    uint32_t *gcur = cursor + (size_t)img * nk;
    constexpr int PER = BRICK_BLOCK_POINTS / 256;
    uint32_t key[PER], rank[PER];
    if (lds) {
        for (int k = threadIdx.x; k < nk; k += 256) h[k] = 0u;
        __syncthreads();
    }
    #pragma unroll
    for (int m = 0; m < PER; m++) {
        const uint32_t p = p0 + threadIdx.x + 256 * m;
        key[m] = 0xFFFFFFFFu; rank[m] = 0;
        if (p < p1) {
            key[m] = point_key(pos[p], g);
            if (lds) rank[m] = atomicAdd(&h[key[m]], 1u);           // rank inside this block
            else { const uint32_t slot = atomicAdd(&gcur[key[m]], 1u); perm[slot] = p; perm_key[slot] = img * (uint32_t)nk + key[m]; }
        }
    }
    if (!lds) return;
    __syncthreads();
    for (int k = threadIdx.x; k < nk; k += 256)
        if (h[k]) h[k] = atomicAdd(&gcur[k], h[k]);                 // block's base slot for the key
    __syncthreads();
    #pragma unroll
    for (int m = 0; m < PER; m++) {
        const uint32_t p = p0 + threadIdx.x + 256 * m;
        if (p < p1) { const uint32_t slot = h[key[m]] + rank[m]; perm[slot] = p; perm_key[slot] = img * (uint32_t)nk + key[m]; }
    }
}

// exclusive scan of n counts -> ptr[0..n] and cursor[0..n), three launches:
// per-block sums (coalesced), scan of the block sums (one block), per-block scan + base.
constexpr int SCAN_BLOCK_ITEMS = 4096;      // 1024 threads x 4

__global__ __launch_bounds__(1024) void scan_block_sums_kernel(const uint32_t *counts, uint32_t n, uint32_t *block_sums)
{
    __shared__ uint32_t sh[1024];
    const uint32_t base = blockIdx.x * SCAN_BLOCK_ITEMS;
    uint32_t v = 0;
    #pragma unroll
    for (int m = 0; m < 4; m++) { const uint32_t i = base + threadIdx.x + 1024 * m; if (i < n) v += counts[i]; }
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int h = 512; h > 0; h >>= 1) { if ((int)threadIdx.x < h) sh[threadIdx.x] += sh[threadIdx.x + h]; __syncthreads(); }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(1024) void scan_of_sums_kernel(uint32_t *block_sums, uint32_t n_blocks, uint32_t *total)
{
    // in-place exclusive scan, chunks of 1024 with a carry
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_blocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? block_sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            uint32_t t = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_blocks) block_sums[i] = sh[threadIdx.x] - v + carry;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(1024) void scan_apply_kernel(const uint32_t *counts, uint32_t n, const uint32_t *block_base,
                                                          const uint32_t *total, uint32_t *ptr, uint32_t *cursor)
{
    __shared__ uint32_t sh[1024];
    const uint32_t base = blockIdx.x * SCAN_BLOCK_ITEMS + threadIdx.x * 4;      // 4 consecutive items per thread
    uint32_t c[4], v = 0;
    #pragma unroll
    for (int m = 0; m < 4; m++) { c[m] = base + m < n ? counts[base + m] : 0; v += c[m]; }
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t t = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = sh[threadIdx.x] - v + block_base[blockIdx.x];
    #pragma unroll
    for (int m = 0; m < 4; m++) {
        if (base + m < n) { ptr[base + m] = run; cursor[base + m] = run; }
        run += c[m];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ptr[n] = *total;
}

// The placement above hands out slots with atomics, so the order of the points INSIDE a cell changes
// from run to run -- and with it the order of the f32 additions of the scatter.  This pass puts every
// cell's points in ascending index order: thread = sorted slot, rank = number of smaller entries in its
// cell (cells hold 5 to a few hundred points; the reads hit L1; it runs once per lattice).  Per slot, not
// per cell: a fine lattice of a large group has 10^8..10^9 cells, nearly all of them empty (the first
// version launched one wavefront per cell, and past 2^32 threads the launch silently covered only part
// of them: unsorted, partly unwritten permutation, 60 % of the points of a 500-image group taken for
// strays by the scatter and sent through global atomics -- 75 ms per step instead of 0.5).
__global__ __launch_bounds__(256) void cell_order_kernel(const uint32_t *key_ptr, const uint32_t *perm_key, uint32_t n_points,
                                                         const uint32_t *perm_in, uint32_t *perm_out)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_points) return;
    const uint32_t key = perm_key[s];
    const uint32_t b = key_ptr[key], e = key_ptr[key + 1];
    const uint32_t v = perm_in[s];
    // (the lanes of a wavefront mostly share the cell: the loads are broadcasts, and what a coarse lattice's cells of
    // ~130 points cost is their number -- four entries per load)
    uint32_t rank = 0, j = b;
    for (; j < e && (j & 3u); j++) rank += perm_in[j] < v ? 1u : 0u;
    for (; j + 4u <= e; j += 4u) {
        const uint4 q = *reinterpret_cast<const uint4 *>(perm_in + j);
        rank += (q.x < v ? 1u : 0u) + (q.y < v ? 1u : 0u) + (q.z < v ? 1u : 0u) + (q.w < v ? 1u : 0u);
    }
    for (; j < e; j++) rank += perm_in[j] < v ? 1u : 0u;
    perm_out[b + rank] = v;
}

// pos in perm's order (ctx.h pos_b): once per lattice set-up
__global__ __launch_bounds__(256) void gather_positions_kernel(const float4 *pos, const uint32_t *perm, uint32_t n, float4 *pos_b)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) pos_b[s] = pos[perm[s]];
}

// ---- the scatter's block table, built on the device (no host round trip inside a lattice set-up) ----------
// A scatter block = (image, brick, run of <= SCATTER_CHUNK of the brick's points).  Brick k (image-major) holds the
// points perm[ptr[k * keys_per_brick] .. ptr[(k + 1) * keys_per_brick]); it gets ceil(count / SCATTER_CHUNK) blocks,
// whose staging slots are consecutive (lattice_step_kernel adds a brick's slots in that order).  The blocks are
// then listed longest first: a block is one wavefront whose time grows with its point count, and bricks on the rim of
// the cloud hold few points -- dispatching the long ones first shortens the tail.  The order among blocks of equal
// length comes from atomics and changes from run to run; it only decides WHEN a block runs, every block writes its own
// staging slot.
#ifndef FROG_SCATTER_CHUNK
#define FROG_SCATTER_CHUNK 384
#endif
constexpr int SCATTER_CHUNK = FROG_SCATTER_CHUNK;


struct ScatterBlock {
    uint32_t key;           // image_local * n_bricks + brick
    uint32_t begin, end;    // range in perm
    uint32_t slot;          // index of the block's tile in the staging buffer (brick-major order)
};

// chunks[k] = number of scatter blocks of brick k
// `chunk` (<= SCATTER_CHUNK): points per block of THIS context (frog_ctx::scatter_chunk) -- a context that owns few points cuts
// its bricks into shorter blocks, so that they fill the chip instead of queueing as a few long ones
__global__ void brick_chunks_kernel(const uint32_t *ptr, uint32_t n_bricks_total, uint32_t keys_per_brick, uint32_t chunk, uint32_t *chunks)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_bricks_total) return;
    const uint32_t n = ptr[(size_t)(k + 1) * keys_per_brick] - ptr[(size_t)k * keys_per_brick];
    chunks[k] = (n + chunk - 1) / chunk;
}

// blocks in brick order + histogram of their lengths (len_hist[SCATTER_CHUNK + 1], zeroed by the caller)
__global__ void block_fill_kernel(const uint32_t *ptr, const uint32_t *slot_ptr, uint32_t n_bricks_total, uint32_t keys_per_brick, uint32_t chunk,
                                  ScatterBlock *blocks, uint32_t *len_hist)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t full = 0;                  // blocks of exactly SCATTER_CHUNK points: nearly all blocks of a coarse lattice -- one
                                        // atomic per wavefront for them (5 500 on one address took 50 us of a 0.4 ms set-up)
    if (k < n_bricks_total) {
        const uint32_t b = ptr[(size_t)k * keys_per_brick], e = ptr[(size_t)(k + 1) * keys_per_brick];
        uint32_t slot = slot_ptr[k];
        const uint32_t step = chunk;
        for (uint32_t b0 = b; b0 < e; b0 += step, slot++) {
            const uint32_t e0 = min(b0 + step, e);
            blocks[slot] = ScatterBlock{ k, b0, e0, slot };
            if (e0 - b0 == chunk) full++; else atomicAdd(&len_hist[e0 - b0], 1u);
        }
    }
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) full += (uint32_t)__shfl_down((int)full, off, 64);
    if ((threadIdx.x & 63) == 0 && full) atomicAdd(&len_hist[chunk], full);
}

// first position of every length in the longest-first order; one block of SCATTER_CHUNK + 1 <= 1024 threads
__global__ void block_len_base_kernel(const uint32_t *len_hist, uint32_t *len_cursor)
{
    __shared__ uint32_t sh[SCATTER_CHUNK + 1];
    const int t = threadIdx.x;
    if (t <= SCATTER_CHUNK) sh[t] = len_hist[t];
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (int len = SCATTER_CHUNK; len >= 0; len--) { const uint32_t c = sh[len]; sh[len] = run; run += c; }
    }
    __syncthreads();
    if (t <= SCATTER_CHUNK) len_cursor[t] = sh[t];
}

__global__ void block_sort_kernel(const ScatterBlock *blocks, const uint32_t *n_blocks, uint32_t chunk, uint32_t *len_cursor, ScatterBlock *sorted)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < *n_blocks;
    ScatterBlock b{};
    if (live) b = blocks[i];
    // the full blocks of a wavefront take their places with ONE atomic (see block_fill_kernel); which full block goes where
    // among the full ones is arbitrary either way
    const bool is_full = live && b.end - b.begin == chunk;
    const unsigned long long m = __ballot(is_full);
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (m) {
        const int leader = (int)__ffsll((long long)m) - 1;
        if (lane == leader) base = atomicAdd(&len_cursor[chunk], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, leader, 64);
    }
    if (is_full) sorted[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = b;
    else if (live) sorted[atomicAdd(&len_cursor[b.end - b.begin], 1u)] = b;
}

#ifndef FROG_TRANSFORM_WAVES
#define FROG_TRANSFORM_WAVES 4
#endif
#ifndef FROG_K11_F32_UNROLL
#define FROG_K11_F32_UNROLL 1          // z-planes of the f32 form unrolled together
#endif
// Strides of the tile's rows and planes in LDS, in float4 entries (bricks of 4^3 cells: a 7 x 7 x 7 tile), packed by default.
// Round 6 built the conflict-free layout for VERDICT r5 item 6 -- rows 12 apart, planes 96: the entry of cell (cx, cy, cz) is then
// cx - 4 cy mod 16 whatever cz, so any 16 consecutive cells of the points' sort order read 16 different groups of four banks, for
// every tap -- and measured it against the packed one on one box (-DFROG_K11_TILE_SY=12 -DFROG_K11_TILE_SZ=96,
// profiles/r06_ab_experiments.txt): level 2 0.0628 against 0.0627 ms, level 0 0.0394 against 0.0364 (the tile grows from 5.5 to
// 10.3 KB: 15 resident blocks per CU instead of 16).  Bank conflicts of the tap reads are not what level 2 pays; DESIGN.md
// section 8 row 33.
#ifndef FROG_K11_TILE_SY
#define FROG_K11_TILE_SY 7
#endif
#ifndef FROG_K11_TILE_SZ
#define FROG_K11_TILE_SZ 49
#endif
constexpr int K11_SY = FROG_K11_TILE_SY, K11_SZ = FROG_K11_TILE_SZ;
constexpr int K11_TILE_ENTRIES = 6 * K11_SZ + 6 * K11_SY + 7;      // bricks of 4^3 cells (the tiled form is not used with others)
static_assert(K11_SY >= 7 && K11_SZ >= 7 * K11_SY, "tile rows / planes overlap");
template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FROG_TRANSFORM_WAVES)))
void transform_bspline_tile_kernel(float4 *pos, const float4 *pos_b, P3 *pos2, const float4 *coeff,
                                   const uint32_t *perm, const ScatterBlock *blocks,
                                   const uint32_t *n_blocks, const GeomDev g, int apply,
                                   const P3 *snap, uint32_t *disp_part,
                                   const float4 *proposal, const double *energy, int guarantee,
                                   const float *disp_allow, uint32_t *cull_state,
                                   double *host_scalars, double seq, double *trailer, uint32_t by_xcd)
{
    publish_step_scalars(energy, host_scalars, seq);
    write_slab_trailer(energy, trailer);
    // the brick's (B+3)^3 coefficients as they are in memory (f32 x, y, z, pad): one ds_read_b128 per tap.  The first
    // version kept them as three f64 arrays ("converted once"): 192 ds_read_b64 per point, and the kernel ran at the LDS's
    // bandwidth (1.5 KB per point; 81 us).  Converting each tap again costs three v_cvt per tap on a vector unit that had
    // time to spare.
    extern __shared__ float4 tile4[];
    const int lane = threadIdx.x;
    uint32_t bid = blockIdx.x;
    if (by_xcd) {
        // by_xcd = the table's capacity (0: off).  `blocks` is the table in BRICK order (image-major, bricks x-fastest: what
        // block_fill_kernel leaves) and the grid the capacity rounded up to a multiple of 8: workgroups are dealt round-robin over the 8 XCDs, so blockIdx % 8 is the XCD, and XCD x takes the x-th
        // eighth of the live entries in table order -- the wavefronts resident on an XCD at any time stage tiles of neighbouring
        // bricks of ONE image (a 7^3 tile shares three of its seven planes with each neighbour's), so the tiles' loads hit that
        // XCD's L2 instead of missing it (every XCD saw bricks of every image: 30 MB of coefficients through 4 MB of L2, 70 %
        // misses by the counters).  Entries past the count are zeroed by the workgroups that have no brick.
        const uint32_t n_live = *n_blocks, per = (n_live + 7u) / 8u;
        const uint32_t x = blockIdx.x & 7u, r = blockIdx.x >> 3;
        if (r < per) {
            bid = x * per + r;
        } else {
            const uint32_t target = 8u * per + (r - per) * 8u + x;
            if (snap && lane == 0 && target < by_xcd) disp_part[target] = 0u;
            return;
        }
        if (bid >= by_xcd) return;
    }
    // the block's entry is asked for together with the count (the table has room for the whole grid, frog_hip.hip lattice_alloc:
    // an entry past the count is stale, never out of bounds): one round trip less at the head of every block
    const ScatterBlock blk = blocks[bid];
    if (bid >= *n_blocks) {                     // the grid is an upper bound of the block count
        if (snap && lane == 0) disp_part[bid] = 0u;
        return;
    }
    if (proposal && !(guarantee && energy[2] > 0.0)) coeff = proposal;
    // the points' indices and positions run one batch ahead of the arithmetic (one wavefront per block: nothing else hides
    // the memory round trip); loads unconditional, from an index clamped into the block.  The positions come from pos_b, the
    // copy in perm's order the set-up made: coalesced and independent of the index load (they were 16-byte gathers behind
    // it: 200 MB fetched per launch for 32 MB of positions)
    const uint32_t s_last = blk.end - 1u;
    uint32_t p_cur = perm[min(blk.begin + lane, s_last)];
    uint32_t p_nxt = perm[min(blk.begin + 64 + lane, s_last)];
    float4 v_cur = pos_b[min(blk.begin + lane, s_last)];
    // bricks of 4^3 cells (the launch site checks): E and the tile's 343 entries are compile-time constants, so that the six
    // entries a lane stages are six loads IN FLIGHT TOGETHER.  With E = g.brick + 3 read at run time the loop below compiled to
    // one load, s_waitcnt vmcnt(0), ds_write_b128 per trip (seen in the ISA): six dependent memory round trips in a row at the
    // head of every block, and level 2 has 25 000 blocks of one and a quarter batches each.
    constexpr int E = 7, n_tile = E * E * E, TRIPS = (n_tile + 63) / 64;
    const uint32_t img = blk.key / g.n_bricks;
    uint32_t bidx = blk.key - img * g.n_bricks;
    const int bx = bidx % g.nbricks[0]; bidx /= g.nbricks[0];
    const int by = bidx % g.nbricks[1];
    const int bz = bidx / g.nbricks[1];
    const int cp0[3] = { bx * 4, by * 4, bz * 4 };
    const int dx = g.dims[0], dy = g.dims[1], dz = g.dims[2];
    {
        float4 staged[TRIPS];
        bool inside[TRIPS];
        #pragma unroll
        for (int t = 0; t < TRIPS; t++) {
            const int k = lane + 64 * t;
            const int tz = k / (E * E), ty = (k - tz * E * E) / E, tx = k - tz * E * E - ty * E;
            const int x = cp0[0] + tx, y = cp0[1] + ty, z = cp0[2] + tz;
            inside[t] = k < n_tile && x < dx && y < dy && z < dz;
            // unconditional, from node 0 for the entries outside (a branch around the load would bring the waits back)
            const uint32_t node = inside[t] ? (uint32_t)x + (uint32_t)dx * ((uint32_t)y + (uint32_t)dy * (uint32_t)z) : 0u;
            staged[t] = coeff[lat(g, img, node)];
        }
        #pragma unroll
        for (int t = 0; t < TRIPS; t++) {
            const int k = lane + 64 * t;
            const int tz = k / (E * E), ty = (k - tz * E * E) / E, tx = k - tz * E * E - ty * E;
            float4 c = inside[t] ? staged[t] : make_float4(0.f, 0.f, 0.f, 0.f);      // BorderModeZero: nodes outside the lattice count as 0
            c.w = 0.f;                                  // the pad the f32 form sums along with x, y, z (fma4)
            if (k < n_tile) tile4[tx + K11_SY * ty + K11_SZ * tz] = c;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    uint32_t dmax = 0;
    for (uint32_t batch = blk.begin; batch < blk.end; batch += 64) {
        const uint32_t p_far = perm[min(batch + 128 + lane, s_last)];
        const float4 v_nxt = pos_b[min(batch + 64 + lane, s_last)];
        const uint32_t p = p_cur;
        const float4 v = v_cur;
        p_cur = p_nxt; p_nxt = p_far; v_cur = v_nxt;
        if (batch + lane >= blk.end) continue;
        const float in[3] = { v.x, v.y, v.z };
        T F[3][4];
        int i0[3];
        #pragma unroll
        for (int k = 0; k < 3; k++) i0[k] = bspline_axis<T>(F[k], in[k], g.origin[k], g.spacing[k], g.inv_spacing[k]);
        const int l0 = i0[0] - cp0[0], l1 = i0[1] - cp0[1], l2 = i0[2] - cp0[2];
        T disp[3] = { 0, 0, 0 };
        if (l0 >= 0 && l1 >= 0 && l2 >= 0 && l0 + 3 < E && l1 + 3 < E && l2 + 3 < E) {
            const int base = l0 + K11_SY * l1 + K11_SZ * l2;
            if constexpr (sizeof(T) == 4) {
                const f32x4 *tile = reinterpret_cast<const f32x4 *>(tile4);
                f32x4 acc = { 0.f, 0.f, 0.f, 0.f };
                #pragma unroll FROG_K11_F32_UNROLL
                for (int k = 0; k < 4; k++) {
                    f32x4 vz = { 0.f, 0.f, 0.f, 0.f };
                    #pragma unroll
                    for (int j = 0; j < 4; j++) {
                        f32x4 vy = { 0.f, 0.f, 0.f, 0.f };
                        const int row = base + K11_SY * j + K11_SZ * k;
                        #pragma unroll
                        for (int i = 0; i < 4; i++) vy = fma4(tile[row + i], F[0][i], vy);
                        vz = fma4(vy, F[1][j], vz);
                    }
                    acc = fma4(vz, F[2][k], acc);
                }
                disp[0] = acc.x + acc.w; disp[1] = acc.y + acc.w; disp[2] = acc.z + acc.w;       // acc.w = +0.0: see fma4
            } else
            // one z-plane of 16 taps at a time: unrolled over all 64 the compiler keeps every tap in registers (232 of them:
            // two wavefronts per SIMD, and the LDS latency shows)
            #pragma unroll 1
            for (int k = 0; k < 4; k++) {
                T vz[3] = { 0, 0, 0 };
                #pragma unroll
                for (int j = 0; j < 4; j++) {
                    T vy[3] = { 0, 0, 0 };
                    const int row = base + K11_SY * j + K11_SZ * k;
                    #pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const T f = F[0][i];
                        const float4 c = tile4[row + i];
                        vy[0] = fma((T)c.x, f, vy[0]); vy[1] = fma((T)c.y, f, vy[1]); vy[2] = fma((T)c.z, f, vy[2]);
                    }
                    const T f = F[1][j];
                    vz[0] = fma(vy[0], f, vz[0]); vz[1] = fma(vy[1], f, vz[1]); vz[2] = fma(vy[2], f, vz[2]);
                }
                const T f = F[2][k];
                disp[0] = fma(vz[0], f, disp[0]); disp[1] = fma(vz[1], f, disp[1]); disp[2] = fma(vz[2], f, disp[2]);
            }
        } else {
            for (int k = 0; k < 4; k++) {               // stencil not inside the tile: from memory, border = zero
                const int z = i0[2] + k;
                if (z < 0 || z >= dz) continue;
                T vz[3] = { 0, 0, 0 };
                for (int j = 0; j < 4; j++) {
                    const int y = i0[1] + j;
                    if (y < 0 || y >= dy) continue;
                    T vy[3] = { 0, 0, 0 };
                    const uint32_t row = (uint32_t)dx * ((uint32_t)y + (uint32_t)dy * (uint32_t)z);
                    for (int i = 0; i < 4; i++) {
                        const int x = i0[0] + i;
                        if (x < 0 || x >= dx) continue;
                        const float4 c = coeff[lat(g, img, row + (uint32_t)x)];
                        const T f = F[0][i];
                        vy[0] = fma((T)c.x, f, vy[0]); vy[1] = fma((T)c.y, f, vy[1]); vy[2] = fma((T)c.z, f, vy[2]);
                    }
                    const T f = F[1][j];
                    vz[0] = fma(vy[0], f, vz[0]); vz[1] = fma(vy[1], f, vz[1]); vz[2] = fma(vy[2], f, vz[2]);
                }
                const T f = F[2][k];
                disp[0] = fma(vz[0], f, disp[0]); disp[1] = fma(vz[1], f, disp[1]); disp[2] = fma(vz[2], f, disp[2]);
            }
            if constexpr (sizeof(T) == 4) { disp[0] += 0.0f; disp[1] += 0.0f; disp[2] += 0.0f; }      // as the branch above
        }
        float4 o;
        o.x = (float)((double)in[0] + (double)disp[0] * 1.0);
        o.y = (float)((double)in[1] + (double)disp[1] * 1.0);
        o.z = (float)((double)in[2] + (double)disp[2] * 1.0);
        o.w = v.w;
        pos2[p] = P3{ o.x, o.y, o.z };
        if (apply) pos[p] = o;
        if (snap) {
            const P3 q = snap[p];
            const float ex = o.x - q.x, ey = o.y - q.y, ez = o.z - q.z;
            dmax = max(dmax, __float_as_uint(__builtin_sqrtf(ex * ex + ey * ey + ez * ez)) & 0x7FFFFFFFu);
        }
    }
    if (snap) {                                 // all 64 lanes are back together here
        #pragma unroll
        for (int off = 32; off > 0; off >>= 1) dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, off, 64));
        if (lane == 0) {
            disp_part[bid] = dmax;
            if (!(__uint_as_float(dmax) <= *disp_allow)) atomicOr(cull_state, 1u);      // see transform_bspline_kernel
        }
    }
}

// ---- K7: scatter of the per-point sums onto the gradient lattice ----------------
// One wavefront per block; block = (image, brick, run of <= SCATTER_CHUNK points of
// the brick, from the table built at set-up).  LDS: the brick's (B+3)^3 control
// points as float4 (sum w*sDisp xyz, sum w*sWeight), private to the wavefront.
//
// Batches of 64 points.  Phase 1, lane = point: the point's sums (loaded a batch ahead), cell, weights, the products
// wx wy and the tile offset of its first tap, handed over through an LDS scratch.  Phase 2, lane = tap (i + 4j + 16k),
// one point per trip: w = (wx[i] wy[j]) wz[k] (imageGroup.cxx:322) and the four products added into registers, the
// LDS tile touched only when the cell changes.  Finally the tile goes to a staging slot; lattice_step_kernel sums the
// slots in a fixed order.
//
// Why not LDS float atomics: ds_add_f32 runs at 0.33 lane-ops/clk/CU on gfx950
// (scripts/microbench/lds_atomic.hip); why not fixed point: control points on the
// shell of the cloud have total weights ~1e-10 and need f32 relative precision.
#ifdef FROG_SCATTER_TRACE
__device__ unsigned long long g_scatter_trace[4 * 65536];
#endif
constexpr int BRICK_CP_MAX = 11;            // brick 8 -> 11^3 control points
#ifndef FROG_SCATTER_FMA
#define FROG_SCATTER_FMA 1         // round 5: 0.0698 -> 0.0670 ms per launch (cfg 3), one rounding per tap as in the reference
#endif


// The energy reduction that rides on the scatter's launch (see scatter_kernel).
struct ScatterEnergy {
    const double *partial;          // [n][2] (sDistances, sWeights) per (tile, partner group); null: no reduction in this launch
    uint32_t n;
    double *block_sums;             // [ENERGY_BLOCKS][2]
    unsigned int *ticket;
    double *energy;                 // [0..1] the sums, [2] = 0 (the lattice step counts into it), [3] = the culling list is out of date
    const uint32_t *list_invalid;
    unsigned int *stray_next;       // the OTHER step's stray counter, zeroed here for the scatter after this one
    unsigned int *stray_total;      // running total of stray points
};

__device__ __forceinline__ void scatter_energy_block(const ScatterEnergy &en, const uint32_t r, const int lane)
{
    const uint32_t per = (en.n + ENERGY_BLOCKS - 1) / ENERGY_BLOCKS;
    const uint32_t b = min(en.n, r * per), e = min(en.n, b + per);
    double a0 = 0, a1 = 0;
    for (uint32_t t = b + lane; t < e; t += 64) { a0 += en.partial[2 * (size_t)t]; a1 += en.partial[2 * (size_t)t + 1]; }
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a0 += __shfl_down(a0, off, 64); a1 += __shfl_down(a1, off, 64); }
    bool last = false;
    if (lane == 0) {
        __hip_atomic_store(&en.block_sums[2 * r], a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&en.block_sums[2 * r + 1], a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(en.ticket, 1u) == (unsigned int)ENERGY_BLOCKS - 1u;
    }
    if (!__builtin_amdgcn_readfirstlane((int)last)) return;
    __threadfence();
    // the last block: the slice sums fetched side by side (two per lane), added in slice order by one lane
    const double v0 = __hip_atomic_load(&en.block_sums[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double v1 = __hip_atomic_load(&en.block_sums[64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double s0 = 0, s1 = 0;
    #pragma unroll 1
    for (int k = 0; k < ENERGY_BLOCKS; k++) {           // entries 2k and 2k+1 of the 128 values held two per lane
        const int i0 = 2 * k, i1 = 2 * k + 1;
        const double x = __shfl(i0 < 64 ? v0 : v1, i0 & 63, 64), y = __shfl(i1 < 64 ? v0 : v1, i1 & 63, 64);
        s0 += x; s1 += y;
    }
    if (lane == 0) {
        en.energy[0] = s0; en.energy[1] = s1; en.energy[2] = 0.0;
        en.energy[3] = en.list_invalid ? (double)en.list_invalid[0] : 0.0;
        *en.stray_next = 0u;
        *en.ticket = 0u;                                // ready for the next launch (same stream: ordered)
    }
}

// Per-point values handed from phase 1 (lane = point) to phase 2 (lane = tap) through LDS, one array per value
// (lane-consecutive writes and reads: no bank conflicts).  The products wx[i] * wy[j] are formed in phase 1, where one
// instruction serves 64 points; in phase 2 an instruction serves ONE point, and that loop is where a block spends its time
// (per-block trace, -DFROG_SCATTER_TRACE + scripts/microbench/scatter_trace_an.py: 63 % of a block's 52 us before this
// layout, with the three weights read separately and multiplied per point).
#ifndef FROG_SCATTER_QUADS
#define FROG_SCATTER_QUADS 1       // phase 2 reads the weights of FOUR points per LDS instruction (see there); 0: point by point (rounds 2-4)
#endif
constexpr int SC_AHEAD = 3, SC_RING = 4;        // phase 2 fetches a point's values SC_AHEAD points before it uses them
#if FROG_SCATTER_QUADS
constexpr int SC_PTS = 64 + 4;                  // ... a group of four ahead, unconditionally: one group past the batch is read (unused)
#else
constexpr int SC_PTS = 64 + SC_AHEAD;           // ... unconditionally, so up to SC_AHEAD entries past the batch are read (unused)
#endif
#ifndef FROG_SCATTER_WXY
#define FROG_SCATTER_WXY 0         // 1: the 16 products wx wy formed in phase 1 and handed over (rounds 2-4); 0: the 12 weights, multiplied in phase 2
#endif
struct ScatterScratch {
#if FROG_SCATTER_WXY
    float wxy[16][SC_PTS];      // wx[i] * wy[j] at [4 j + i]: f32 products of the f32-rounded weights (imageGroup.cxx:322, left to right)
#else
    // Round 5: the twelve weights as they are -- 68 instead of 100 bytes per point, 4.6 instead of 6.7 KB per block, which with the
    // 5.5 KB tile is what decides how many one-wavefront blocks a CU holds: 16 (the register file's limit) instead of 12.  The
    // per-block trace (scripts/microbench/scatter_trace_an.py) shows the launch as rounds of resident blocks -- 5 527 blocks of
    // 23 us in 1.8 rounds of 3 072 on level 0 -- and phase 2 is not bound by its instruction count (a branch-free form for runs of
    // one cell issued half the instructions per point and was no faster), so one more read and one more multiplication per
    // point (wx wy, the same f32 product phase 1 used to form) cost less than the fourth block per SIMD gains.
    float wx[4][SC_PTS], wy[4][SC_PTS];
#endif
    float wz[4][SC_PTS];
    float4 sm[SC_PTS];          // sDisp xyz, sWeight
    int base[SC_PTS];           // tile offset of tap (0,0,0)
};
static_assert(offsetof(ScatterScratch, sm) % 16 == 0, "ScatterScratch::sm is read with ds_read_b128");

// run += w * (the four values `s` holds in lane (t & ~3) + P): four v_fmac_f32 with quad_perm [P, P, P, P] on their first source -- the
// broadcast inside the quad is a modifier of the multiply-add, not an instruction.  Written out because the compiler, given a builtin
// DPP move and four multiply-adds, forms two v_pk_fma_f32 (which cannot take the modifier) behind four v_mov_b32_dpp: six
// instructions for four.  ONE asm statement per point, led by s_nop 1: a DPP source written by a vector instruction needs two wait
// states before it is read, the compiler's hazard recogniser cannot see inside an asm statement, and nothing can be scheduled
// into the middle of one.
#define FROG_FMAC_DPP(D, S, Q) "v_fmac_f32_dpp %" #D ", %" #S ", %8 quad_perm:[" #Q "," #Q "," #Q "," #Q "] row_mask:0xf bank_mask:0xf\n\t"
// (Round 6, ADVICE r5: the statement is `volatile` now, and tests/test_gpu_round6.py holds this form against a build without it
// bit for bit.  s_nop 4 -- which would also cover a vector write of EXEC, v_cmpx, right in front of the statement; the compiler
// forms its masks with s_and_saveexec here -- was measured: scatter 0.0645 against 0.0629 ms, 650 steps 2 212 against 2 232 it/s;
// -DFROG_DPP_NOP='"s_nop 4"' builds it.)
#ifndef FROG_DPP_NOP
#define FROG_DPP_NOP "s_nop 1"
#endif
#define FROG_FMAC4_DPP(Q) asm volatile(FROG_DPP_NOP "\n\t" FROG_FMAC_DPP(0, 4, Q) FROG_FMAC_DPP(1, 5, Q) FROG_FMAC_DPP(2, 6, Q) FROG_FMAC_DPP(3, 7, Q)      \
                              : "+v"(run.x), "+v"(run.y), "+v"(run.z), "+v"(run.w) : "v"(s.x), "v"(s.y), "v"(s.z), "v"(s.w), "v"(w))
template <int P> __device__ __forceinline__ void fmac4_quad_bcast(float4 &run, const float4 s, float w)
{
    static_assert(P >= 0 && P < 4, "quad lane");
    if constexpr (P == 0) FROG_FMAC4_DPP(0);
    if constexpr (P == 1) FROG_FMAC4_DPP(1);
    if constexpr (P == 2) FROG_FMAC4_DPP(2);
    if constexpr (P == 3) FROG_FMAC4_DPP(3);
}

__global__ __launch_bounds__(64) void scatter_kernel(const float4 *pos_b, const float4 *point_sums,
                                                     const float4 *group_sums, uint32_t own_points, uint32_t own_pt_begin,
                                                     const uint32_t *perm, const ScatterBlock *blocks, const uint32_t *n_blocks,
                                                     float4 *gradf, float4 *stage, unsigned int *stray, const GeomDev g,
                                                     const ScatterEnergy en)
{
    // The first ENERGY_BLOCKS blocks of the grid add up the sweep's (sDistances, sWeights) tile partials instead -- a launch
    // of its own cost 7 us + two gaps of every iteration, and nothing in this kernel waits for the result.  Fixed slices,
    // fixed tree, the 64 slice sums added in order by whichever block finishes last: deterministic.
    if (en.partial && blockIdx.x < (uint32_t)ENERGY_BLOCKS) {
        scatter_energy_block(en, blockIdx.x, threadIdx.x);
        return;
    }
    const uint32_t bid = blockIdx.x - (en.partial ? (uint32_t)ENERGY_BLOCKS : 0u);
#ifdef FROG_SCATTER_TRACE
    const unsigned long long trace_t0 = wall_clock64();
    unsigned tr_load = 0, tr_p1 = 0, tr_p2 = 0;
#endif
    // the grid is an upper bound (the block table is built on the device and its length never visits the host); the block's
    // entry is asked for together with the count -- the table has room for the whole grid, an entry past the count is stale,
    // never out of bounds -- which takes one round trip off the chain count -> entry -> index -> sums at the head of every block
    const ScatterBlock blk = blocks[bid];
    if (bid >= *n_blocks) return;
    // the brick's (B+3)^3 control points: sized at launch ((B+3)^3 * 16 bytes), so that bricks of 4^3 cells
    // take 5.4 KB instead of the 21 KB of the largest brick
    extern __shared__ float4 tile[];
    __shared__ ScatterScratch sc;
    const int lane = threadIdx.x;
    const int E = g.brick + 3;                  // control points per brick edge
    const int n_tile = E * E * E;

    // The loads run ahead of the arithmetic: a block is ONE wavefront, and perm -> (sums, position) is a chain of two
    // memory round trips per batch that nothing else on the wavefront hides (11 of a block's 38 us were spent waiting for
    // it).  The index of batch b + 2 and the values of batch b + 1 are requested before batch b is worked on; every load is
    // unconditional, from an index clamped into the block (see k_links.hip.h on loads under lane-dependent branches).
    const uint32_t s_last = blk.end - 1u;
    const auto load_index = [&](uint32_t s) -> uint32_t { return perm[min(s, s_last)]; };
    const auto load_point = [&](uint32_t p, uint32_t s, float4 (&part)[N_XCD], float4 &v) __attribute__((always_inline)) {
        if (group_sums) {                        // kernel-uniform
            const uint32_t li = p - own_pt_begin;
            #pragma unroll
            for (int q = 0; q < N_XCD; q++) part[q] = group_sums[group_sum_index(q, li, own_points)];
        } else {
            part[0] = point_sums[p];
        }
        v = pos_b[min(s, s_last)];                // the position, in perm's order (the set-up's copy: coalesced)
    };
    uint32_t p_cur = load_index(blk.begin + lane);
    uint32_t p_nxt = load_index(blk.begin + 64 + lane);
    float4 part_cur[N_XCD], v_cur;
    #pragma unroll
    for (int q = 1; q < N_XCD; q++) part_cur[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    load_point(p_cur, blk.begin + lane, part_cur, v_cur);

    for (int k = lane; k < n_tile; k += 64) tile[k] = make_float4(0.f, 0.f, 0.f, 0.f);

    const uint32_t img = blk.key / g.n_bricks;
    uint32_t bidx = blk.key - img * g.n_bricks;
    const int bx = bidx % g.nbricks[0]; bidx /= g.nbricks[0];
    const int by = bidx % g.nbricks[1];
    const int bz = bidx / g.nbricks[1];
    // first control point of the brick: cell c (1-based) uses control points c-1..c+2
    const int cp0[3] = { bx * g.brick, by * g.brick, bz * g.brick };

    const int ti = lane & 3, tj = (lane >> 2) & 3, tk = lane >> 4;
    const int tap_off = ti + E * (tj + E * tk);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    float4 run = make_float4(0.f, 0.f, 0.f, 0.f);      // contributions of the current cell, this lane's tap
    int run_base = -1;
    for (uint32_t batch = blk.begin; batch < blk.end; batch += 64) {
        const uint32_t p_far = load_index(batch + 128 + lane);
        float4 part_nxt[N_XCD], v_nxt;
        #pragma unroll
        for (int q = 1; q < N_XCD; q++) part_nxt[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        load_point(p_nxt, batch + 64 + lane, part_nxt, v_nxt);
#ifdef FROG_SCATTER_TRACE
        const unsigned long long tb0 = wall_clock64();
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");        // the current batch's loads are the oldest
        const unsigned long long tb1 = wall_clock64();
#endif
        // ---- phase 1: lane = point.  Cell, the 12 cubic weights in f64 as the reference computes them
        // (imageGroup.cxx:303-310), rounded once to f32, the 16 products wx wy, the tile offset of the first tap.
        float wxy[16], wz[4], wx4[4], wy4[4];
        float4 sm = part_cur[0];
        int base = -1;
        if (batch + lane < blk.end) {
            if (group_sums) {
                // the point's sums from the N_XCD partial sums of the sweep, added in group order exactly as
                // combine_groups_kernel does (same bits): saves that kernel's pass over 288 MB
                #pragma unroll
                for (int q = 1; q < N_XCD; q++) { sm.x += part_cur[q].x; sm.y += part_cur[q].y; sm.z += part_cur[q].z; sm.w += part_cur[q].w; }
            }
            if (sm.w != 0.f) {                       // imageGroup.cxx:299
                const float in[3] = { v_cur.x, v_cur.y, v_cur.z };
                int ic[3]; float fr[3];
                scatter_cell(in, g, ic, fr);
                double F[4];
                float w12[12];
                #pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    bspline_weights(F, (double)fr[ax]);
                    #pragma unroll
                    for (int m = 0; m < 4; m++) w12[ax * 4 + m] = (float)F[m];
                }
                #pragma unroll
                for (int j = 0; j < 4; j++)
                    #pragma unroll
                    for (int i = 0; i < 4; i++) wxy[4 * j + i] = w12[i] * w12[4 + j];
                #pragma unroll
                for (int k = 0; k < 4; k++) { wz[k] = w12[8 + k]; wx4[k] = w12[k]; wy4[k] = w12[4 + k]; }
                const int lx = ic[0] - 1 - cp0[0], ly = ic[1] - 1 - cp0[1], lz = ic[2] - 1 - cp0[2];
                if (lx >= 0 && ly >= 0 && lz >= 0 && lx + 3 < E && ly + 3 < E && lz + 3 < E) {
                    base = lx + E * (ly + E * lz);
                } else {
                    // stray point clamped into this brick (outside the scaled box): its taps go
                    // straight to HBM, one lane doing all 64; lattice_step_kernel then folds the gradient lattice in
                    atomicAdd(stray, 1u);
                    atomicAdd(en.stray_total, 1u);  // running total, never cleared (frog_test_stray_points)
                    for (int k = 0; k < 4; k++) for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) {
                        const int gx = ic[0] - 1 + i, gy = ic[1] - 1 + j, gz = ic[2] - 1 + k;
                        if (gx < 0 || gy < 0 || gz < 0 || gx >= g.dims[0] || gy >= g.dims[1] || gz >= g.dims[2]) continue;
                        const float w = wxy[4 * j + i] * wz[k];
                        float *dst = reinterpret_cast<float *>(gradf + lat(g, img, (uint32_t)gx + (uint32_t)g.dims[0] * ((uint32_t)gy + (uint32_t)g.dims[1] * (uint32_t)gz)));
                        atomicAdd(dst + 0, w * sm.x); atomicAdd(dst + 1, w * sm.y);
                        atomicAdd(dst + 2, w * sm.z); atomicAdd(dst + 3, w * sm.w);
                    }
                }
            }
        }
        // the contributing points go to the front of the scratch (order kept)
        const unsigned long long live = __ballot(base >= 0);
        const int n_live = __popcll(live);
        if (base >= 0) {
            const int slot = __popcll(live & ((1ull << lane) - 1ull));
#if FROG_SCATTER_WXY
            #pragma unroll
            for (int k = 0; k < 16; k++) sc.wxy[k][slot] = wxy[k];
#else
            #pragma unroll
            for (int k = 0; k < 4; k++) { sc.wx[k][slot] = wx4[k]; sc.wy[k][slot] = wy4[k]; }
#endif
            #pragma unroll
            for (int k = 0; k < 4; k++) sc.wz[k][slot] = wz[k];
            sc.sm[slot] = sm;
            sc.base[slot] = base;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#ifdef FROG_SCATTER_TRACE
        const unsigned long long tb2 = wall_clock64();
#endif

        // ---- phase 2: lane = tap (i + 4 j + 16 k); per point two lane-constant reads (wxy[i + 4 j], wz[k]) and one
        // broadcast (the sums), w = (wx wy) wz (imageGroup.cxx:322) and the four products added into REGISTERS: points
        // are sorted by cell, so consecutive points mostly share their 64 tap addresses and the LDS tile is touched (one
        // ds_read_b128 + ds_write_b128, no atomic: the 64 taps are 64 distinct control points and the tile is private)
        // only when the cell changes -- bit q of `chg`, known before the loop.
        unsigned long long chg;
        {
            const int mine = lane < n_live ? sc.base[lane] : -1;
            const int before = lane == 0 ? run_base : (lane < n_live ? sc.base[lane - 1] : -1);
            chg = __ballot(lane < n_live && mine != before);
        }
#if FROG_SCATTER_QUADS && !FROG_SCATTER_WXY
        // Round 5.  The launch is bound by the LDS: point by point a tap's three weights are three ds_read_b32 (2 LDS cycles
        // each for the wavefront, however many lanes share an address) and the sums a broadcast ds_read_b128 (4): 10 cycles per
        // point and CU, 7 800 points per CU -> 33 us of a 64 us launch (level 0; on level 2 the tile's read-add-write at nearly
        // every point adds 17 more).  A ds_read_b128 costs 4 cycles and brings the weights of FOUR consecutive points
        // (the scratch is point-minor), and the four points' sums arrive with ONE more: lane t reads the sums of point q0 + (t & 3)
        // and the quad hands them round by DPP, a modifier of the multiply-add (fmac4_quad_bcast), not an instruction: 16 cycles per four points.
        // Same operations on the same values in the same order as point by point: identical bits.
        {
            const f32x4 *gwx = reinterpret_cast<const f32x4 *>(sc.wx[lane & 3]), *gwy = reinterpret_cast<const f32x4 *>(sc.wy[(lane >> 2) & 3]),
                        *gwz = reinterpret_cast<const f32x4 *>(sc.wz[lane >> 4]);
            const float4 *gsm = sc.sm + (lane & 3);
            f32x4 cwx = gwx[0], cwy = gwy[0], cwz = gwz[0];
            float4 csm4 = gsm[0];
            auto spill_and_restart = [&](int q) __attribute__((always_inline)) {
                if (run_base >= 0) {
                    float4 t = tile[run_base + tap_off];
                    t.x += run.x; t.y += run.y; t.z += run.z; t.w += run.w;
                    tile[run_base + tap_off] = t;
                }
                run = make_float4(0.f, 0.f, 0.f, 0.f);
                run_base = __builtin_amdgcn_readfirstlane(sc.base[q]);
            };
            for (int q0 = 0; q0 < n_live; q0 += 4) {
                const int g1 = (q0 >> 2) + 1;
                const f32x4 nwx = gwx[g1], nwy = gwy[g1], nwz = gwz[g1];       // the next group's, one trip ahead (unconditional)
                const float4 nsm = gsm[4 * g1];
                const f32x4 w4 = (cwx * cwy) * cwz;                              // (wx wy) wz, imageGroup.cxx:322 left to right
                const unsigned m = (unsigned)(chg >> q0) & 15u;
                const int left = n_live - q0;                                    // >= 1
#define FROG_SCATTER_ADD(P)                                                                                            \
                if (P < left) {                                                                                         \
                    if (m >> P & 1u) spill_and_restart(q0 + P);                                                         \
                    fmac4_quad_bcast<P>(run, csm4, w4[P]);                                                              \
                }
                FROG_SCATTER_ADD(0) FROG_SCATTER_ADD(1) FROG_SCATTER_ADD(2) FROG_SCATTER_ADD(3)
#undef FROG_SCATTER_ADD
                cwx = nwx; cwy = nwy; cwz = nwz; csm4 = nsm;
            }
        }
#else
        float rwxy[SC_RING], rwz[SC_RING];
        float4 rsm[SC_RING];
#if FROG_SCATTER_WXY
        const float *lane_wxy = sc.wxy[lane & 15], *lane_wz = sc.wz[lane >> 4];
        auto fetch = [&](int q, int slot) __attribute__((always_inline)) {
            rwxy[slot] = lane_wxy[q];
            rwz[slot] = lane_wz[q];
            rsm[slot] = sc.sm[q];
        };
#else
        float rwy[SC_RING];
        const float *lane_wx = sc.wx[lane & 3], *lane_wy = sc.wy[(lane >> 2) & 3], *lane_wz = sc.wz[lane >> 4];
        auto fetch = [&](int q, int slot) __attribute__((always_inline)) {
            rwxy[slot] = lane_wx[q];
            rwy[slot] = lane_wy[q];
            rwz[slot] = lane_wz[q];
            rsm[slot] = sc.sm[q];
        };
#endif
        auto add_point = [&](int q, int slot) __attribute__((always_inline)) {
            if ((chg >> q) & 1ull) {                               // wave-uniform: cell changed -> spill the run
                if (run_base >= 0) {
                    float4 t = tile[run_base + tap_off];
                    t.x += run.x; t.y += run.y; t.z += run.z; t.w += run.w;
                    tile[run_base + tap_off] = t;
                }
                run = make_float4(0.f, 0.f, 0.f, 0.f);
                run_base = __builtin_amdgcn_readfirstlane(sc.base[q]);
            }
#if FROG_SCATTER_WXY
            const float w = rwxy[slot] * rwz[slot];
#else
            const float w = (rwxy[slot] * rwy[slot]) * rwz[slot];          // (wx wy) wz, imageGroup.cxx:322 left to right
#endif
            const float4 csm = rsm[slot];
#if FROG_SCATTER_FMA
            // g += w s with ONE rounding: what imageGroup.cxx:330-337 computes ((float)((double) g + w * s): the product of two
            // f32 values is exact in f64), and two v_pk_fma_f32 instead of two v_pk_mul_f32 + two v_pk_add_f32
            run.x = fmaf(w, csm.x, run.x); run.y = fmaf(w, csm.y, run.y); run.z = fmaf(w, csm.z, run.z); run.w = fmaf(w, csm.w, run.w);
#else
            run.x += w * csm.x; run.y += w * csm.y; run.z += w * csm.z; run.w += w * csm.w;
#endif
        };
        #pragma unroll
        for (int j = 0; j < SC_AHEAD; j++) fetch(j, j);
        const int n_full = n_live & ~(SC_RING - 1);
        for (int q0 = 0; q0 < n_full; q0 += SC_RING) {
            #pragma unroll
            for (int j = 0; j < SC_RING; j++) {
                fetch(q0 + j + SC_AHEAD, (j + SC_AHEAD) % SC_RING);
                add_point(q0 + j, j);
            }
        }
        #pragma unroll
        for (int j = 0; j < SC_RING - 1; j++)                       // the last n_live % 4 points: fetched by the last full trip
            if (n_full + j < n_live) add_point(n_full + j, j);
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#ifdef FROG_SCATTER_TRACE
        {
            const unsigned long long tb3 = wall_clock64();
            tr_load += __builtin_amdgcn_readfirstlane((unsigned)(tb1 - tb0));
            tr_p1 += __builtin_amdgcn_readfirstlane((unsigned)(tb2 - tb1));
            tr_p2 += __builtin_amdgcn_readfirstlane((unsigned)(tb3 - tb2));
        }
#endif
        p_cur = p_nxt; p_nxt = p_far; v_cur = v_nxt;
        #pragma unroll
        for (int q = 0; q < N_XCD; q++) part_cur[q] = part_nxt[q];
    }
    if (run_base >= 0) {
        float4 t = tile[run_base + tap_off];
        t.x += run.x; t.y += run.y; t.z += run.z; t.w += run.w;
        tile[run_base + tap_off] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // the tile goes to the block's slot of the staging buffer as it is; lattice_step_kernel adds the
    // slots that cover a control point in a fixed order (a flush with float atomics is as fast, but its
    // order changes from run to run, and one ulp in a coefficient can move a half-link across the
    // inlier threshold a few iterations later)
    float4 *dst = stage + (size_t)blk.slot * n_tile;
    for (int k = lane; k < n_tile; k += 64) dst[k] = tile[k];
#ifdef FROG_SCATTER_TRACE
    if (lane == 0 && bid < 65536) {
        unsigned int hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_scatter_trace[4 * bid + 0] = trace_t0;
        g_scatter_trace[4 * bid + 1] = wall_clock64();
        g_scatter_trace[4 * bid + 2] = ((unsigned long long)xcc << 32) | hw;
        g_scatter_trace[4 * bid + 3] = (unsigned long long)(blk.end - blk.begin) | ((unsigned long long)(tr_load & 0xFFFF) << 16)
                                              | ((unsigned long long)(tr_p1 & 0xFFFF) << 32) | ((unsigned long long)(tr_p2 & 0xFFFF) << 48);
    }
#endif
}

// ---- K7 (flush) + K8 + K9 in one launch ---------------------------------------------------------------------
// lattice_step_kernel<CENTER>: for a tile of LS_CPB control points (consecutive in x) and ALL owned images
//   1. gradient g, gw = sum of the staged scatter tiles that cover the control point, in a fixed order: bricks in
//      (z, y, x) order, the blocks of a brick in point order (+ what stray points added with atomics, if any did);
//   2. proposal  c + alpha g / gw  (gw > 0, else c), f32 left to right (imageGroup.cxx:346-375) -> grad (w = gw);
//   3. sum over the images in ASCENDING order, f64, as imageGroup.cxx:411-415 -> gridsum;
//   CENTER (a context that owns the whole group, so the sum is the group's):
//   4. subtract sum / nImages from every image's proposal ((float)((double) v - mean), :417-423; nothing when images
//      are fixed, :398), count coefficients beyond maxDisplacementRatio * spacing (:424-428) -> energy[2].
// One launch instead of lattice_reduce + cp_propose + cp_center (77 us -> see DESIGN.md), and the gradient lattice is
// never written to memory.  Images are taken LS_IC at a time: thread = (image, control point) computes steps 1-2 and
// parks the proposal in LDS, then one thread per (control point, axis) adds the LS_IC values in image order to its
// running f64 sum -- the reference's order exactly.  Step 4 re-reads the proposals this block has just written (L2).
// Without CENTER the launch stops after step 3: the host all-reduces gridsum over the ranks and cp_center_kernel does 4.
constexpr int LS_CPB = 16;          // control points per block
constexpr int LS_IC = 64;           // images per pass (with 16 a group of 100 images took 7 dependent rounds of
                                    // slot-pointer -> tile loads per block: 130 us where the three kernels took 77)
constexpr int LS_THREADS = LS_CPB * LS_IC;
constexpr int LS_CPB_SMALL = 4;      // block shape for lattices below LS_SMALL_NODES control points: 4 x 64 = 256 threads
constexpr int LS_SMALL_NODES = 2048;     // measured on cfg 3: 704 nodes 35.7 -> 31.8 us with the narrow shape, 3 042 nodes 28.8 -> 34.6 us (slower)
constexpr int LS_KEEP = 8;          // passes whose proposals stay in registers until the mean is known
static_assert(LS_THREADS <= 1024 && 3 * LS_CPB <= LS_THREADS, "lattice_step_kernel thread mapping");

struct LatticeStepArgs {
    const float4 *stage;            // staged scatter tiles
    const uint32_t *brick_slot_ptr;
    float4 *gradf;                  // only read (and cleared) when *stray != 0
    unsigned int *stray;            // number of stray points of the last scatter
    const float4 *coeff;
    float4 *grad;
    double *gridsum;
    double *energy_tail;            // null, or gridsum + 3 G: the step's energy sums and list flag ride on the all-reduce of the proposal sums
    uint32_t n_owned, n_images;     // n_images = 0: fixed images present, no mean removal
    float alpha;
    double lim[3];
    double *energy;
    // sparse lattices: the active pairs' bit sets, the value every inactive pair of a node holds (standing / proposed)
    const uint32_t *mask;
    const uint32_t *n_inactive;     // [G] owned images for which the node is inactive
    const float4 *ucoeff;
    float4 *ugrad;
};

// CPB: control points per block (LS_CPB, or LS_CPB_SMALL for lattices of so few nodes that blocks of LS_CPB would leave most
// of the chip idle: 704 nodes of a coarse level are 44 blocks of 16 -- 17 % of the CUs, each waiting on its own loads).
template <bool CENTER, int CPB = LS_CPB>
__global__ __launch_bounds__(CPB * LS_IC) void lattice_step_kernel(const LatticeStepArgs a, const GeomDev g)
{
    __shared__ float prop[LS_IC][CPB][3];
    __shared__ double mean[CPB][3];
    const int tid = threadIdx.x;
    const int c = tid % CPB, il = tid / CPB;
    const int cp = blockIdx.x * CPB + c;
    const bool has_stray = *a.stray != 0u;

    // bricks that cover this control point (brick b holds control points b*B .. b*B + B + 2)
    const int B = g.brick, E = B + 3, n_tile = E * E * E;
    int cc[3] = { 0, 0, 0 }, lo[3] = { 0, 0, 0 }, hi[3] = { -1, -1, -1 };
    if (cp < g.n_cp) {
        cc[0] = cp % g.dims[0]; cc[1] = (cp / g.dims[0]) % g.dims[1]; cc[2] = cp / (g.dims[0] * g.dims[1]);
        #pragma unroll
        for (int k = 0; k < 3; k++) {
            lo[k] = cc[k] <= 2 ? 0 : max(0, (cc[k] - 2 + B - 1) / B - 1);
            hi[k] = min(g.nbricks[k] - 1, cc[k] / B);
        }
    }
    // summation thread: (control point sc, axis sa)
    const int sc = tid % CPB, sa = tid / CPB;
    double run = 0.0;
    // CENTER: the proposals of up to LS_KEEP passes (groups of up to 512 images) wait in registers for the mean, so that
    // every coefficient is written once; larger groups write the raw proposals and re-read them (the blocks' own L2 lines)
    float4 keep[LS_KEEP];
    const bool kept = CENTER && a.n_owned <= (uint32_t)(LS_KEEP * LS_IC);

    // Sparse lattices: which of this thread's pairs are active, for all the passes it keeps in registers -- asked for up front, side
    // by side: behind a load of its own at the head of every pass's chain the passes ran one after the other (a wait in front of
    // each pass's branch; cfg 5 level 4: a third of the bytes gone and a tenth of the time)
    uint32_t act = ~0u;
    float4 u_node = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.mask_words && cp < g.n_cp) {
        u_node = a.ucoeff[cp];
        if (kept) {
            act = 0u;
            #pragma unroll
            for (int pass = 0; pass < LS_KEEP; pass++) {
                const uint32_t img = (uint32_t)pass * LS_IC + il;
                if (img < a.n_owned) act |= (lat_active(a.mask, g, img, (uint32_t)cp) ? 1u : 0u) << pass;
            }
        }
    }
    // steps 1-2 for this thread's (image i0 + il, control point): the proposal (0 for threads without one)
    auto propose = [&](uint32_t i0, bool store) {
        const uint32_t img = i0 + il;
        float4 n4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (img < a.n_owned && cp < g.n_cp && g.mask_words && !(kept ? (act >> (i0 / LS_IC)) & 1u : (uint32_t)lat_active(a.mask, g, img, (uint32_t)cp))) {
            // no point of this image reaches the node while the lattice stands: gw = 0, the proposal is the standing value -- the
            // one every such pair of the node holds (they started at 0 together and only ever had the node's mean subtracted)
            n4 = u_node;
            n4.w = -1.0f;                               // marks the pair for centre(): nothing of it is stored
        } else if (img < a.n_owned && cp < g.n_cp) {
            const size_t o = lat(g, img, (uint32_t)cp);
            const float4 c4 = a.coeff[o];                       // needed last, asked for first
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            // (a per-image box of the non-empty bricks used to trim the candidates: one more memory round trip in front of
            // the slot ranges, for look-ups that an empty range answers just as well -- slower on every lattice measured)
            const int x0 = lo[0], y0 = lo[1], z0 = lo[2], x1 = hi[0], y1 = hi[1], z1 = hi[2];
            // At most two bricks per axis cover a node (E = B + 3 <= 2 B).  The eight candidates in (z, y, x) order; their
            // slot ranges, then the first tile of each, are fetched side by side (this used to be a chain of dependent
            // loads per candidate: the kernel spent its time waiting), the sums are then formed in the fixed order.
            uint32_t kb[8], ke[8];
            int loc[8];
            #pragma unroll
            for (int n = 0; n < 8; n++) {
                const int bx = x0 + (n & 1), by = y0 + ((n >> 1) & 1), bz = z0 + (n >> 2);
                const bool in = bx <= x1 && by <= y1 && bz <= z1;
                const uint32_t key = img * g.n_bricks + (uint32_t)(bx + g.nbricks[0] * (by + g.nbricks[1] * bz));
                kb[n] = in ? a.brick_slot_ptr[key] : 0u;
                ke[n] = in ? a.brick_slot_ptr[key + 1] : 0u;
                loc[n] = (cc[0] - bx * B) + E * ((cc[1] - by * B) + E * (cc[2] - bz * B));
            }
            float4 first[8];
            #pragma unroll
            for (int n = 0; n < 8; n++)
                first[n] = kb[n] < ke[n] ? a.stage[(size_t)kb[n] * n_tile + loc[n]] : make_float4(0.f, 0.f, 0.f, 0.f);
            #pragma unroll
            for (int n = 0; n < 8; n++) {
                if (kb[n] < ke[n]) { s.x += first[n].x; s.y += first[n].y; s.z += first[n].z; s.w += first[n].w; }
                // bricks of more than SCATTER_CHUNK points: their further slots four at a time, the loads together (from a
                // slot index clamped into the brick's range) and then the adds in slot order -- one load per trip made this
                // a chain of up to 32 memory round trips per pass on the coarsest lattice (24 of a block's 29 us)
                for (uint32_t sl = kb[n] + 1; sl < ke[n]; sl += 4) {
                    float4 v[4];
                    #pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = a.stage[(size_t)min(sl + j, ke[n] - 1u) * n_tile + loc[n]];
                    #pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (sl + j < ke[n]) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
                }
            }
            if (has_stray) {                    // added to what the stray points' atomics left, as the separate flush did
                float4 t = a.gradf[o];
                t.x += s.x; t.y += s.y; t.z += s.z; t.w += s.w;
                s = t;
                a.gradf[o] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (s.w > 0) {
                n4.x = c4.x + a.alpha * s.x / s.w;
                n4.y = c4.y + a.alpha * s.y / s.w;
                n4.z = c4.z + a.alpha * s.z / s.w;
            } else {
                n4.x = c4.x; n4.y = c4.y; n4.z = c4.z;
            }
            n4.w = s.w;
            if (store) a.grad[o] = n4;
        }
        return n4;
    };
    // step 3 for one pass: the LS_IC proposals of every control point added in image order
    auto accumulate = [&](uint32_t i0, const float4 n4) {
        prop[il][c][0] = n4.x; prop[il][c][1] = n4.y; prop[il][c][2] = n4.z;
        __syncthreads();
        if (tid < 3 * CPB) {
            const uint32_t n = min((uint32_t)LS_IC, a.n_owned - i0);
            for (uint32_t k = 0; k < n; k++) run += (double)prop[k][sc][sa];
        }
        __syncthreads();
    };
    if (kept) {
        // all passes' proposals first (independent chains of loads: side by side), then the sums in image order
        #pragma unroll
        for (int pass = 0; pass < LS_KEEP; pass++) {
            const uint32_t i0 = (uint32_t)pass * LS_IC;
            if (i0 >= a.n_owned) break;                 // block-uniform
            keep[pass] = propose(i0, false);
        }
        #pragma unroll
        for (int pass = 0; pass < LS_KEEP; pass++) {
            const uint32_t i0 = (uint32_t)pass * LS_IC;
            if (i0 >= a.n_owned) break;
            accumulate(i0, keep[pass]);
        }
    } else {
        for (uint32_t i0 = 0; i0 < a.n_owned; i0 += LS_IC) accumulate(i0, propose(i0, true));
    }
    if (tid < 3 * CPB) {
        const int scp = blockIdx.x * CPB + sc;
        if (scp < g.n_cp) a.gridsum[3 * (size_t)scp + sa] = run;
        mean[sc][sa] = a.n_images ? run / a.n_images : 0.0;
    }
    if (!CENTER) {
        // sharded contexts with two collectives per iteration: energy[0], [1] (this rank's sums, final since the scatter's launch)
        // and [3] (list flag) behind the proposal sums, so that ONE all-reduce adds both up; cp_center_kernel puts them back
        if (a.energy_tail && blockIdx.x == 0 && tid == 0) {
            a.energy_tail[0] = a.energy[0]; a.energy_tail[1] = a.energy[1]; a.energy_tail[2] = 0.0; a.energy_tail[3] = a.energy[3];
        }
        return;
    }
    __syncthreads();
    unsigned int cnt = 0;
    if (cp < g.n_cp) {
        const double mx = mean[c][0], my = mean[c][1], mz = mean[c][2];
        auto centre = [&](uint32_t img, float4 v) {
            if (g.mask_words && v.w < 0.0f) return;     // an inactive pair (propose): the node's shared value stands for it, below
            v.x = (float)((double)v.x - mx);
            v.y = (float)((double)v.y - my);
            v.z = (float)((double)v.z - mz);
            a.grad[lat(g, img, (uint32_t)cp)] = v;
            cnt += ((double)fabsf(v.x) > a.lim[0]) + ((double)fabsf(v.y) > a.lim[1]) + ((double)fabsf(v.z) > a.lim[2]);
        };
        if (kept) {
            #pragma unroll
            for (int pass = 0; pass < LS_KEEP; pass++) {
                const uint32_t img = (uint32_t)pass * LS_IC + il;
                if (img < a.n_owned) centre(img, keep[pass]);
            }
        } else {
            for (uint32_t img = il; img < a.n_owned; img += LS_IC) {
                if (g.mask_words && !lat_active(a.mask, g, img, (uint32_t)cp)) continue;
                centre(img, a.grad[lat(g, img, (uint32_t)cp)]);
            }
        }
        if (g.mask_words && il == 0) {
            // the inactive pairs of the node, all at once: their shared value minus the mean (the same (float)((double) v - mean) each
            // of them would get), and as many oversize coefficients as there are such pairs
            float4 u = u_node;
            u.x = (float)((double)u.x - mx); u.y = (float)((double)u.y - my); u.z = (float)((double)u.z - mz);
            a.ugrad[cp] = u;
            cnt += a.n_inactive[cp] * (((double)fabsf(u.x) > a.lim[0]) + ((double)fabsf(u.y) > a.lim[1]) + ((double)fabsf(u.z) > a.lim[2]));
        }
    }
    // oversize count -> energy[2] (zeroed by energy_reduce_kernel earlier in the step): integers added as f64 are exact and
    // their order does not matter.  Hardly any wavefront has something to add; no ticket, no last block.
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += (unsigned int)__shfl_down((int)cnt, off, 64);
    if ((tid & 63) == 0 && cnt) unsafeAtomicAdd(&a.energy[2], (double)cnt);
}

// ---- second half of K9 for contexts that own a sub-range of the images: subtract the group mean, count oversize
// coefficients.  Thread per control point walking its images; the loads of CP_BATCH images are issued together
// before any of them is used (one load per step made this a chain of ~n_images memory round trips).
constexpr int CP_BATCH = 10;

// (imageGroup.cxx:417-428); gridsum holds the sum over ALL images.
__global__ __launch_bounds__(256) void cp_center_kernel(float4 *__restrict__ grad, uint32_t n_owned, const GeomDev g, uint32_t n_images,
                                                        const double *gridsum, double lim_x, double lim_y, double lim_z,
                                                        double *energy, const double *energy_tail,
                                                        const uint32_t *mask, const uint32_t *n_inactive, const float4 *ucoeff, float4 *ugrad)
{
    const int n_cp = g.n_cp;
    const int cp = blockIdx.x * blockDim.x + threadIdx.x;
    // two collectives per iteration: the all-reduced energy sums and list flag come back from behind the proposal sums
    // (lattice_step_kernel put them there); energy[2], which the other threads count into, is not touched
    if (energy_tail && cp == 0) { energy[0] = energy_tail[0]; energy[1] = energy_tail[1]; energy[3] = energy_tail[3]; }
    unsigned int cnt = 0;
    if (cp < n_cp) {
        // n_images == 0: fixed images present, `sum` stays 0 (imageGroup.cxx:398,409-419)
        const double mx = n_images ? gridsum[3 * (size_t)cp] / n_images : 0.0;
        const double my = n_images ? gridsum[3 * (size_t)cp + 1] / n_images : 0.0;
        const double mz = n_images ? gridsum[3 * (size_t)cp + 2] / n_images : 0.0;
        for (uint32_t i0 = 0; i0 < n_owned; i0 += CP_BATCH) {
            float4 v[CP_BATCH];
            #pragma unroll
            for (int b = 0; b < CP_BATCH; b++) v[b] = grad[lat(g, min(i0 + b, n_owned - 1), (uint32_t)cp)];
            #pragma unroll
            for (int b = 0; b < CP_BATCH; b++) {
                if (i0 + b >= n_owned) break;
                if (g.mask_words && !lat_active(mask, g, i0 + b, (uint32_t)cp)) continue;      // inactive: the node's shared value, below
                v[b].x = (float)((double)v[b].x - mx);
                v[b].y = (float)((double)v[b].y - my);
                v[b].z = (float)((double)v[b].z - mz);
                grad[lat(g, i0 + b, (uint32_t)cp)] = v[b];
                cnt += ((double)fabsf(v[b].x) > lim_x) + ((double)fabsf(v[b].y) > lim_y) + ((double)fabsf(v[b].z) > lim_z);
            }
        }
        if (g.mask_words) {                 // as lattice_step_kernel<CENTER>: the inactive pairs of the node at once
            float4 u = ucoeff[cp];
            u.x = (float)((double)u.x - mx); u.y = (float)((double)u.y - my); u.z = (float)((double)u.z - mz);
            ugrad[cp] = u;
            cnt += n_inactive[cp] * (((double)fabsf(u.x) > lim_x) + ((double)fabsf(u.y) > lim_y) + ((double)fabsf(u.z) > lim_z));
        }
    }
    // the count goes to energy[2] as a double (so that one f64 all-reduce carries it); energy_reduce_kernel zeroed it
    // earlier in the step; integers added as f64 are exact, the order of the wavefronts does not matter
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += (unsigned int)__shfl_down((int)cnt, off, 64);
    if ((threadIdx.x & 63) == 0 && cnt) unsafeAtomicAdd(&energy[2], (double)cnt);
}

// One image's lattice out of coeff / gradf / a retired lattice into a contiguous array (the getters; layouts: lat())
__global__ __launch_bounds__(256) void lattice_extract_kernel(const float4 *src, const GeomDev g, uint32_t img, const uint32_t *mask,
                                                              const float4 *u, float4 *dst)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (uint32_t)g.n_cp) return;
    dst[n] = (g.mask_words && !lat_active(mask, g, img, n)) ? u[n] : src[lat(g, img, n)];
}

// Active pairs of a sparse lattice: (image, node) for every node in the 4^3 stencil of any of the image's points -- the cell of a
// point comes from `pos`, which does not change while the lattice stands, so the set is the lattice's for life: no other pair ever
// receives a gradient (gw = 0: the proposal is the standing value, imageGroup.cxx:346-375) or is read by the transform of the
// image's points.  One wavefront per scatter block (image, brick, run of the brick's points): the block's nodes are collected in
// an LDS bit set over the brick's (B + 3)^3 tile and go to the image's bit map ROW BY ROW -- a tile row is E consecutive nodes,
// i.e. E consecutive bits: one or two atomics.  (A thread per point with 64 atomics each took 11 ms per lattice on cfg 5's finest
// level; node-major bit sets over the images, one atomic per set node: 9.7.)  The mask was zeroed; pos_b = the positions in
// perm's order.
__device__ __forceinline__ void lat_mask_set(uint32_t *words, uint32_t first_bit, uint32_t bits)      // bits: <= 32 consecutive, from first_bit on
{
    const uint32_t w = first_bit >> 5, off = first_bit & 31u;
    const uint32_t lo = bits << off, hi = off ? bits >> (32u - off) : 0u;
    if (lo && (words[w] & lo) != lo) atomicOr(&words[w], lo);
    if (hi && (words[w + 1] & hi) != hi) atomicOr(&words[w + 1], hi);
}

__global__ __launch_bounds__(64) void lattice_mask_kernel(const float4 *pos_b, const ScatterBlock *blocks, const uint32_t *n_blocks,
                                                          const GeomDev g, uint32_t *mask)
{
    __shared__ uint32_t bits[(BRICK_CP_MAX * BRICK_CP_MAX * BRICK_CP_MAX + 31) / 32 + 1];
    if (blockIdx.x >= *n_blocks) return;
    const int lane = threadIdx.x;
    const ScatterBlock blk = blocks[blockIdx.x];
    const int E = g.brick + 3, n_tile = E * E * E;
    for (int k = lane; k < (n_tile + 31) / 32 + 1; k += 64) bits[k] = 0u;
    const uint32_t img = blk.key / g.n_bricks;
    uint32_t bidx = blk.key - img * g.n_bricks;
    const int bx = bidx % g.nbricks[0]; bidx /= g.nbricks[0];
    const int by = bidx % g.nbricks[1];
    const int bz = bidx / g.nbricks[1];
    const int cp0[3] = { bx * g.brick, by * g.brick, bz * g.brick };
    uint32_t *words = mask + (size_t)img * g.mask_words;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t sidx = blk.begin + lane; sidx < blk.end; sidx += 64) {
        const float4 v = pos_b[sidx];
        const float in[3] = { v.x, v.y, v.z };
        int ic[3]; float fr[3];
        scatter_cell(in, g, ic, fr);
        // ... and the cell as the transforms find it (bspline_axis: the f64 quotient, by division or by the reciprocal): a point within
        // an ulp of a cell face sits one cell further there, and the transform reads that stencil
        int lo[3], hi[3];
        #pragma unroll
        for (int k = 0; k < 3; k++) {
            const int c1 = (int)floor(((double)in[k] - g.origin[k]) / g.spacing[k]), c2 = (int)floor(((double)in[k] - g.origin[k]) * g.inv_spacing[k]);
            lo[k] = min(ic[k], min(c1, c2)) - 1;
            hi[k] = max(ic[k], max(c1, c2)) + 2;
        }
        for (int gz = lo[2]; gz <= hi[2]; gz++) for (int gy = lo[1]; gy <= hi[1]; gy++) for (int gx = lo[0]; gx <= hi[0]; gx++) {
            if (gx < 0 || gy < 0 || gz < 0 || gx >= g.dims[0] || gy >= g.dims[1] || gz >= g.dims[2]) continue;
            const int lx = gx - cp0[0], ly = gy - cp0[1], lz = gz - cp0[2];
            if (lx >= 0 && ly >= 0 && lz >= 0 && lx < E && ly < E && lz < E) {
                const int t = lx + E * (ly + E * lz);
                atomicOr(&bits[t >> 5], 1u << (t & 31));
            } else {                            // a stray point, or a stencil across a face of the brick's tile: straight to the bit map
                lat_mask_set(words, (uint32_t)gx + (uint32_t)g.dims[0] * ((uint32_t)gy + (uint32_t)g.dims[1] * (uint32_t)gz), 1u);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int row = lane; row < E * E; row += 64) {          // tile row (ly, lz): nodes cp0x .. cp0x + E - 1 of lattice row (gy, gz)
        const int ly = row % E, lz = row / E;
        const int gy = cp0[1] + ly, gz = cp0[2] + lz;
        if (gy >= g.dims[1] || gz >= g.dims[2]) continue;
        const int t0 = E * row;
        const unsigned long long two = (unsigned long long)bits[t0 >> 5] | ((unsigned long long)bits[(t0 >> 5) + 1] << 32);
        uint32_t rb = (uint32_t)(two >> (t0 & 31)) & ((1u << E) - 1u);
        const int in_x = g.dims[0] - cp0[0];                // nodes of the row inside the lattice
        if (in_x < E) rb &= (1u << max(in_x, 0)) - 1u;
        if (rb) lat_mask_set(words, (uint32_t)cp0[0] + (uint32_t)g.dims[0] * ((uint32_t)gy + (uint32_t)g.dims[1] * (uint32_t)gz), rb);
    }
}

// n_inactive[node] = owned images for which the node is inactive (the guard's count of a node's shared value, lattice_step_kernel)
__global__ __launch_bounds__(256) void lattice_inactive_kernel(const uint32_t *mask, uint32_t n_owned, const GeomDev g, uint32_t *n_inactive)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (uint32_t)g.n_cp) return;
    uint32_t active = 0;
    for (uint32_t img = 0; img < n_owned; img++) active += (mask[(size_t)img * g.mask_words + (n >> 5)] >> (n & 31u)) & 1u;
    n_inactive[n] = n_owned - active;
}

} // namespace frog

// prep.h -- host-side construction of the device link layout from the
// reference-order CSR (frog_model).
//
// The reference walks   for image A: for point pA: for link (B, pB)   through
// per-point heap vectors (point.h:19-32).  On the device the half-links of each
// run of TILE_POINTS consecutive points of an image ("tile", one wavefront) are
// regrouped partner-major: all links into image 0, then image 1, ... each group
// in point order.  Every tile therefore sweeps the partner images in the same
// ascending order and, since all tiles of a launch are resident at once and
// advance at about the same pace, the whole chip gathers partner coordinates
// from a window of a few images at a time: the gathers hit the 4 MiB per-XCD L2
// instead of scattering over the whole coordinate table.  Points are first renumbered
// along a Morton curve inside each image (see Layout).  Within a point the
// order is partner-image ascending (stable: links into one partner image keep
// their file order).  What that guarantees about the f32 per-point sums of the
// deformable step (imageGroup.cxx:270-278):
//   * inside ONE partner group (1/8 of the partner images, one XCD) the adds of a
//     point happen in partner-ascending order, which is the reference's order
//     for files whose blocks are i-major / j-ascending -- what `match` writes
//     (match.cpp:727-742) and readPairs then stores (imageGroup.cxx:1405-1406);
//   * the 8 group sums of a point are added afterwards, in group order: the
//     association differs from the reference's single running sum (last-ulp
//     differences, inside the 1e-5 the parity tests allow for these sums);
//   * for files with another block order (other producers) the order inside a
//     group is still partner-ascending, i.e. NOT the file's: same class of
//     last-ulp difference (tests/test_gpu_round2.py checks a shuffled-block file).
// Weights within 1e-4 of the inlier threshold are re-evaluated with the
// reference's own arithmetic (k_links.hip.h), so these ulps cannot flip a link.
#pragma once

#include "ctx.h"
#include "../common/usable_cpus.h"
#include "../common/bulk_alloc.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <utility>
#include <omp.h>

// Host threads of the layout build: frog::host_threads() (common/usable_cpus.h) -- what OpenMP offers, capped by the CPUs the
// process may use (affinity mask, cgroup quota) and by 64: on a 256-thread host, with one process per GPU, eight uncapped pools
// would be 2048 threads for a job that takes half a second.

namespace frog {

// Bulk<T> (common/bulk_alloc.h): the arrays with an entry per half-link -- no zero fill on resize(), huge pages.

struct Layout {
    // Internal point numbering: inside every image the points are renumbered along a
    // Morton curve of their coordinates normalised to the image's bounding box, so that
    // (a) a tile is a compact blob and the partner points it links to sit in a narrow
    // index window of the partner image, and (b) "the k-th eighth of an image" means the
    // same region of space in every image.  new_of_old / old_of_new map global indices.
    std::vector<uint32_t> new_of_old, old_of_new;
    // Partner images are split into n_groups contiguous ranges of equal point count;
    // group_begin[g] is the first image of group g.  A sweep block handles (4 tiles,
    // ONE group); blocks are dealt round-robin over the 8 XCDs, so with block % 8 = group
    // every XCD gathers from one group's coordinates only: 1/8 of the table, which fits
    // its 4 MiB L2 -- also for false matches, whose partner points are uniformly random.
    std::vector<uint32_t> group_begin;      // [n_groups + 1]
    std::vector<Tile> tiles;
    Bulk<LinkRec> recs;                     // wide records, or
    Bulk<uint32_t> recs32;                  // narrow records (RecFormat)
    RecFormat format{};
    std::vector<uint32_t> img_tile_ptr;     // [nI + 1]
    std::vector<uint64_t> ref_rowptr;       // owned rows, relative to first owned link
    Bulk<uint32_t> ref_link;                // partner global index, reference order
    std::vector<uint64_t> img_link_begin;   // [nI + 1] (relative, owned images only meaningful)
};

inline uint32_t spread3(uint32_t v)          // 10 bits -> every third bit
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// Morton permutation of every image (all ranks compute the same one: they all hold the model).
inline void build_numbering(const frog_model &m, Layout &out)
{
    const uint32_t nI = m.n_images;
    const uint32_t *poff = m.point_offset;
    const uint64_t P = poff[nI];
    out.new_of_old.resize(P);
    out.old_of_new.resize(P);
#ifdef KMP_VERSION_MAJOR
    // libomp's workers spin for 200 ms after a parallel region by default; with the serial stretches
    // between the regions below and the caller's own threads that more than doubled the set-up time
    // (1.15 s -> 0.49 s for 100 images, measured), so they go to sleep at once
    kmp_set_blocktime(0);
#endif
    #pragma omp parallel for schedule(dynamic) num_threads(host_threads())
    for (int i = 0; i < (int)nI; i++) {
        const uint32_t b = poff[i], e = poff[i + 1];
        float mn[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, mx[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
        for (uint32_t p = b; p < e; p++)
            for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], m.xyz[3 * (size_t)p + k]); mx[k] = std::max(mx[k], m.xyz[3 * (size_t)p + k]); }
        // (Morton key, old index) ascending = a stable sort by key of the points in index order: three counting passes of 10 bits
        // (std::sort of 20 000 pairs was 1.3 ms per image -- with a hundred images over sixteen threads the slowest stage of the
        // layout build)
        const uint32_t n_pts = e - b;
        std::vector<std::pair<uint32_t, uint32_t>> key(n_pts), tmp(n_pts);
        for (uint32_t p = b; p < e; p++) {
            uint32_t q[3];
            for (int k = 0; k < 3; k++) {
                const float ext = mx[k] - mn[k];
                float u = ext > 0 ? (m.xyz[3 * (size_t)p + k] - mn[k]) / ext : 0.f;
                if (!(u >= 0.f)) u = 0.f;                  // also catches NaN
                if (u > 1.f) u = 1.f;
                q[k] = (uint32_t)(u * 1023.0f);
            }
            key[p - b] = { spread3(q[0]) | (spread3(q[1]) << 1) | (spread3(q[2]) << 2), p };
        }
        for (int pass = 0; pass < 3; pass++) {
            uint32_t count[1025] = { 0 };
            const int sh = 10 * pass;
            for (uint32_t n = 0; n < n_pts; n++) count[((key[n].first >> sh) & 1023u) + 1]++;
            for (int c = 0; c < 1024; c++) count[c + 1] += count[c];
            for (uint32_t n = 0; n < n_pts; n++) tmp[count[(key[n].first >> sh) & 1023u]++] = key[n];
            key.swap(tmp);
        }
        for (uint32_t n = 0; n < e - b; n++) {
            out.old_of_new[b + n] = key[n].second;
            out.new_of_old[key[n].second] = b + n;
        }
    }
}

inline int build_layout(const frog_model &m, uint32_t ib, uint32_t ie, bool force_wide, int n_groups, Layout &out, std::string &err)
{
    const uint32_t nI = m.n_images;
    const uint32_t *poff = m.point_offset;
    const auto t_in = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (std::getenv("FROG_TIMING")) std::printf("[timing] build_layout, %s : %gs\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_in).count());
    };
    build_numbering(m, out);
    lap("numbering");
    const std::vector<uint32_t> &new_of_old = out.new_of_old, &old_of_new = out.old_of_new;
    const uint32_t *new_of_old_base = poff;      // internal numbering keeps every image's index range
    const uint32_t p0 = poff[ib], p1 = poff[ie];
    const uint64_t l0 = m.row_ptr[p0], l1 = m.row_ptr[p1];
    const uint64_t L = l1 - l0;
    if (L >= 0xFFFFFFFFull) { err = "more than 2^32-1 half-links in one context"; return FROG_E_INVALID; }

    // reference-order CSR of the owned rows
    out.ref_rowptr.resize((size_t)(p1 - p0) + 1);
    for (uint32_t p = p0; p <= p1; p++) out.ref_rowptr[p - p0] = m.row_ptr[p] - l0;
    out.ref_link.resize(L);
    int bad = 0;
    #pragma omp parallel for reduction(| : bad) num_threads(host_threads())
    for (long long l = 0; l < (long long)L; l++) {
        const uint16_t im = m.link_image[l0 + l];
        const uint32_t pt = m.link_point[l0 + l];
        if (im >= nI || pt >= poff[im + 1] - poff[im]) { bad = 1; out.ref_link[l] = 0; continue; }
        out.ref_link[l] = new_of_old[poff[im] + pt];
    }
    if (bad) { err = "link references a point outside its image"; return FROG_E_INVALID; }
    lap("+ reference-order links renumbered");
    out.img_link_begin.assign(nI + 1, 0);
    for (uint32_t i = 0; i <= nI; i++) {
        uint32_t c = std::min(std::max(i, ib), ie);
        out.img_link_begin[i] = m.row_ptr[poff[c]] - l0;
    }

    // partner groups: contiguous image ranges balanced by point count
    out.group_begin.assign(n_groups + 1, nI);
    out.group_begin[0] = 0;
    {
        const uint64_t Pall = poff[nI];
        uint32_t img = 0;
        for (int g = 1; g < n_groups; g++) {
            const uint64_t target = Pall * (uint64_t)g / n_groups;
            while (img < nI && poff[img] < target) img++;
            out.group_begin[g] = img;
        }
    }
    std::vector<uint8_t> group_of(nI);
    for (int g = 0; g < n_groups; g++)
        for (uint32_t i = out.group_begin[g]; i < out.group_begin[g + 1]; i++) group_of[i] = (uint8_t)g;

    // tiles
    out.img_tile_ptr.assign(nI + 1, 0);
    for (uint32_t i = ib; i < ie; i++) {
        const uint32_t np = poff[i + 1] - poff[i];
        const uint32_t nt = (np + TILE_POINTS - 1) / TILE_POINTS;
        for (uint32_t t = 0; t < nt; t++) {
            Tile tl{};
            tl.pt_begin = poff[i] + t * TILE_POINTS;          // NEW numbering
            tl.pt_count = std::min<uint32_t>(TILE_POINTS, poff[i + 1] - tl.pt_begin);
            tl.image = i;
            out.tiles.push_back(tl);
        }
        out.img_tile_ptr[i + 1] = (uint32_t)out.tiles.size();
    }
    for (uint32_t i = 0; i < ib; i++) out.img_tile_ptr[i + 1] = 0;
    for (uint32_t i = ie; i < nI; i++) out.img_tile_ptr[i + 1] = (uint32_t)out.tiles.size();

    // records per (tile, group), then the padded offsets
    const long long nT = (long long)out.tiles.size();
    #pragma omp parallel for schedule(dynamic, 64) num_threads(host_threads())
    for (long long t = 0; t < nT; t++) {
        Tile &tl = out.tiles[t];
        for (uint32_t n = tl.pt_begin; n < tl.pt_begin + tl.pt_count; n++) {
            const uint32_t o = old_of_new[n];
            for (uint64_t l = m.row_ptr[o]; l < m.row_ptr[o + 1]; l++) tl.group_cnt[group_of[m.link_image[l]]]++;
        }
    }
    lap("+ records counted");
    uint64_t rec_total = 0;
    for (long long t = 0; t < nT; t++) {
        Tile &tl = out.tiles[t];
        tl.rec_begin = (uint32_t)rec_total;
        uint32_t off = 0;
        for (int g = 0; g < n_groups; g++) {
            tl.group_off[g] = off;
            off += (tl.group_cnt[g] + REC_CHUNK - 1) / REC_CHUNK * REC_CHUNK;
        }
        rec_total += off;
        if (rec_total >= 0xFFFFFFFFull) { err = "more than 2^32-1 link records in one context"; return FROG_E_INVALID; }
    }
    const bool no_records = rec_total == 0;
    if (no_records) rec_total = REC_CHUNK;                   // the sweep's clamped prefetch needs one readable chunk

    // record format: narrow when (own point, partner image in its group, partner point in its image) fit 32 bits
    {
        uint32_t widest_group = 1, largest_image = 1;
        for (int g = 0; g < n_groups; g++) widest_group = std::max(widest_group, out.group_begin[g + 1] - out.group_begin[g]);
        for (uint32_t i = 0; i < nI; i++) largest_image = std::max(largest_image, poff[i + 1] - poff[i]);
        auto bits_for = [](uint32_t n) { uint32_t b = 1; while ((1ull << b) < n) b++; return b; };   // values 0 .. n-1
        const uint32_t img_bits = bits_for(widest_group), pt_bits = bits_for(largest_image);
        // img_bits <= 8: the narrow kernel keeps the group's constants and first-point table in 2^8-entry LDS arrays
        // (k_links.hip.h EMD_LDS_IMAGES); a group of more images takes the wide form, whose kernel reads them from memory
        out.format.narrow = (8 + img_bits + pt_bits <= 32 && img_bits <= 8 && !force_wide) ? 1u : 0u;
        out.format.img_bits = img_bits;
    }
    const bool narrow = out.format.narrow != 0;
    const uint32_t img_bits = out.format.img_bits;

    // partner-major records, stable counting sort per tile
    // (every slot is written below, padding included -- except the one readable chunk of a context without records)
    if (narrow) { out.recs32.resize(rec_total); if (no_records) std::fill(out.recs32.begin(), out.recs32.end(), 0u); }
    else { out.recs.resize(rec_total); if (no_records) std::fill(out.recs.begin(), out.recs.end(), LinkRec{ 0u, 0u }); }
    #pragma omp parallel num_threads(host_threads())
    {
        std::vector<uint32_t> cnt(nI + 1);
        std::vector<LinkRec> logical;
        #pragma omp for schedule(dynamic, 16)
        for (long long t = 0; t < nT; t++) {
            const Tile &tl = out.tiles[t];
            std::fill(cnt.begin(), cnt.end(), 0u);
            for (uint32_t n = tl.pt_begin; n < tl.pt_begin + tl.pt_count; n++) {
                const uint32_t o = old_of_new[n];
                for (uint64_t l = m.row_ptr[o]; l < m.row_ptr[o + 1]; l++) cnt[m.link_image[l] + 1]++;
            }
            for (uint32_t i = 0; i < nI; i++) cnt[i + 1] += cnt[i];
            uint32_t group_first[MAX_GROUPS + 1];               // logical (unpadded) start of every group
            for (int g = 0; g <= n_groups; g++) group_first[g] = cnt[out.group_begin[g]];
            logical.resize(cnt[nI]);
            LinkRec *dst = logical.data();
            for (uint32_t n = tl.pt_begin; n < tl.pt_begin + tl.pt_count; n++) {
                const uint32_t o = old_of_new[n];
                for (uint64_t l = m.row_ptr[o]; l < m.row_ptr[o + 1]; l++) {
                    LinkRec r;
                    r.a = ((uint32_t)m.link_image[l] << 8) | (n - tl.pt_begin);
                    r.b = out.ref_link[l - l0];
                    dst[cnt[m.link_image[l]]++] = r;
                }
            }
            // logical order -> chunked, transposed storage (ctx.h, REC_CHUNK)
            for (int g = 0; g < n_groups; g++) {
                const size_t at = (size_t)tl.rec_begin + tl.group_off[g];
                const LinkRec *src = dst + group_first[g];
                for (uint32_t k = 0; k < tl.group_cnt[g]; k++) {
                    const size_t phys = at + (k / REC_CHUNK) * REC_CHUNK + (k % 64u) * 2u + (k % REC_CHUNK) / 64u;
                    if (narrow) {
                        const uint32_t img = src[k].a >> 8;
                        out.recs32[phys] = ((src[k].b - new_of_old_base[img]) << (8 + img_bits))
                                         | ((img - out.group_begin[g]) << 8) | (src[k].a & 0xFFu);
                    } else {
                        out.recs[phys] = src[k];
                    }
                }
                const uint32_t padded = (tl.group_cnt[g] + REC_CHUNK - 1) / REC_CHUNK * REC_CHUNK;      // null records to the chunk's end
                for (uint32_t k = tl.group_cnt[g]; k < padded; k++) {
                    const size_t phys = at + (k / REC_CHUNK) * REC_CHUNK + (k % 64u) * 2u + (k % REC_CHUNK) / 64u;
                    if (narrow) out.recs32[phys] = 0u; else out.recs[phys] = LinkRec{ 0u, 0u };
                }
            }
        }
    }

    lap("+ records placed");
    return FROG_OK;
}

} // namespace frog

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
// instead of scattering over the whole coordinate table.  Within a point the
// order stays partner-ascending, which is the order readPairs produces for
// files written by match (blocks i-major, j-ascending: imageGroup.cxx:1405-1406,
// match.cpp:727-742), so per-point f32 sums keep the reference's order.
#pragma once

#include "ctx.h"

#include <algorithm>
#include <omp.h>

namespace frog {

struct Layout {
    std::vector<Tile> tiles;
    std::vector<LinkRec> recs;
    std::vector<uint32_t> img_tile_ptr;     // [nI + 1]
    std::vector<uint64_t> ref_rowptr;       // owned rows, relative to first owned link
    std::vector<uint32_t> ref_link;         // partner global index, reference order
    std::vector<uint64_t> img_link_begin;   // [nI + 1] (relative, owned images only meaningful)
};

inline int build_layout(const frog_model &m, uint32_t ib, uint32_t ie, Layout &out, std::string &err)
{
    const uint32_t nI = m.n_images;
    const uint32_t *poff = m.point_offset;
    const uint32_t p0 = poff[ib], p1 = poff[ie];
    const uint64_t l0 = m.row_ptr[p0], l1 = m.row_ptr[p1];
    const uint64_t L = l1 - l0;
    if (L >= 0xFFFFFFFFull) { err = "more than 2^32-1 half-links in one context"; return FROG_E_INVALID; }

    // reference-order CSR of the owned rows
    out.ref_rowptr.resize((size_t)(p1 - p0) + 1);
    for (uint32_t p = p0; p <= p1; p++) out.ref_rowptr[p - p0] = m.row_ptr[p] - l0;
    out.ref_link.resize(L);
    int bad = 0;
    #pragma omp parallel for reduction(| : bad)
    for (long long l = 0; l < (long long)L; l++) {
        const uint16_t im = m.link_image[l0 + l];
        const uint32_t pt = m.link_point[l0 + l];
        if (im >= nI || pt >= poff[im + 1] - poff[im]) { bad = 1; out.ref_link[l] = 0; continue; }
        out.ref_link[l] = poff[im] + pt;
    }
    if (bad) { err = "link references a point outside its image"; return FROG_E_INVALID; }
    out.img_link_begin.assign(nI + 1, 0);
    for (uint32_t i = 0; i <= nI; i++) {
        uint32_t c = std::min(std::max(i, ib), ie);
        out.img_link_begin[i] = m.row_ptr[poff[c]] - l0;
    }

    // tiles
    out.img_tile_ptr.assign(nI + 1, 0);
    for (uint32_t i = ib; i < ie; i++) {
        const uint32_t np = poff[i + 1] - poff[i];
        const uint32_t nt = (np + TILE_POINTS - 1) / TILE_POINTS;
        for (uint32_t t = 0; t < nt; t++) {
            Tile tl{};
            tl.pt_begin = poff[i] + t * TILE_POINTS;
            tl.pt_count = std::min<uint32_t>(TILE_POINTS, poff[i + 1] - tl.pt_begin);
            tl.rec_begin = (uint32_t)(m.row_ptr[tl.pt_begin] - l0);
            tl.rec_count = (uint32_t)(m.row_ptr[tl.pt_begin + tl.pt_count] - m.row_ptr[tl.pt_begin]);
            tl.image = i;
            out.tiles.push_back(tl);
        }
        out.img_tile_ptr[i + 1] = (uint32_t)out.tiles.size();
    }
    for (uint32_t i = 0; i < ib; i++) out.img_tile_ptr[i + 1] = 0;
    for (uint32_t i = ie; i < nI; i++) out.img_tile_ptr[i + 1] = (uint32_t)out.tiles.size();

    // partner-major records, stable counting sort per tile
    out.recs.resize(L);
    const long long nT = (long long)out.tiles.size();
    #pragma omp parallel
    {
        std::vector<uint32_t> cnt(nI + 1);
        #pragma omp for schedule(dynamic, 16)
        for (long long t = 0; t < nT; t++) {
            const Tile &tl = out.tiles[t];
            std::fill(cnt.begin(), cnt.end(), 0u);
            const uint64_t a = m.row_ptr[tl.pt_begin], b = m.row_ptr[tl.pt_begin + tl.pt_count];
            for (uint64_t l = a; l < b; l++) cnt[m.link_image[l] + 1]++;
            for (uint32_t i = 0; i < nI; i++) cnt[i + 1] += cnt[i];
            LinkRec *dst = out.recs.data() + tl.rec_begin;
            for (uint32_t p = tl.pt_begin; p < tl.pt_begin + tl.pt_count; p++)
                for (uint64_t l = m.row_ptr[p]; l < m.row_ptr[p + 1]; l++) {
                    LinkRec r;
                    r.a = p;
                    r.b = out.ref_link[l - l0];
                    dst[cnt[m.link_image[l]]++] = r;
                }
        }
    }
    return FROG_OK;
}

} // namespace frog

// pairs_store.h -- in-memory pairs.bin (host side).
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/frog_host.h"
#include "../common/bulk_alloc.h"

template <class T> using frog_bulk = frog::Bulk<T>;       // common/bulk_alloc.h: no zero fill on resize(), huge pages

struct frog_pairs {
    uint32_t n_images = 0;
    std::vector<std::string> names;
    std::vector<double> ref_translation;     // 3 per image (read, never used: imageGroup.cxx:1370)
    std::vector<uint32_t> point_offset;      // n_images + 1
    std::vector<float> xyz;                  // 3 * P
    std::vector<float> other;                // 3 * P  (scale, laplacian sign, response)

    // pair blocks in file order
    std::vector<uint16_t> block_image1, block_image2;
    std::vector<uint64_t> block_ptr;         // n_blocks + 1
    frog_bulk<uint32_t> p1, p2;               // point index inside image1 / image2

    // half-link CSR in reference order (build_links)
    std::vector<uint64_t> row_ptr;           // P + 1
    frog_bulk<uint16_t> link_image;
    frog_bulk<uint32_t> link_point;

    uint64_t num_points() const { return point_offset.empty() ? 0 : point_offset.back(); }
    uint64_t num_pairs() const { return p1.size(); }
    // replays readPairs' push_back order (imageGroup.cxx:1400-1408)
    void build_links();
};

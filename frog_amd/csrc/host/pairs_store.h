// pairs_store.h -- in-memory pairs.bin (host side).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/frog_host.h"

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
    std::vector<uint32_t> p1, p2;            // point index inside image1 / image2

    // half-link CSR in reference order (build_links)
    std::vector<uint64_t> row_ptr;           // P + 1
    std::vector<uint16_t> link_image;
    std::vector<uint32_t> link_point;

    uint64_t num_points() const { return point_offset.empty() ? 0 : point_offset.back(); }
    uint64_t num_pairs() const { return p1.size(); }
    // replays readPairs' push_back order (imageGroup.cxx:1400-1408)
    void build_links();
};

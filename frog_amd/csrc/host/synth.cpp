// synth.cpp -- deterministic synthetic keypoint groups (pairs.bin content).
//
// The reference ships no sample data and its keypoint extractor (vtkOpenSURF3D)
// is an empty submodule, so benchmarks and tests synthesise what `match` would
// have written:
//   * K landmarks uniform in a 400 x 400 x 600 mm common space;
//   * image i sees each landmark with probability p_i in [0.5,1], at
//     S_i (x + bump_i(x)) + t_i + N(0, sigma): anisotropic scale, translation, a
//     smooth displacement (3 Gaussian bumps) and localisation noise;
//   * clutter points fill every image up to exactly points_per_image;
//   * for each linked image pair, true matches between co-observed landmarks
//     plus a fraction of random false matches, sorted by the index in image1
//     (the order `match` emits: match.cpp loops over image1's points).
// All randomness is counter-based (splitmix64 of (seed, stream, counter)), so
// the result does not depend on thread count or libstdc++ distributions.

#include "../common/usable_cpus.h"
#include "pairs_store.h"

#include <algorithm>
#include <cmath>
#include <numeric>

#include <omp.h>

namespace {

inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct Stream {
    uint64_t key, ctr = 0;
    Stream(uint64_t seed, uint64_t a, uint64_t b = 0) : key(mix64(mix64(seed ^ 0xF20Cull) + mix64(a * 0x100000001B3ull + b))) {}
    uint64_t next() { return mix64(key + (ctr++) * 0xD1342543DE82EF95ull); }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }    // [0,1)
    double range(double a, double b) { return a + (b - a) * uniform(); }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    double normal()
    {
        double u1 = uniform(), u2 = uniform();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

struct Warp {
    double scale[3], shift[3];
    double bump_c[3][3], bump_a[3][3];      // 3 Gaussians: centre, amplitude vector
    double sigma2;
    void apply(const double x[3], double out[3]) const
    {
        double d[3] = { 0, 0, 0 };
        for (int b = 0; b < 3; b++) {
            double r2 = 0;
            for (int k = 0; k < 3; k++) r2 += (x[k] - bump_c[b][k]) * (x[k] - bump_c[b][k]);
            double w = std::exp(-0.5 * r2 / sigma2);
            for (int k = 0; k < 3; k++) d[k] += w * bump_a[b][k];
        }
        for (int k = 0; k < 3; k++) out[k] = scale[k] * (x[k] + d[k]) + shift[k];
    }
};

const double BOX[3] = { 400.0, 400.0, 600.0 };

} // namespace

extern "C" {

void frog_synth_defaults(frog_synth_params *p)
{
    p->n_images = 4;
    p->points_per_image = 20000;
    p->n_landmarks = 0;
    p->pairs_per_block = 10000;
    p->partners_per_image = 0;
    p->outlier_fraction = 0.3f;
    p->noise_sigma = 2.0f;
    p->bump_amplitude = 15.0f;
    p->scale_min = 0.8f;
    p->scale_max = 1.25f;
    p->translation_range = 100.0f;
    p->seed = 1;
}

frog_pairs *frog_synth_generate(const frog_synth_params *sp)
{
    const uint32_t nI = sp->n_images;
    const uint32_t nP = sp->points_per_image;
    const uint32_t K = sp->n_landmarks ? sp->n_landmarks : nP;
    if (nI < 2 || nI > 65535 || nP == 0) return nullptr;

    frog_pairs *out = new frog_pairs;
    out->n_images = nI;
    out->point_offset.resize(nI + 1);
    for (uint32_t i = 0; i <= nI; i++) out->point_offset[i] = i * nP;
    out->xyz.resize((size_t)3 * nI * nP);
    out->other.resize((size_t)3 * nI * nP);
    out->ref_translation.assign((size_t)3 * nI, 0.0);
    for (uint32_t i = 0; i < nI; i++) out->names.push_back("synthetic_" + std::to_string(i) + ".csv.gz");

    // common-space landmarks
    std::vector<double> lm((size_t)3 * K);
    {
        Stream s(sp->seed, 0xA11CE);
        for (size_t k = 0; k < (size_t)3 * K; k++) lm[k] = s.uniform() * BOX[k % 3];
    }

    // per image: which landmark sits at which point index (-1: not observed)
    std::vector<int32_t> lm2pt((size_t)nI * K, -1);

    #pragma omp parallel for schedule(dynamic) num_threads(frog::host_threads())
    for (int i = 0; i < (int)nI; i++) {
        Stream s(sp->seed, 0x1A6E, (uint64_t)i);
        Warp w;
        for (int k = 0; k < 3; k++) {
            w.scale[k] = s.range(sp->scale_min, sp->scale_max);
            w.shift[k] = s.range(-sp->translation_range, sp->translation_range);
        }
        w.sigma2 = 80.0 * 80.0;
        for (int b = 0; b < 3; b++)
            for (int k = 0; k < 3; k++) {
                w.bump_c[b][k] = s.uniform() * BOX[k];
                w.bump_a[b][k] = s.range(-1.0, 1.0) * sp->bump_amplitude / 1.7;
            }
        const double p_obs = s.range(0.5, 1.0);
        // slot permutation: point index of the n-th generated point
        std::vector<uint32_t> slot(nP);
        std::iota(slot.begin(), slot.end(), 0u);
        for (uint32_t k = nP - 1; k > 0; k--) std::swap(slot[k], slot[s.below(k + 1)]);
        uint32_t used = 0;
        float *xyz = &out->xyz[(size_t)3 * i * nP];
        float *oth = &out->other[(size_t)3 * i * nP];
        auto emit = [&](const double x[3]) {
            double y[3];
            w.apply(x, y);
            uint32_t pt = slot[used++];
            for (int k = 0; k < 3; k++) xyz[3 * (size_t)pt + k] = (float)(y[k] + sp->noise_sigma * s.normal());
            oth[3 * (size_t)pt] = 2.0f;
            oth[3 * (size_t)pt + 1] = (s.next() & 1) ? 1.0f : -1.0f;
            oth[3 * (size_t)pt + 2] = (float)s.uniform();
            return pt;
        };
        for (uint32_t k = 0; k < K && used < nP; k++) {
            if (s.uniform() >= p_obs) continue;
            lm2pt[(size_t)i * K + k] = (int32_t)emit(&lm[(size_t)3 * k]);
        }
        while (used < nP) {           // clutter
            double x[3] = { s.uniform() * BOX[0], s.uniform() * BOX[1], s.uniform() * BOX[2] };
            emit(x);
        }
    }

    // linked image pairs, i-major / j-ascending like match.cpp:727-742
    std::vector<std::pair<uint16_t, uint16_t>> blocks;
    for (uint32_t i = 0; i < nI; i++)
        for (uint32_t j = i + 1; j < nI; j++) {
            if (sp->partners_per_image && sp->partners_per_image < nI - 1) {
                Stream s(sp->seed, 0xB10C, (uint64_t)i * 65536 + j);
                if (s.uniform() >= (double)sp->partners_per_image / (double)(nI - 1)) continue;
            }
            blocks.emplace_back((uint16_t)i, (uint16_t)j);
        }
    const size_t nb = blocks.size();
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> bp(nb);

    #pragma omp parallel for schedule(dynamic, 4) num_threads(frog::host_threads())
    for (long b = 0; b < (long)nb; b++) {
        const uint32_t i = blocks[b].first, j = blocks[b].second;
        Stream s(sp->seed, 0x9A125, (uint64_t)i * 65536 + j);
        uint32_t n = (uint32_t)std::llround(sp->pairs_per_block * s.range(0.9, 1.1));
        if (n < 1) n = 1;
        uint32_t n_false = (uint32_t)std::llround((double)n * sp->outlier_fraction);
        uint32_t n_true = n - n_false;
        std::vector<std::pair<uint32_t, uint32_t>> co;
        for (uint32_t k = 0; k < K; k++) {
            int32_t a = lm2pt[(size_t)i * K + k], c = lm2pt[(size_t)j * K + k];
            if (a >= 0 && c >= 0) co.emplace_back((uint32_t)a, (uint32_t)c);
        }
        if (n_true > co.size()) n_true = (uint32_t)co.size();
        for (uint32_t k = 0; k < n_true; k++) {          // partial Fisher-Yates
            uint32_t r = k + s.below((uint32_t)co.size() - k);
            std::swap(co[k], co[r]);
        }
        auto &v = bp[b];
        v.assign(co.begin(), co.begin() + n_true);
        for (uint32_t k = 0; k < n_false; k++) v.emplace_back(s.below(nP), s.below(nP));
        std::stable_sort(v.begin(), v.end(),
                         [](const std::pair<uint32_t, uint32_t> &x, const std::pair<uint32_t, uint32_t> &y) { return x.first < y.first; });
    }

    out->block_ptr.assign(1, 0);
    for (size_t b = 0; b < nb; b++) {
        if (bp[b].empty()) continue;
        out->block_image1.push_back(blocks[b].first);
        out->block_image2.push_back(blocks[b].second);
        out->block_ptr.push_back(out->block_ptr.back() + bp[b].size());
    }
    out->p1.resize(out->block_ptr.back());
    out->p2.resize(out->block_ptr.back());
    size_t ob = 0;
    for (size_t b = 0; b < nb; b++) {
        if (bp[b].empty()) continue;
        uint64_t base = out->block_ptr[ob++];
        for (size_t k = 0; k < bp[b].size(); k++) { out->p1[base + k] = bp[b][k].first; out->p2[base + k] = bp[b][k].second; }
        std::vector<std::pair<uint32_t, uint32_t>>().swap(bp[b]);
    }
    out->build_links();
    return out;
}

} // extern "C"

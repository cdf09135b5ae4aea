// pairs_store.cpp -- pairs.bin reader / writer and the reference-order link CSR.
//
// Wire format (little-endian, unpadded; reader imageGroup.cxx:1355-1412, writer
// match.cpp:684-742, pointIdType = u32 because INT_PTIDS defaults ON):
//   u16 nImages
//   nImages x { u16 nameLength; char name[nameLength]; f64 refTranslation[3];
//               u32 nPoints; nPoints x { f32 xyz[3]; f32 other[3] } }
//   until EOF: { u16 image1; u16 image2; u32 size; size x { u32 p1; u32 p2 } }

#include "../common/usable_cpus.h"
#include "pairs_store.h"

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <cstring>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

struct Reader {
    FILE *f;
    bool ok = true;
    template <class T> bool get(T *v, size_t n = 1)
    {
        if (fread(v, sizeof(T), n, f) != n) { ok = false; return false; }
        return true;
    }
};

} // namespace

void frog_pairs::build_links()
{
    const uint64_t P = num_points();
    row_ptr.assign(P + 1, 0);
    const size_t nb = block_image1.size();
    // A point's links are ordered as readPairs push_back()s them: file block order.  One thread per IMAGE walks the
    // blocks that image takes part in, in file order, and touches only its own points: same order, no two threads on
    // one row (100 images: 0.8 s -> 0.15 s for 10^8 links).
    bool self_block = false;                                       // a block of an image with itself interleaves its two
    for (size_t b = 0; b < nb; b++) self_block = self_block || block_image1[b] == block_image2[b];   // sides pair by pair: serial
    if (self_block || nb > 0x7FFFFFFFull) {
        for (size_t b = 0; b < nb; b++) {
            const uint64_t o1 = point_offset[block_image1[b]], o2 = point_offset[block_image2[b]];
            for (uint64_t k = block_ptr[b]; k < block_ptr[b + 1]; k++) { row_ptr[o1 + p1[k] + 1]++; row_ptr[o2 + p2[k] + 1]++; }
        }
        for (uint64_t p = 0; p < P; p++) row_ptr[p + 1] += row_ptr[p];
        link_image.resize(row_ptr[P]);
        link_point.resize(row_ptr[P]);
        std::vector<uint64_t> cur(row_ptr.begin(), row_ptr.end() - 1);
        for (size_t b = 0; b < nb; b++) {                          // fill in file order: this IS the push_back order of readPairs
            const uint16_t i1 = block_image1[b], i2 = block_image2[b];
            const uint64_t o1 = point_offset[i1], o2 = point_offset[i2];
            for (uint64_t k = block_ptr[b]; k < block_ptr[b + 1]; k++) {
                const uint64_t a = cur[o1 + p1[k]]++;
                link_image[a] = i2; link_point[a] = p2[k];
                const uint64_t c = cur[o2 + p2[k]]++;
                link_image[c] = i1; link_point[c] = p1[k];
            }
        }
        return;
    }
    const auto t_in = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (std::getenv("FROG_TIMING")) std::printf("[timing] build_links, %s : %gs\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_in).count());
    };
    std::vector<std::vector<uint32_t>> blocks_of(n_images);        // block index * 2 + side (0: as image1, 1: as image2)
    for (size_t b = 0; b < nb; b++) {
        blocks_of[block_image1[b]].push_back((uint32_t)(2 * b));
        blocks_of[block_image2[b]].push_back((uint32_t)(2 * b + 1));
    }
    #pragma omp parallel for schedule(dynamic, 1) num_threads(frog::host_threads())
    for (int i = 0; i < (int)n_images; i++) {
        const uint64_t o = point_offset[i];
        for (uint32_t e : blocks_of[i]) {
            const size_t b = e >> 1;
            const frog_bulk<uint32_t> &mine = (e & 1) ? p2 : p1;
            for (uint64_t k = block_ptr[b]; k < block_ptr[b + 1]; k++) row_ptr[o + mine[k] + 1]++;
        }
    }
    lap("counted");
    for (uint64_t p = 0; p < P; p++) row_ptr[p + 1] += row_ptr[p];
    const uint64_t L = row_ptr[P];
    link_image.resize(L);
    link_point.resize(L);
    std::vector<uint64_t> cursor(row_ptr.begin(), row_ptr.end() - 1);
    #pragma omp parallel for schedule(dynamic, 1) num_threads(frog::host_threads())
    for (int i = 0; i < (int)n_images; i++) {
        const uint64_t o = point_offset[i];
        for (uint32_t e : blocks_of[i]) {
            const size_t b = e >> 1;
            const bool second = e & 1;
            const frog_bulk<uint32_t> &mine = second ? p2 : p1, &other = second ? p1 : p2;
            const uint16_t partner = second ? block_image1[b] : block_image2[b];
            for (uint64_t k = block_ptr[b]; k < block_ptr[b + 1]; k++) {
                const uint64_t a = cursor[o + mine[k]]++;
                link_image[a] = partner; link_point[a] = other[k];
            }
        }
    }
    lap("filled");
}

extern "C" {

frog_pairs *frog_pairs_read(const char *path, int *status)
{
    if (status) *status = FROG_OK;
    FILE *f = fopen(path, "rb");
    if (!f) { if (status) *status = FROG_E_INVALID; return nullptr; }
    // counts read from the file are checked against what is left of it before anything is allocated
    fseek(f, 0, SEEK_END);
    const long long file_size = ftell(f);
    fseek(f, 0, SEEK_SET);
    auto remaining = [&]() { return (unsigned long long)std::max(0LL, file_size - (long long)ftell(f)); };
    Reader r{ f };
    frog_pairs *p = new frog_pairs;
    uint16_t n = 0;
    r.get(&n);
    p->n_images = n;
    p->point_offset.assign(1, 0);
    for (uint32_t i = 0; i < n && r.ok; i++) {
        uint16_t len = 0;
        r.get(&len);
        std::string name(len, '\0');
        if (len) r.get(&name[0], len);
        p->names.push_back(name);
        double t[3] = { 0, 0, 0 };
        r.get(t, 3);
        p->ref_translation.insert(p->ref_translation.end(), t, t + 3);
        uint32_t np = 0;
        r.get(&np);
        if (!r.ok) break;
        if ((unsigned long long)np * 24ull > remaining()) { r.ok = false; break; }
        if ((unsigned long long)p->point_offset.back() + np > 0xFFFFFFFFull) { r.ok = false; break; }
        std::vector<float> rec((size_t)np * 6);
        if (np) r.get(rec.data(), rec.size());
        const size_t base = p->xyz.size();
        p->xyz.resize(base + (size_t)np * 3);
        p->other.resize(base + (size_t)np * 3);
        for (size_t k = 0; k < np; k++) {
            std::memcpy(&p->xyz[base + 3 * k], &rec[6 * k], 3 * sizeof(float));
            std::memcpy(&p->other[base + 3 * k], &rec[6 * k + 3], 3 * sizeof(float));
        }
        p->point_offset.push_back(p->point_offset.back() + np);
    }
    if (!r.ok) { fclose(f); delete p; if (status) *status = FROG_E_INVALID; return nullptr; }

    // The pair blocks are most of the file (8 bytes per pair: 0.4 GB for the 100-image benchmark group).  Round 6: the rest of
    // the file in one read, the block headers walked serially (a few thousand), the pairs checked and split into p1 / p2 on
    // all host threads into arrays sized beforehand -- one fread and two push_back()s per pair took 0.3 s of bin/frog's 2.2.
    p->block_ptr.assign(1, 0);
    int err = FROG_OK;
    // the blocks are read in place from a mapping of the file (no 0.4 GB copy through a buffer that first has to be zeroed);
    // a file that cannot be mapped (a pipe, a file system without mmap) goes through a buffer as before
    static const bool timing = std::getenv("FROG_TIMING") != nullptr;
    const auto t_blocks = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (timing) std::printf("[timing] readPairs, %s : %gs\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_blocks).count());
    };
    const size_t rest_at = (size_t)ftell(f), rest_size = (size_t)remaining();
    struct Span {
        const unsigned char *p = nullptr; size_t n = 0;
        void *map = nullptr; size_t map_len = 0; std::vector<unsigned char> buf;
        size_t size() const { return n; }
        const unsigned char &operator[](size_t i) const { return p[i]; }
        ~Span() { if (map) munmap(map, map_len); }
    } rest;
    rest.n = rest_size;
    if (rest_size) {
        void *m = mmap(nullptr, (size_t)file_size, PROT_READ, MAP_PRIVATE, fileno(f), 0);
        if (m != MAP_FAILED) {
            rest.map = m; rest.map_len = (size_t)file_size; rest.p = static_cast<const unsigned char *>(m) + rest_at;
            (void)madvise(m, (size_t)file_size, MADV_SEQUENTIAL);
        } else {
            rest.buf.resize(rest_size);
            if (fread(rest.buf.data(), 1, rest_size, f) != rest_size) err = FROG_E_INVALID;
            rest.p = rest.buf.data();
        }
    }
    std::vector<size_t> block_data;                            // offset of every block's first pair in `rest`
    for (size_t at = 0; !err && at < rest.size();) {
        if (rest.size() - at < 2) break;                       // (a lone trailing byte: fread of image1 fails upstream too -> EOF)
        uint16_t i1, i2;
        uint32_t size = 0;
        if (rest.size() - at < 8) { err = FROG_E_INVALID; break; }
        std::memcpy(&i1, &rest[at], 2); std::memcpy(&i2, &rest[at + 2], 2); std::memcpy(&size, &rest[at + 4], 4);
        at += 8;
        if (!size) { err = FROG_E_INVALID; break; }            // imageGroup.cxx:1393-1398
        if (i1 >= n || i2 >= n) { err = FROG_E_INVALID; break; }
        if ((unsigned long long)size * 8ull > rest.size() - at) { err = FROG_E_INVALID; break; }
        block_data.push_back(at);
        at += (size_t)size * 8;
        p->block_image1.push_back(i1);
        p->block_image2.push_back(i2);
        p->block_ptr.push_back(p->block_ptr.back() + size);
    }
    if (!err) {
        p->p1.resize(p->block_ptr.back());
        p->p2.resize(p->block_ptr.back());
        int bad = 0;
        #pragma omp parallel for schedule(dynamic, 8) reduction(|| : bad) num_threads(frog::host_threads())
        for (long long b = 0; b < (long long)block_data.size(); b++) {
            const uint32_t n1 = p->point_offset[p->block_image1[b] + 1] - p->point_offset[p->block_image1[b]];
            const uint32_t n2 = p->point_offset[p->block_image2[b] + 1] - p->point_offset[p->block_image2[b]];
            const unsigned char *src = &rest[block_data[b]];
            const uint64_t o = p->block_ptr[b], size = p->block_ptr[b + 1] - o;
            for (uint64_t k = 0; k < size; k++) {
                uint32_t a[2];
                std::memcpy(a, src + 8 * k, 8);
                if (a[0] >= n1 || a[1] >= n2) { bad = 1; break; }
                p->p1[o + k] = a[0];
                p->p2[o + k] = a[1];
            }
        }
        if (bad) err = FROG_E_INVALID;
    }
    lap("pair blocks split");
    fclose(f);
    if (err) { delete p; if (status) *status = err; return nullptr; }
    p->build_links();
    lap("+ reference-order CSR");
    return p;
}

int frog_pairs_append_points(frog_pairs *p, uint32_t image, const float *xyz, uint32_t n)
{
    if (!p || image >= p->n_images || (n && !xyz)) return FROG_E_INVALID;
    if (!n) return FROG_OK;
    if (p->num_points() + n > 0xFFFFFFFFull) return FROG_E_INVALID;
    const size_t at = 3 * (size_t)p->point_offset[image + 1];
    p->xyz.insert(p->xyz.begin() + at, xyz, xyz + 3 * (size_t)n);
    p->other.insert(p->other.begin() + at, 3 * (size_t)n, 0.f);
    for (uint32_t i = image + 1; i <= p->n_images; i++) p->point_offset[i] += n;
    p->build_links();                       // point indices inside an image are unchanged; rows shift
    return FROG_OK;
}

int frog_pairs_set_points(frog_pairs *p, uint32_t image, const float *xyz)
{
    if (!p || image >= p->n_images || !xyz) return FROG_E_INVALID;
    const size_t at = 3 * (size_t)p->point_offset[image], n = 3 * (size_t)(p->point_offset[image + 1] - p->point_offset[image]);
    std::copy(xyz, xyz + n, p->xyz.begin() + at);
    return FROG_OK;
}

int frog_pairs_write(const frog_pairs *p, const char *path)
{
    FILE *f = fopen(path, "wb");
    if (!f) return FROG_E_INVALID;
    uint16_t n = (uint16_t)p->n_images;
    fwrite(&n, sizeof n, 1, f);
    for (uint32_t i = 0; i < p->n_images; i++) {
        const std::string &name = p->names[i];
        uint16_t len = (uint16_t)name.size();
        fwrite(&len, sizeof len, 1, f);
        fwrite(name.data(), 1, len, f);
        fwrite(&p->ref_translation[3 * (size_t)i], sizeof(double), 3, f);
        uint32_t np = p->point_offset[i + 1] - p->point_offset[i];
        fwrite(&np, sizeof np, 1, f);
        std::vector<float> rec((size_t)np * 6);
        const size_t base = 3 * (size_t)p->point_offset[i];
        for (size_t k = 0; k < np; k++) {
            std::memcpy(&rec[6 * k], &p->xyz[base + 3 * k], 3 * sizeof(float));
            std::memcpy(&rec[6 * k + 3], &p->other[base + 3 * k], 3 * sizeof(float));
        }
        fwrite(rec.data(), sizeof(float), rec.size(), f);
    }
    for (size_t b = 0; b < p->block_image1.size(); b++) {
        uint16_t i1 = p->block_image1[b], i2 = p->block_image2[b];
        uint32_t size = (uint32_t)(p->block_ptr[b + 1] - p->block_ptr[b]);
        fwrite(&i1, sizeof i1, 1, f);
        fwrite(&i2, sizeof i2, 1, f);
        fwrite(&size, sizeof size, 1, f);
        std::vector<uint32_t> rec((size_t)size * 2);
        for (size_t k = 0; k < size; k++) {
            rec[2 * k] = p->p1[p->block_ptr[b] + k];
            rec[2 * k + 1] = p->p2[p->block_ptr[b] + k];
        }
        fwrite(rec.data(), sizeof(uint32_t), rec.size(), f);
    }
    int rc = ferror(f) ? FROG_E_INVALID : FROG_OK;
    fclose(f);
    return rc;
}

void frog_pairs_free(frog_pairs *p) { delete p; }

int frog_host_threads(void) { return frog::host_threads(); }

void frog_pairs_model(const frog_pairs *p, frog_model *out)
{
    out->n_images = p->n_images;
    out->point_offset = p->point_offset.data();
    out->xyz = p->xyz.data();
    out->row_ptr = p->row_ptr.data();
    out->link_image = p->link_image.data();
    out->link_point = p->link_point.data();
}

uint64_t frog_pairs_num_pairs(const frog_pairs *p) { return p->num_pairs(); }
uint64_t frog_pairs_num_points(const frog_pairs *p) { return p->num_points(); }
uint32_t frog_pairs_num_images(const frog_pairs *p) { return p->n_images; }
uint32_t frog_pairs_num_blocks(const frog_pairs *p) { return (uint32_t)p->block_image1.size(); }

int frog_pairs_block(const frog_pairs *p, uint32_t b, uint16_t *image1, uint16_t *image2,
                     uint32_t *size, const uint32_t **pp1, const uint32_t **pp2)
{
    if (b >= p->block_image1.size()) return FROG_E_INVALID;
    if (image1) *image1 = p->block_image1[b];
    if (image2) *image2 = p->block_image2[b];
    if (size) *size = (uint32_t)(p->block_ptr[b + 1] - p->block_ptr[b]);
    if (pp1) *pp1 = p->p1.data() + p->block_ptr[b];
    if (pp2) *pp2 = p->p2.data() + p->block_ptr[b];
    return FROG_OK;
}

frog_pairs *frog_pairs_from_arrays(uint32_t n_images, const uint32_t *point_offset,
                                   const float *xyz, const float *other,
                                   uint32_t n_blocks, const uint16_t *bi1, const uint16_t *bi2,
                                   const uint64_t *block_ptr, const uint32_t *p1, const uint32_t *p2)
{
    frog_pairs *p = new frog_pairs;
    p->n_images = n_images;
    p->point_offset.assign(point_offset, point_offset + n_images + 1);
    const uint64_t P = p->point_offset.back();
    p->xyz.assign(xyz, xyz + 3 * P);
    if (other) p->other.assign(other, other + 3 * P); else p->other.assign(3 * P, 0.f);
    p->ref_translation.assign(3 * (size_t)n_images, 0.0);
    for (uint32_t i = 0; i < n_images; i++) p->names.push_back("image" + std::to_string(i));
    p->block_image1.assign(bi1, bi1 + n_blocks);
    p->block_image2.assign(bi2, bi2 + n_blocks);
    p->block_ptr.assign(block_ptr, block_ptr + n_blocks + 1);
    const uint64_t np = p->block_ptr.back();
    p->p1.assign(p1, p1 + np);
    p->p2.assign(p2, p2 + np);
    // validate indices
    for (uint32_t b = 0; b < n_blocks; b++) {
        if (bi1[b] >= n_images || bi2[b] >= n_images) { delete p; return nullptr; }
        const uint32_t n1 = p->point_offset[bi1[b] + 1] - p->point_offset[bi1[b]];
        const uint32_t n2 = p->point_offset[bi2[b] + 1] - p->point_offset[bi2[b]];
        for (uint64_t k = block_ptr[b]; k < block_ptr[b + 1]; k++)
            if (p1[k] >= n1 || p2[k] >= n2) { delete p; return nullptr; }
    }
    p->build_links();
    return p;
}

} // extern "C"

// keypoints_io.cpp -- surf3d keypoint files, as the reference's `match` reads them
// (match/match.cpp:48-83 readCSVGZ, :117-146 readCSV, :149-179 readBinary) and writes them
// (:85-114 writeCSV).  One row per keypoint: x, y, z, scale, laplacianSign, response,
// descriptor values.
#include "frog_host.h"

#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <system_error>
#include <vector>
#include <zlib.h>

struct frog_keypoint_file {
    uint32_t dim = 0;
    std::vector<float> xyz, scale, laplacian, response, desc;
};

namespace {

// one CSV line -> values; stops at a cell starting with '\r' like upstream (`(int)cell[0] != 13`)
void parse_line(const char *line, std::vector<float> &vals)
{
    vals.clear();
    const char *p = line;
    while (*p && *p != '\n') {
        if (*p == '\r') break;
        // std::stof(cell) = strtof.  A plain decimal number (the only thing the detector writes) goes through std::from_chars:
        // correctly rounded like glibc's strtof, so the same float, at a fifth of the cost (1.08 M values per image: 65 of the
        // 110 ms a 20 000-keypoint csv.gz took to read).  Anything else -- leading blanks, '+', hex, inf / nan, a value out of
        // float's range -- keeps strtof and its answer.
        float v = 0.f;
        const char *end = p;
        bool fast = false;
        if ((*p >= '0' && *p <= '9') || *p == '-' || *p == '.') {
            const char *q = p;
            while (*q && *q != ',' && *q != '\n' && *q != '\r') q++;
            const auto r = std::from_chars(p, q, v, std::chars_format::general);
            // (strtof would read "0x1p3" as hex and "1e5x" up to the x: from_chars stops at the same places for decimals; a cell it
            // does not consume entirely takes the slow path, which decides)
            if (r.ec == std::errc() && r.ptr == q) { end = q; fast = true; }
        }
        if (!fast) {
            char *e2 = nullptr;
            v = std::strtof(p, &e2);
            end = e2;
        }
        if (end == p) break;
        vals.push_back(v);
        p = end;
        while (*p && *p != ',' && *p != '\n') p++;      // rest of the cell
        if (*p == ',') p++;
    }
}

void push_row(frog_keypoint_file &f, const std::vector<float> &v)
{
    if (v.size() <= 6) return;                          // `if ( count > 6 ) points->push_back( row )`
    const uint32_t dim = (uint32_t)v.size() - 6;
    if (f.scale.empty()) f.dim = dim;
    f.xyz.insert(f.xyz.end(), v.begin(), v.begin() + 3);
    f.scale.push_back(v[3]); f.laplacian.push_back(v[4]); f.response.push_back(v[5]);
    // a ragged row keeps the file's descriptor length (upstream would carry a ragged vector)
    for (uint32_t k = 0; k < f.dim; k++) f.desc.push_back(k < dim ? v[6 + k] : 0.f);
}

bool ends_with(const std::string &s, const char *suffix)
{
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

} // namespace

extern "C" {

frog_keypoint_file *frog_keypoints_read(const char *path, int *status)
{
    auto fail = [&](int code) { if (status) *status = code; return (frog_keypoint_file *)nullptr; };
    if (!path) return fail(FROG_E_INVALID);
    const std::string p(path);
    frog_keypoint_file *f = new frog_keypoint_file;
    std::vector<float> vals;
    if (ends_with(p, ".bin")) {                         // match.cpp:149-179
        FILE *in = std::fopen(path, "rb");
        if (!in) { delete f; return fail(FROG_E_IO); }
        f->dim = 48;
        float head[6] = { 0, 0, 0, 0, 0, 0 }, valF = 0;
        // upstream loops on !feof(file): after the last row one more pass runs whose freads fail, which
        // appends a point made of the last float read (the previous response) and a zero descriptor
        while (!std::feof(in)) {
            float desc[48] = {};
            for (int k = 0; k < 6; k++) { if (std::fread(&valF, sizeof(float), 1, in) != 1) { /* keeps valF */ } head[k] = valF; }
            (void)!std::fread(desc, sizeof(float), 48, in);
            f->xyz.insert(f->xyz.end(), head, head + 3);
            f->scale.push_back(head[3]); f->laplacian.push_back(head[4]); f->response.push_back(head[5]);
            f->desc.insert(f->desc.end(), desc, desc + 48);
        }
        std::fclose(in);
    } else if (ends_with(p, ".gz")) {                   // :48-83
        gzFile in = gzopen(path, "rb");
        if (!in) { delete f; return fail(FROG_E_IO); }
        (void)gzbuffer(in, 1u << 20);                   // zlib's default is 8 KiB per inflate call
        std::string line;
        char buf[65536];
        while (gzgets(in, buf, sizeof buf)) {
            line += buf;
            if (line.empty() || line.back() != '\n') { if (!gzeof(in)) continue; }
            parse_line(line.c_str(), vals);
            push_row(*f, vals);
            line.clear();
        }
        gzclose(in);
    } else if (ends_with(p, ".csv")) {                  // :117-146
        FILE *in = std::fopen(path, "r");
        if (!in) { delete f; return fail(FROG_E_IO); }
        std::string line;
        char buf[65536];
        while (std::fgets(buf, sizeof buf, in)) {
            line += buf;
            if ((line.empty() || line.back() != '\n') && !std::feof(in)) continue;   // a NUL byte yields an empty piece
            parse_line(line.c_str(), vals);
            push_row(*f, vals);
            line.clear();
        }
        std::fclose(in);
    } else {
        delete f;
        return fail(FROG_E_INVALID);                    // "Bad file format"
    }
    if (status) *status = FROG_OK;
    return f;
}

void frog_keypoints_free(frog_keypoint_file *f) { delete f; }

uint32_t frog_keypoints_count(const frog_keypoint_file *f) { return f ? (uint32_t)f->scale.size() : 0; }

void frog_keypoints_view(const frog_keypoint_file *f, frog_keypoints *out)
{
    if (!f || !out) return;
    out->n = (uint32_t)f->scale.size();
    out->dim = f->dim;
    out->xyz = f->xyz.data(); out->scale = f->scale.data(); out->laplacian = f->laplacian.data();
    out->response = f->response.data(); out->desc = f->desc.data();
}

// keep the rows listed in `keep` (in that order): the pruning steps of match.cpp:548-590
int frog_keypoints_select(frog_keypoint_file *f, const uint32_t *keep, uint32_t n_keep)
{
    if (!f || (n_keep && !keep)) return FROG_E_INVALID;
    frog_keypoint_file g;
    g.dim = f->dim;
    for (uint32_t i = 0; i < n_keep; i++) {
        const uint32_t p = keep[i];
        if (p >= f->scale.size()) return FROG_E_INVALID;
        g.xyz.insert(g.xyz.end(), f->xyz.begin() + 3 * (size_t)p, f->xyz.begin() + 3 * (size_t)p + 3);
        g.scale.push_back(f->scale[p]); g.laplacian.push_back(f->laplacian[p]); g.response.push_back(f->response[p]);
        g.desc.insert(g.desc.end(), f->desc.begin() + (size_t)p * f->dim, f->desc.begin() + (size_t)(p + 1) * f->dim);
    }
    *f = std::move(g);
    return FROG_OK;
}

// writeCSV, match.cpp:85-114 (default ostream formatting); ".gz" compresses, ".bin" writes readBinary's layout
int frog_keypoints_write(const char *path, const frog_keypoints *k)
{
    if (!path || !k) return FROG_E_INVALID;
    const std::string p(path);
    if (ends_with(p, ".bin")) {
        if (k->dim != 48) return FROG_E_INVALID;
        FILE *out = std::fopen(path, "wb");
        if (!out) return FROG_E_IO;
        for (uint32_t i = 0; i < k->n; i++) {
            const float head[6] = { k->xyz[3 * (size_t)i], k->xyz[3 * (size_t)i + 1], k->xyz[3 * (size_t)i + 2],
                                    k->scale[i], k->laplacian[i], k->response[i] };
            std::fwrite(head, sizeof(float), 6, out);
            std::fwrite(k->desc + (size_t)i * 48, sizeof(float), 48, out);
        }
        return std::fclose(out) == 0 ? FROG_OK : FROG_E_IO;
    }
    std::string text;
    char num[64];
    auto put = [&](float v, char sep) { std::snprintf(num, sizeof num, "%.9g", (double)v); text += num; text += sep; };
    for (uint32_t i = 0; i < k->n; i++) {
        for (int c = 0; c < 3; c++) put(k->xyz[3 * (size_t)i + c], ',');
        put(k->scale[i], ','); put(k->laplacian[i], ','); put(k->response[i], ',');
        for (uint32_t d = 0; d < k->dim; d++) put(k->desc[(size_t)i * k->dim + d], d + 1 < k->dim ? ',' : '\n');
    }
    if (ends_with(p, ".gz")) {
        gzFile out = gzopen(path, "wb6");
        if (!out) return FROG_E_IO;
        const bool ok = text.empty() || gzwrite(out, text.data(), (unsigned)text.size()) == (int)text.size();
        return (gzclose(out) == Z_OK && ok) ? FROG_OK : FROG_E_IO;
    }
    FILE *out = std::fopen(path, "w");
    if (!out) return FROG_E_IO;
    const bool ok = std::fwrite(text.data(), 1, text.size(), out) == text.size();
    return (std::fclose(out) == 0 && ok) ? FROG_OK : FROG_E_IO;
}

}

// transform_io.cpp -- reader of the transform files frog writes (transforms/<i>.json, with the
// coefficients inline or in .nii.gz sidecars), i.e. what tools/transformIO.h:375-460
// (readJSONfromString) rebuilds as a vtkGeneralTransform, here as a frog_chain_link list
// (include/frog_chain.h).  Also the voxel grid of a NIfTI-1 / MetaImage volume, for
// CheckDiffeomorphism's sampling grid.
#include "frog_host.h"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include <zlib.h>

namespace {

// ---- a small JSON reader (objects, arrays, strings, numbers, literals) ----------------
struct JValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    double num = 0;
    bool b = false;
    std::string str;
    std::vector<JValue> arr;
    std::map<std::string, JValue> obj;
    const JValue *get(const char *key) const
    {
        auto it = obj.find(key);
        return it == obj.end() ? nullptr : &it->second;
    }
};

struct JParser {
    const char *p, *end;
    bool ok = true;
    void ws() { while (p < end && std::isspace((unsigned char)*p)) p++; }
    bool eat(char c) { ws(); if (p < end && *p == c) { p++; return true; } return false; }
    JValue value()
    {
        JValue v;
        ws();
        if (p >= end) { ok = false; return v; }
        if (*p == '{') {
            p++; v.kind = JValue::Object;
            if (eat('}')) return v;
            do {
                ws();
                JValue k = value();
                if (k.kind != JValue::String || !eat(':')) { ok = false; return v; }
                v.obj[k.str] = value();
                if (!ok) return v;
            } while (eat(','));
            if (!eat('}')) ok = false;
        } else if (*p == '[') {
            p++; v.kind = JValue::Array;
            if (eat(']')) return v;
            do { v.arr.push_back(value()); if (!ok) return v; } while (eat(','));
            if (!eat(']')) ok = false;
        } else if (*p == '"') {
            p++; v.kind = JValue::String;
            while (p < end && *p != '"') {
                if (*p == '\\' && p + 1 < end) {
                    p++;
                    switch (*p) { case 'n': v.str += '\n'; break; case 't': v.str += '\t'; break; case 'u': p += 4; v.str += '?'; break; default: v.str += *p; }
                    p++;
                } else v.str += *p++;
            }
            if (p >= end) ok = false; else p++;
        } else if (!std::strncmp(p, "null", 4)) { p += 4; }
        else if (!std::strncmp(p, "true", 4)) { p += 4; v.kind = JValue::Bool; v.b = true; }
        else if (!std::strncmp(p, "false", 5)) { p += 5; v.kind = JValue::Bool; }
        else {
            char *e = nullptr;
            v.num = std::strtod(p, &e);
            if (e == p) { ok = false; return v; }
            v.kind = JValue::Number; p = e;
        }
        return v;
    }
};

bool ends_with(const std::string &s, const char *suffix)
{
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

bool read_all(const std::string &path, std::vector<unsigned char> &out)
{
    out.clear();
    if (ends_with(path, ".gz")) {
        gzFile f = gzopen(path.c_str(), "rb");
        if (!f) return false;
        unsigned char buf[1 << 16];
        int n;
        while ((n = gzread(f, buf, sizeof buf)) > 0) out.insert(out.end(), buf, buf + n);
        gzclose(f);
        return n == 0;
    }
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    std::fclose(f);
    return true;
}

template <class T> T at(const std::vector<unsigned char> &raw, size_t off) { T v; std::memcpy(&v, raw.data() + off, sizeof(T)); return v; }

// NIfTI-1 header fields the reference's readers use: dims, pixdim spacing, qform offsets as origin
bool nifti_geometry(const std::vector<unsigned char> &raw, uint32_t dims[3], double spacing[3], double origin[3],
                    uint32_t *n_comp, size_t *vox_offset, int *datatype)
{
    if (raw.size() < 352 || at<int32_t>(raw, 0) != 348) return false;
    const int16_t d0 = at<int16_t>(raw, 40);
    for (int k = 0; k < 3; k++) {
        dims[k] = (uint32_t)std::max<int>(1, at<int16_t>(raw, 42 + 2 * k));
        spacing[k] = at<float>(raw, 80 + 4 * k);
        origin[k] = at<float>(raw, 268 + 4 * k);
    }
    if (n_comp) *n_comp = d0 >= 5 ? (uint32_t)std::max<int>(1, at<int16_t>(raw, 50)) : 1u;
    if (vox_offset) *vox_offset = (size_t)at<float>(raw, 108);
    if (datatype) *datatype = at<int16_t>(raw, 70);
    return true;
}

} // namespace

struct frog_transform_file {
    std::vector<frog_chain_link> links;
    std::vector<std::unique_ptr<std::vector<float>>> storage;
};

extern "C" {

frog_transform_file *frog_transform_read(const char *json_path, int *status)
{
    auto fail = [&](int code) { if (status) *status = code; return (frog_transform_file *)nullptr; };
    if (!json_path) return fail(FROG_E_INVALID);
    std::ifstream in(json_path, std::ios::binary);
    if (!in) return fail(FROG_E_IO);
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string text = ss.str();
    JParser jp{ text.data(), text.data() + text.size() };
    const JValue root = jp.value();
    const JValue *list = root.get("transforms");
    if (!jp.ok || !list || list->kind != JValue::Array) return fail(FROG_E_INVALID);
    std::string dir(json_path);
    const size_t slash = dir.find_last_of("/\\");
    dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
    std::unique_ptr<frog_transform_file> f(new frog_transform_file);
    for (const JValue &t : list->arr) {
        const JValue *type = t.get("type");
        if (!type || type->kind != JValue::String) return fail(FROG_E_INVALID);
        frog_chain_link l;
        std::memset(&l, 0, sizeof l);
        if (type->str == "vtkMatrixToLinearTransform") {                       // transformIO.h:386-398
            const JValue *m = t.get("matrix");
            if (!m || m->arr.size() != 16) return fail(FROG_E_INVALID);
            l.type = FROG_T_LINEAR;
            for (int k = 0; k < 16; k++) l.matrix[k] = m->arr[k].num;
        } else if (type->str == "vtkBSplineTransform") {                       // :400-455
            l.type = FROG_T_BSPLINE;
            f->storage.emplace_back(new std::vector<float>);
            std::vector<float> &co = *f->storage.back();
            const JValue *file = t.get("file");
            if (file && file->kind == JValue::String) {
                std::vector<unsigned char> raw;
                uint32_t nc = 0; size_t off = 0; int dt = 0;
                if (!read_all(dir + "/" + file->str, raw)) return fail(FROG_E_IO);
                if (!nifti_geometry(raw, l.dims, l.spacing, l.origin, &nc, &off, &dt) || dt != 16 || nc < 3) return fail(FROG_E_INVALID);
                const size_t G = (size_t)l.dims[0] * l.dims[1] * l.dims[2];
                if (raw.size() < off + G * nc * sizeof(float)) return fail(FROG_E_INVALID);
                co.resize(3 * G);
                for (uint32_t c = 0; c < 3; c++)                               // stored plane by plane
                    for (size_t v = 0; v < G; v++) co[3 * v + c] = at<float>(raw, off + (c * G + v) * sizeof(float));
            } else {
                const JValue *dm = t.get("dimensions"), *og = t.get("origin"), *sp = t.get("spacing"), *cf = t.get("coeffs");
                if (!dm || !og || !sp || !cf || dm->arr.size() != 3 || og->arr.size() != 3 || sp->arr.size() != 3) return fail(FROG_E_INVALID);
                for (int k = 0; k < 3; k++) { l.dims[k] = (uint32_t)dm->arr[k].num; l.origin[k] = og->arr[k].num; l.spacing[k] = sp->arr[k].num; }
                const size_t G = (size_t)l.dims[0] * l.dims[1] * l.dims[2];
                if (cf->arr.size() != 3 * G) return fail(FROG_E_INVALID);
                co.resize(3 * G);
                for (size_t k = 0; k < 3 * G; k++) co[k] = (float)cf->arr[k].num;
            }
            l.coeffs = co.data();
        } else {
            return fail(FROG_E_INVALID);                                        // "Error : transform type ... not supported"
        }
        f->links.push_back(l);
    }
    if (status) *status = FROG_OK;
    return f.release();
}

void frog_transform_free(frog_transform_file *f) { delete f; }
uint32_t frog_transform_num_links(const frog_transform_file *f) { return f ? (uint32_t)f->links.size() : 0; }
const frog_chain_link *frog_transform_links(const frog_transform_file *f) { return f && !f->links.empty() ? f->links.data() : nullptr; }

// voxel grid (dimensions, spacing, origin) of a volume: NIfTI-1 (.nii, .nii.gz) or MetaImage header (.mhd)
int frog_volume_geometry(const char *path, uint32_t dims[3], double spacing[3], double origin[3])
{
    if (!path || !dims || !spacing || !origin) return FROG_E_INVALID;
    const std::string p(path);
    if (ends_with(p, ".mhd")) {
        std::ifstream in(path);
        if (!in) return FROG_E_IO;
        std::string line;
        bool have = false;
        for (int k = 0; k < 3; k++) { spacing[k] = 1; origin[k] = 0; dims[k] = 1; }
        while (std::getline(in, line)) {
            const size_t eq = line.find('=');
            if (eq == std::string::npos) continue;
            std::string key = line.substr(0, eq);
            key.erase(key.find_last_not_of(" \t") + 1);
            std::stringstream vals(line.substr(eq + 1));
            if (key == "DimSize") { for (int k = 0; k < 3; k++) vals >> dims[k]; have = true; }
            else if (key == "ElementSpacing") { for (int k = 0; k < 3; k++) vals >> spacing[k]; }
            else if (key == "Offset" || key == "Position" || key == "Origin") { for (int k = 0; k < 3; k++) vals >> origin[k]; }
        }
        return have ? FROG_OK : FROG_E_INVALID;
    }
    std::vector<unsigned char> raw;
    if (!read_all(p, raw)) return FROG_E_IO;
    return nifti_geometry(raw, dims, spacing, origin, nullptr, nullptr, nullptr) ? FROG_OK : FROG_E_INVALID;
}

}

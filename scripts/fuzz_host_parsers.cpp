// fuzz_host_parsers.cpp -- the host-side file parsers (pairs.bin, keypoint files, transform JSON,
// NIfTI headers) on mutated and truncated inputs, to be built with -fsanitize=address,undefined:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fopenmp -Iinclude scripts/fuzz_host_parsers.cpp \
//       frog_amd/csrc/host/{pairs_store,keypoints_io,transform_io,nifti_out,synth,volume_io}.cpp -lz -o /tmp/fuzz_host
//   /tmp/fuzz_host /tmp/fuzzdir 3000
// Every parser must either succeed or fail with a status; no crash, no out-of-bounds access.
#include "frog_host.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

static std::vector<unsigned char> slurp(const std::string &p)
{
    std::ifstream in(p, std::ios::binary);
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}
static void spit(const std::string &p, const std::vector<unsigned char> &d)
{
    std::ofstream out(p, std::ios::binary | std::ios::trunc);
    out.write((const char *)d.data(), (std::streamsize)d.size());
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::printf("usage: fuzz_host dir iterations\n"); return 2; }
    const std::string dir = argv[1];
    const int iters = std::atoi(argv[2]);
    std::mt19937 rng(12345);

    // ---- seeds: valid files written by the library itself
    frog_synth_params sp;
    frog_synth_defaults(&sp);
    sp.n_images = 3; sp.points_per_image = 40; sp.pairs_per_block = 20;
    frog_pairs *pairs = frog_synth_generate(&sp);
    if (!pairs || frog_pairs_write(pairs, (dir + "/seed.bin").c_str())) { std::printf("cannot write seed pairs\n"); return 2; }
    frog_pairs_free(pairs);
    {
        std::vector<float> xyz(30), sc(10, 1.5f), lap(10, 1.f), rsp(10, 0.5f), desc(480, 0.25f);
        frog_keypoints k{ 10, 48, xyz.data(), sc.data(), lap.data(), rsp.data(), desc.data() };
        frog_keypoints_write((dir + "/seed.csv").c_str(), &k);
        frog_keypoints_write((dir + "/seed.csv.gz").c_str(), &k);
        frog_keypoints_write((dir + "/seed.bin2.bin").c_str(), &k);
        const uint32_t dims[3] = { 4, 3, 5 };
        const double s3[3] = { 1, 2, 3 }, o3[3] = { -1, 0, 1 };
        std::vector<float> vox(4 * 3 * 5 * 3, 0.5f);
        frog_nifti_write((dir + "/seed.json.0.nii.gz").c_str(), dims, s3, o3, 3, vox.data());
        frog_nifti_write((dir + "/seed.nii").c_str(), dims, s3, o3, 3, vox.data());
        std::ofstream js(dir + "/seed.json");
        js << "{\"transforms\":[{\"type\":\"vtkMatrixToLinearTransform\",\"matrix\":[1,0,0,0,0,1,0,0,0,0,1,0,0,0,0,1]},"
              "{\"type\":\"vtkBSplineTransform\",\"file\":\"seed.json.0.nii.gz\"},"
              "{\"type\":\"vtkBSplineTransform\",\"dimensions\":[2,2,2],\"origin\":[0,0,0],\"spacing\":[1,1,1],\"coeffs\":["
              "0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0]}]}";
    }
    {
        std::vector<int16_t> vox(6 * 5 * 4, 7);
        frog_volume v{ { 6, 5, 4 }, { 1, 2, 3 }, { -1, 0, 1 }, FROG_V_I16, vox.data() };
        frog_volume_write((dir + "/vol.nii.gz").c_str(), &v);
        frog_volume_write((dir + "/vol.nii").c_str(), &v);
        frog_volume_write((dir + "/vol.mhd").c_str(), &v);
    }
    struct Target { std::string seed, out; int kind; };
    const std::vector<Target> targets = {
        { dir + "/seed.bin", dir + "/m.bin", 0 }, { dir + "/seed.csv", dir + "/m.csv", 1 }, { dir + "/seed.csv.gz", dir + "/m.csv.gz", 1 },
        { dir + "/seed.bin2.bin", dir + "/m2.bin", 1 }, { dir + "/seed.json", dir + "/m.json", 2 }, { dir + "/seed.nii", dir + "/m.nii", 3 },
        { dir + "/seed.json.0.nii.gz", dir + "/m.json.0.nii.gz", 4 },
        { dir + "/vol.nii.gz", dir + "/mv.nii.gz", 5 }, { dir + "/vol.mhd", dir + "/mv.mhd", 5 }, { dir + "/vol.zraw", dir + "/mv.zraw", 6 },
        { dir + "/vol.nii", dir + "/mv.nii", 5 },
    };
    long ok = 0, rejected = 0;
    for (int it = 0; it < iters; it++) {
        const Target &t = targets[it % targets.size()];
        std::vector<unsigned char> d = slurp(t.seed);
        const int mode = (int)(rng() % 4);
        if (mode == 0 && !d.empty()) d.resize(rng() % d.size());                       // truncate
        else if (mode == 1) for (int k = 0; k < 8 && !d.empty(); k++) d[rng() % d.size()] = (unsigned char)rng();   // flip bytes
        else if (mode == 2 && d.size() > 16) { const size_t a = rng() % (d.size() - 8); for (int k = 0; k < 8; k++) d[a + k] = 0xFF; }   // huge counts
        else if (mode == 3) for (int k = 0; k < 64; k++) d.push_back((unsigned char)rng());   // trailing garbage
        spit(t.out, d);
        int status = 0;
        if (t.kind == 0) {
            frog_pairs *p = frog_pairs_read(t.out.c_str(), &status);
            if (p) { frog_model m; frog_pairs_model(p, &m); (void)frog_pairs_num_pairs(p); frog_pairs_free(p); ok++; } else rejected++;
        } else if (t.kind == 1) {
            frog_keypoint_file *f = frog_keypoints_read(t.out.c_str(), &status);
            if (f) { frog_keypoints v; frog_keypoints_view(f, &v); (void)frog_keypoints_count(f); frog_keypoints_free(f); ok++; } else rejected++;
        } else if (t.kind == 2 || t.kind == 4) {
            if (t.kind == 4) spit(dir + "/m.json", slurp(dir + "/seed.json"));           // valid JSON, mutated sidecar
            frog_transform_file *f = frog_transform_read((dir + "/m.json").c_str(), &status);
            if (f) { (void)frog_transform_links(f); frog_transform_free(f); ok++; } else rejected++;
        } else if (t.kind == 5 || t.kind == 6) {
            // a mutated header with the intact data file beside it, or the intact header with mutated data
            std::string header = t.out;
            if (t.kind == 6) { header = dir + "/mv.mhd"; }
            if (t.kind == 6 || t.out == dir + "/mv.mhd") {
                std::vector<unsigned char> h = slurp(t.kind == 6 ? dir + "/vol.mhd" : t.out);
                std::string text(h.begin(), h.end());
                const size_t at = text.find("vol.zraw");
                if (at != std::string::npos) text.replace(at, 8, "mv.zraw");
                spit(header, std::vector<unsigned char>(text.begin(), text.end()));
                if (t.kind == 5) spit(dir + "/mv.zraw", slurp(dir + "/vol.zraw"));
            }
            frog_volume_file *f = frog_volume_read(header.c_str(), &status);
            if (f) { frog_volume v; frog_volume_view(f, &v); double lo, hi; (void)frog_volume_range(&v, &lo, &hi); frog_volume_free(f); ok++; } else rejected++;
        } else {
            uint32_t dims[3]; double s3[3], o3[3];
            if (frog_volume_geometry(t.out.c_str(), dims, s3, o3) == 0) ok++; else rejected++;
        }
    }
    std::printf("fuzz: %ld accepted, %ld rejected, no crash\n", ok, rejected);
    return 0;
}

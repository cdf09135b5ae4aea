// match_main.cpp -- `match`: the reference's pairing tool on the GPU (match/match.cpp main, :338-747).
//
//   match pointFiles.txt|directory [options]
//
// Same flags (-n -sp -np -nt -d -d2 -zmin -zmax -o -p -anat -sym -targ), same input files, same
// pairs.bin, plus -transformPrefix (positions for the -anat test).  -all is upstream's matchAll, quirk included (include/frog_match.h).
// A directory is read in sorted name order (upstream: the file system's order).
#include "../common/usable_cpus.h"
#include "../common/bulk_alloc.h"
#include "frog_host.h"
#include "frog_match.h"
#include "frog_hip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace fs = std::filesystem;
using std::cout;
using std::endl;

extern "C" const char *frog_last_error(void);

int main(int argc, char *argv[])
{
    int N = 1000000;
    float sp = 0;
    int np = 1000000;
    if (argc < 2) {
        std::cout << "Usage : match pointFiles.txt [options] " << std::endl;
        return 1;
    }
    const fs::path full_path = fs::absolute(fs::path(argv[1]));
    float dist = 0.22f, dist2second = 1, zmin = -1e20f, zmax = 1e20f, anatVal = 0.0f;
    bool writePoints = false, symFlag = false, matchAll = false;
    char *outputFileName = 0;
    int target = -1, device = 0;
    char *transformPrefix = 0;
    int argumentsIndex = 2;
    while (argumentsIndex < argc) {                                         // match.cpp:367-435
        char *key = argv[argumentsIndex];
        char *value = argumentsIndex + 1 < argc ? argv[argumentsIndex + 1] : (char *)"";
        if (strcmp(key, "-n") == 0) N = atoi(value);
        if (strcmp(key, "-sp") == 0) sp = atof(value);
        if (strcmp(key, "-np") == 0) np = atoi(value);
        if (strcmp(key, "-nt") == 0) { /* host threads: the pairing runs on the GPU */ }
        if (strcmp(key, "-d") == 0) dist = atof(value);
        if (strcmp(key, "-d2") == 0) dist2second = atof(value);
        if (strcmp(key, "-zmin") == 0) zmin = atof(value);
        if (strcmp(key, "-zmax") == 0) zmax = atof(value);
        if (strcmp(key, "-o") == 0) outputFileName = value;
        if (strcmp(key, "-dev") == 0) device = atoi(value);
        if (strcmp(key, "-all") == 0) matchAll = true;                 // as -p: the next word is skipped too (:406-408, :433)
        if (strcmp(key, "-p") == 0) writePoints = true;
        if (strcmp(key, "-anat") == 0) anatVal = atof(value);
        if (strcmp(key, "-sym") == 0) { symFlag = true; argumentsIndex -= 1; }
        if (strcmp(key, "-targ") == 0) target = atoi(value);
        if (strcmp(key, "-transformPrefix") == 0) transformPrefix = value;
        argumentsIndex += 2;
    }

    std::vector<std::string> filenames;
    std::vector<std::array<double, 3>> rigids;
    if (fs::is_directory(full_path)) {                                      // :443-457
        for (const auto &e : fs::directory_iterator(full_path))
            if (fs::is_regular_file(e.status())) filenames.push_back(e.path().string());
        std::sort(filenames.begin(), filenames.end());
    } else if (fs::is_regular_file(full_path)) {                            // :459-494
        std::string line;
        std::ifstream file(full_path.string());
        while (std::getline(file, line)) {
            std::stringstream lineStream(line);
            std::string cell;
            std::getline(lineStream, cell, ',');
            if (cell.find("/") == 0) {
                filenames.push_back(cell);
                cout << cell << endl;
            } else {
                filenames.push_back(full_path.parent_path().string() + "/" + cell + ".csv");
                cout << full_path.parent_path().string() + cell << endl;
            }
            std::array<double, 3> point = { 0, 0, 0 };
            try {
                for (int k = 0; k < 3; k++) { std::getline(lineStream, cell, ','); point[k] = std::stof(cell); }
            } catch (...) {}
            rigids.push_back(point);
        }
    } else {
        std::cerr << "Bad argument, first arg must be a valid file or a directory" << endl;
        return 1;
    }

    // the HIP runtime comes up (0.05-0.2 s) on a thread of its own beside the reading of the keypoint files, as in bin/frog;
    // joined by an exit handler too, so that an early exit does not leave it in the middle of the runtime's start-up
    static std::thread warm;
    warm = std::thread([device] { (void)frog_device_warm(device); });
    std::atexit([] { if (warm.joinable()) warm.join(); });

    cout << "Found " << filenames.size() << " files, loading : " << fmin(N, filenames.size()) << endl;
    auto start = std::chrono::system_clock::now();
    if ((int)filenames.size() > N) filenames.resize(N);
    const int nb = (int)filenames.size();
    if (nb == 0) { std::cerr << "no keypoint file" << endl; return 1; }
    std::vector<frog_keypoint_file *> allPoints(nb, nullptr);
    bool bad = false;
    #pragma omp parallel for schedule(dynamic) num_threads(frog::host_threads())
    for (int it = 0; it < nb; ++it) {                                       // :509-556
        int status = 0;
        frog_keypoint_file *points = frog_keypoints_read(filenames[it].c_str(), &status);
        if (!points) {
            #pragma omp critical
            { std::cerr << "Bad file format or unreadable file : " << filenames[it] << endl; bad = true; }
            continue;
        }
        const float zT = rigids.size() ? (float)rigids[it][2] : 0;
        frog_keypoints v;
        frog_keypoints_view(points, &v);
        std::vector<uint32_t> keep;
        for (uint32_t p = 0; p < v.n; p++) {
            const float z = v.xyz[3 * (size_t)p + 2] + zT;
            if (!(z < zmin || z > zmax)) keep.push_back(p);
        }
        const uint32_t before = v.n;
        if (keep.size() != v.n) frog_keypoints_select(points, keep.data(), (uint32_t)keep.size());
        #pragma omp critical
        cout << "image " << it << " rigid : " << (rigids.size() ? rigids[it][0] : 0.0) << ", " << (rigids.size() ? rigids[it][1] : 0.0)
             << ", " << (rigids.size() ? rigids[it][2] : 0.0) << " before : " << before << " points, after : "
             << frog_keypoints_count(points) << endl << std::flush;
        allPoints[it] = points;
    }
    if (bad) return 1;
    auto end = std::chrono::system_clock::now();
    cout << " : " << std::chrono::duration<float>(end - start).count() << "s" << endl;
    start = end;
    {
        frog_keypoints v;
        frog_keypoints_view(allPoints[0], &v);
        cout << v.dim << " values per descriptor" << endl;
    }
    cout << "Sorting and pruning..." << endl;
    for (int it = 0; it < nb; ++it) {                                       // :566-596
        frog_keypoints v;
        frog_keypoints_view(allPoints[it], &v);
        std::vector<uint32_t> keep;
        for (uint32_t p = 0; p < v.n; p++)
            if (!(v.response[p] < sp)) keep.push_back(p);
        if ((int)keep.size() > np) {
            std::partial_sort(keep.begin(), keep.begin() + np, keep.end(),
                              [&v](uint32_t a, uint32_t b) { return v.response[a] > v.response[b]; });
            keep.resize(np);
        }
        if (keep.size() != v.n || (int)v.n > np) frog_keypoints_select(allPoints[it], keep.data(), (uint32_t)keep.size());
        cout << ". (" << frog_keypoints_count(allPoints[it]) << ")" << std::flush;
        if (writePoints) {
            std::stringstream outfilename;
            outfilename << "points" << it << ".csv";
            cout << " writing " << outfilename.str() << endl;
            frog_keypoints_view(allPoints[it], &v);
            frog_keypoints_write(outfilename.str().c_str(), &v);
        }
    }
    end = std::chrono::system_clock::now();
    cout << " : " << std::chrono::duration<float>(end - start).count() << "s" << endl;
    start = end;

    std::vector<uint16_t> first, second;                                    // :603-614
    for (int i = 0; i < nb - 1; i++) {
        if (target >= 0) {
            if (i != target) { first.push_back((uint16_t)i); second.push_back((uint16_t)target); }
        } else {
            for (int j = i + 1; j < nb; j++) { first.push_back((uint16_t)i); second.push_back((uint16_t)j); }
        }
    }

    cout << "Pairing... " << endl;
    std::vector<frog_keypoints> views(nb);
    for (int it = 0; it < nb; ++it) frog_keypoints_view(allPoints[it], &views[it]);
    // -transformPrefix (:517-558): the -anat test compares keypoint positions moved by <prefix><image>.json.
    // (Upstream transforms a COPY of every point -- `for ( auto point : *points )` -- so its test reads
    // transformedCoordinates that were never written; this does what the option is for.)  The chain is
    // applied link by link in float, as vtkGeneralTransform's float TransformPoint; pairs.bin keeps the
    // original coordinates.
    std::vector<frog_keypoints> matchViews(views);
    std::vector<std::vector<float>> moved(nb);
    if (transformPrefix) {
        for (int it = 0; it < nb; ++it) {
            const std::string transformFile = std::string(transformPrefix) + std::to_string(it) + ".json";
            cout << "Reading transform " << transformFile << endl;
            int status = 0;
            frog_transform_file *tf = frog_transform_read(transformFile.c_str(), &status);
            if (!tf) { cout << "Error : cannot read transform " << transformFile << endl; return 1; }
            const size_t n = views[it].n;
            moved[it].assign(views[it].xyz, views[it].xyz + 3 * n);
            std::vector<double> in(3 * n), out(3 * n);
            for (uint32_t l = 0; l < frog_transform_num_links(tf) && n; l++) {
                frog_chain *chain = nullptr;
                if (frog_chain_create(frog_transform_links(tf) + l, 1, device, &chain)) { cout << "Error : " << frog_last_error() << endl; return 1; }
                for (size_t k = 0; k < in.size(); k++) in[k] = moved[it][k];
                if (frog_chain_apply(chain, in.data(), out.data(), n)) { cout << "Error : " << frog_last_error() << endl; return 1; }
                for (size_t k = 0; k < in.size(); k++) moved[it][k] = (float)out[k];
                frog_chain_destroy(chain);
            }
            frog_transform_free(tf);
            matchViews[it].xyz = moved[it].data();
        }
    }
    const bool timing = std::getenv("FROG_TIMING") != nullptr;           // [timing] lines, as bin/frog's
    const auto t_pairing = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (timing) cout << endl << "[timing] " << what << " : " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pairing).count() << "s" << endl;
    };
    frog_matcher *m = nullptr;
    if (frog_matcher_create(matchViews.data(), (uint32_t)nb, device, &m)) { cout << "Error : " << frog_last_error() << endl; return 1; }
    lap("matcher created (keypoints sorted, operands built, uploaded)");
    frog_match_options o;
    frog_match_options_default(&o);
    o.threshold = dist; o.dist2second = dist2second; o.anat = anatVal; o.sym = symFlag ? 1 : 0; o.all = matchAll ? 1 : 0;
    std::vector<uint64_t> offset(first.size() + 1, 0);
    uint32_t *pa = nullptr, *pb = nullptr;
    if (frog_matcher_run(m, first.data(), second.data(), first.size(), &o, offset.data(), &pa, &pb)) {
        cout << "Error : " << frog_last_error() << endl;
        return 1;
    }
    lap("+ every image pair matched");
    for (size_t k = 0; k < first.size(); k++) cout << "." << std::flush;
    const uint64_t sum = offset[first.size()];
    end = std::chrono::system_clock::now();
    cout << " : " << std::chrono::duration<float>(end - start).count() << "s" << endl;
    cout << "Nb Match : " << sum << endl;

    std::stringstream outfilename;                                          // :668-682
    if (outputFileName) outfilename << outputFileName;
    else outfilename << "out_" << full_path.stem().string() << "_" << filenames.size() << ".bin";
    FILE *file = fopen(outfilename.str().c_str(), "wb");
    if (file == NULL) { cout << "write error : " << outfilename.str() << endl; exit(1); }
    unsigned short nbAcq = (unsigned short)filenames.size();
    fwrite(&nbAcq, sizeof(unsigned short), 1, file);
    for (int it = 0; it < nb; it++) {                                       // :686-722
        const size_t found = filenames[it].find_last_of("/\\");
        const std::string currFile = filenames[it].substr(found + 1);
        unsigned short sizeString = (unsigned short)currFile.size();
        fwrite(&sizeString, sizeof(unsigned short), 1, file);
        fwrite(currFile.c_str(), sizeof(char), currFile.size(), file);
        double tmp[3] = { 0, 0, 0 };
        if (rigids.size() > 0) for (int k = 0; k < 3; k++) tmp[k] = rigids[it][k];
        fwrite(tmp, sizeof(double), 3, file);
        const frog_keypoints &v = views[it];
        uint32_t nbPoints = v.n;
        fwrite(&nbPoints, sizeof(uint32_t), 1, file);
        // (one fwrite per image instead of four per keypoint: 8 M calls for the benchmark group)
        std::vector<float> rec((size_t)v.n * 6);
        for (uint32_t r = 0; r < v.n; r++) {
            std::memcpy(&rec[6 * (size_t)r], v.xyz + 3 * (size_t)r, 3 * sizeof(float));
            rec[6 * (size_t)r + 3] = v.scale[r]; rec[6 * (size_t)r + 4] = v.laplacian[r]; rec[6 * (size_t)r + 5] = v.response[r];
        }
        if (v.n) fwrite(rec.data(), sizeof(float), rec.size(), file);
    }
    // blocks i-major, j ascending (:724-742); jobs were generated in that order unless -targ is set
    std::vector<size_t> order(first.size());
    for (size_t k = 0; k < order.size(); k++) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) {
        return first[x] != first[y] ? first[x] < first[y] : second[x] < second[y]; });
    // The blocks as one image of the file's tail, filled on all host threads, written once: two 4-byte fwrite per pair were
    // 132 M library calls for the benchmark group's 66 M pairs -- 3 of bin/match's 5.3 s.
    {
        std::vector<uint64_t> at(order.size() + 1, 0);                     // byte offset of every block in the tail
        for (size_t n = 0; n < order.size(); n++) at[n + 1] = at[n] + 8 + 8 * (offset[order[n] + 1] - offset[order[n]]);
        frog::Bulk<unsigned char> tail(at[order.size()]);
        #pragma omp parallel for schedule(dynamic, 16) num_threads(frog::host_threads())
        for (long long n = 0; n < (long long)order.size(); n++) {
            const size_t k = order[n];
            const unsigned short i = first[k], j = second[k];
            const unsigned int size = (unsigned int)(offset[k + 1] - offset[k]);
            unsigned char *dst = tail.data() + at[n];
            std::memcpy(dst, &i, 2); std::memcpy(dst + 2, &j, 2); std::memcpy(dst + 4, &size, 4);
            uint32_t *rec = reinterpret_cast<uint32_t *>(dst + 8);          // 8-byte aligned: every block is a multiple of 8 bytes
            for (uint64_t r = 0; r < size; r++) { rec[2 * r] = pa[offset[k] + r]; rec[2 * r + 1] = pb[offset[k] + r]; }
        }
        if (!tail.empty() && fwrite(tail.data(), 1, tail.size(), file) != tail.size()) { cout << "write error : " << outfilename.str() << endl; exit(1); }
    }
    fclose(file);
    lap("+ pairs.bin written");
    cout << "Output file : " << outfilename.str() << endl;
    frog_match_free(pa); frog_match_free(pb);
    frog_matcher_destroy(m);
    for (auto *p : allPoints) frog_keypoints_free(p);
    return 0;
}

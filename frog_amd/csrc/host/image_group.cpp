// image_group.cpp -- control flow and file output of the groupwise solver on the
// host; all arithmetic is in libfrog_hip.so.  Follows ImageGroup::run
// (registration/imageGroup.cxx:31-157) step for step; stdout wording follows the
// reference because the DESK UI greps it (js/groupwiseDeformableRegistration.js:522-545).

#include "../common/usable_cpus.h"
#include "image_group.h"
#include "json_out.h"
#include "pairs_store.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <numeric>
#include <omp.h>
#include <zlib.h>
#include <sstream>
#include <dlfcn.h>

#include "../../../include/frog_comm.h"

#include "comm_api.h"

// libfrog_comm.so (RCCL) is loaded on demand, for -ng / -ngl and frog_run_schedule only: a process that brings its own
// communicator (the Python drivers over torch.distributed) uses this library without ever mapping a second RCCL.
bool CommApi::load(std::string &err)
{
    if (create_rccl) return true;
    Dl_info info;
    std::string dir;
    if (dladdr((void *)&frog_pairs_read, &info) && info.dli_fname) {
        dir = info.dli_fname;
        const size_t slash = dir.rfind('/');
        dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
    }
    void *h = dlopen((dir + "libfrog_comm.so").c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libfrog_comm.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) { err = dlerror(); return false; }
#define LOAD(field, name) field = (decltype(field))dlsym(h, name); if (!field) { err = std::string("missing symbol ") + name; return false; }
    LOAD(create_loopback, "frog_comm_create_loopback");
    LOAD(destroy_all, "frog_comm_destroy_all"); LOAD(bind, "frog_comm_bind");
    LOAD(all_gather_xyz2, "frog_comm_all_gather_xyz2"); LOAD(all_reduce, "frog_comm_all_reduce");
    LOAD(gather_points, "frog_comm_gather_points");
    LOAD(all_reduce_bounds, "frog_comm_all_reduce_bounds"); LOAD(barrier, "frog_comm_barrier");
    LOAD(timing, "frog_comm_timing"); LOAD(timing_read, "frog_comm_timing_read");
    LOAD(create_rccl, "frog_comm_create_rccl");
#undef LOAD
    return true;
}
CommApi &host_comm_api() { static CommApi api; return api; }
#define g_comm (host_comm_api())

using std::cout;
using std::endl;

#define print(v, size) { for (int i_ = 0; i_ < size; i_++) { cout << v[i_]; if (i_ < (size - 1)) cout << " "; } cout << endl; }

ImageGroup::ImageGroup()
{
    // imageGroup.h:52-82
    boundingBoxMargin = 0.1f;
    deformableAlpha = 0.02f;
    deformableIterations = 200;
    fixedTransformsDirectory = 0;
    guaranteeDiffeomorphism = true;
    invertLandmarksCoordinates = true;
    landmarksConstraintsWeight = 50;
    initialGridSize = 100;
    inlierThreshold = 0.5f;
    linearAlpha = 0.5f;
    linearInitializationAnchor[0] = linearInitializationAnchor[1] = linearInitializationAnchor[2] = 0.5f;
    linearIterations = 50;
    maxDisplacementRatio = 0.4f;
    deformableLevels = 3;
    numberOfFixedImages = 0;
    numberOfRANSACIterations = 5000;
    printLinear = false;
    printStats = false;
    RANSACInlierDistance = 50;
    RANSACMaxScale = 10;
    statIntervalUpdate = 10;
    useRANSAC = true;
    useScale = true;
    writePairs = false;
    writeSingleFileTransforms = false;
    transformSubdirectory = "transforms";
    errorMapsSubdirectory = "errorMaps";
    statsMaxSize = 10000;       // stats.cxx:10-12
    statsMaxIterations = 10000;
    statsEpsilon = 1e-6f;
}

ImageGroup::~ImageGroup()
{
    if (!comms.empty() && g_comm.destroy_all) g_comm.destroy_all((int)comms.size(), comms.data());
    if (!ctxs.empty()) { for (frog_ctx *c : ctxs) if (c) frog_destroy(c); }
    else if (ctx) frog_destroy(ctx);
    if (pairs && ownPairs) frog_pairs_free(pairs);
}

frog_ctx *ImageGroup::ctxOf(uint32_t image) const
{
    if (ctxs.empty()) return ctx;
    for (size_t r = 0; r + 1 < shardBegin.size(); r++)
        if (image >= shardBegin[r] && image < shardBegin[r + 1]) return ctxs[r];
    return ctx;
}

// Contiguous image ranges, one per rank, balanced by half-link count (the cost of every loop of the solver is
// proportional to the half-links of the images it walks); every rank gets at least one image.
void ImageGroup::planShards()
{
    frog_model m;
    frog_pairs_model(pairs, &m);
    const uint32_t nI = m.n_images;
    if ((uint32_t)nGpus > nI) { cout << "Error : more GPUs than images" << endl; exit(1); }
    const uint64_t total = m.row_ptr[m.point_offset[nI]];
    shardBegin.assign(1, 0);
    for (int r = 1; r < nGpus; r++) {
        const double target = (double)total * r / nGpus;
        uint32_t i = shardBegin.back() + 1;
        // nearest image boundary to the target, leaving room for the ranks on both sides
        while (i < nI - (uint32_t)(nGpus - r) && (double)m.row_ptr[m.point_offset[i]] < target
               && std::fabs((double)m.row_ptr[m.point_offset[i + 1]] - target) <= std::fabs((double)m.row_ptr[m.point_offset[i]] - target))
            i++;
        shardBegin.push_back(i);
    }
    shardBegin.push_back(nI);
}

void ImageGroup::createShardedContexts()
{
    std::string err;
    if (!g_comm.load(err)) { cout << "Error : cannot load libfrog_comm.so (" << err << ")" << endl; exit(1); }
    if (numberOfFixedImages) { cout << "Error : -fi cannot be combined with -ng / -ngl" << endl; exit(1); }
    planShards();
    frog_options o;
    frog_options_default(&o);
    o.linear_alpha = linearAlpha; o.use_scale = useScale; o.initial_grid_size = initialGridSize;
    o.bounding_box_margin = boundingBoxMargin; o.inlier_threshold = inlierThreshold;
    o.guarantee_diffeomorphism = guaranteeDiffeomorphism; o.max_displacement_ratio = maxDisplacementRatio;
    o.max_levels_hint = deformableLevels; o.stats_max_size = statsMaxSize; o.stats_max_iterations = statsMaxIterations; o.stats_epsilon = statsEpsilon;
    o.reference_order = exact;
    o.selections_in_background = 1;             // a whole run: the ahead-of-time draws beside the first iterations
    frog_model m;
    frog_pairs_model(pairs, &m);
    std::vector<int> devices(nGpus);
    for (int r = 0; r < nGpus; r++) devices[r] = loopback ? device : device + r;
    comms.assign(nGpus, nullptr);
    check(loopback ? g_comm.create_loopback(nGpus, comms.data()) : g_comm.create_rccl(nGpus, devices.data(), comms.data()),
          "frog_comm_create");
    ctxs.assign(nGpus, nullptr);
    cout << "Images sharded over " << nGpus << (loopback ? " contexts on device " : " GPUs, first device ") << device << " :";
    for (int r = 0; r < nGpus; r++) cout << " [" << shardBegin[r] << "," << shardBegin[r + 1] << ")";
    cout << endl;
    for (int r = 0; r < nGpus; r++)        // one after the other: the layout build of each context already uses all host threads
        check(frog_create(&m, &o, devices[r], shardBegin[r], shardBegin[r + 1], &ctxs[r]), "frog_create");
    ctx = ctxs[0];
    if (!hardLinks.empty()) {
        std::vector<uint64_t> a, b;
        for (const auto &hl : hardLinks) {
            a.push_back((uint64_t)m.point_offset[hl.first.image] + hl.first.point);
            b.push_back((uint64_t)m.point_offset[hl.second.image] + hl.second.point);
        }
        const float constraintWeight = m.n_images * landmarksConstraintsWeight;
        for (frog_ctx *c : ctxs)        // every context keeps the links of its own points
            check(frog_set_hard_links(c, a.data(), b.data(), a.size(), constraintWeight * constraintWeight), "frog_set_hard_links");
    }
    counts.assign(m.n_images, frog_counts{});
}

// run()'s loops for images sharded over several GPUs (imageGroup.cxx:31-128): one host thread per rank, all executing
// the same control flow on their own context; the places where the reference's loops read another image's state are
// collectives (include/frog_comm.h).  E and the oversize count are all-reduced, so every rank takes the same branch.
// Rank 0 prints and records the measures.
void ImageGroup::runSharded()
{
    using clk = std::chrono::steady_clock;
    const int N = nGpus;
    std::vector<uint64_t> replicaHashes(N, 0);
    #pragma omp parallel num_threads(N)
    {
        const int r = omp_get_thread_num();
        frog_ctx *c = ctxs[r];
        frog_comm *cm = comms[r];
        const bool root = r == 0;
        auto ck = [&](int rc, const char *what) { if (rc) { 
            #pragma omp critical
            { cout << "Error : " << what << " failed on rank " << r << " (" << rc << "): " << frog_last_error() << endl; exit(1); } } };
        // two collectives per deformable iteration, one per linear iteration (include/frog_hip.h frog_comm_mode; rank_schedule.cpp
        // has the same flow); FROG_THREE_COLLECTIVES=1: the flow of rounds 2-4
        const bool two = !getenv("FROG_THREE_COLLECTIVES");
        const bool speculate = two && !getenv("FROG_NO_SPECULATION");      // the next step's phase A queued before this step's decision (frog_step_speculate)
        bool gathered = false, phaseAQueued = false;
        ck(frog_comm_mode(c, two ? 1 : 0), "frog_comm_mode");
        auto transformPoints = [&](int apply) {
            if (two) {
                const bool done = gathered && !apply;
                gathered = false;
                if (!done) ck(g_comm.gather_points(cm, apply, 0, 0u), "frog_comm_gather_points");
                return;
            }
            ck(frog_transform_points_local(c, apply), "frog_transform_points_local");
            ck(g_comm.all_gather_xyz2(cm), "frog_comm_all_gather_xyz2");
        };
        auto updateStats = [&]() {
            ck(frog_update_stats_local(c), "frog_update_stats_local");
            ck(g_comm.all_reduce(cm, FROG_BUF_EM), "frog_comm_all_reduce");
            ck(frog_stats_publish(c), "frog_stats_publish");
        };
        auto setup = [&](int level) {
            double mn[3], mx[3];
            frog_grid_info info;
            ck(frog_bounds_local(c, mn, mx), "frog_bounds_local");
            ck(g_comm.all_reduce_bounds(cm, mn, mx), "frog_comm_all_reduce_bounds");
            ck(frog_deformable_setup_bounds(c, level, mn, mx, &info), "frog_deformable_setup_bounds");
            if (root) {
                double length[3];
                for (int k = 0; k < 3; k++) length[k] = info.bbox[2 * k + 1] - info.bbox[2 * k];
                cout << "Bounding box : "; print(info.bbox, 6);
                cout << "Box length : "; print(length, 3);
                cout << "Grid origin : "; print(info.origin, 3);
                cout << "Grid spacing : "; print(info.spacing, 3);
                cout << "Grid dimensions (control points): "; print(info.dims, 3);
            }
        };
        auto census = [&]() {
            ck(frog_count_inliers(c, counts.data()), "frog_count_inliers");       // every rank fills its own images' entries
            g_comm.barrier(cm);
            if (root) {
                long long nPairs = 0, nInliers = 0, nOutliers = 0;
                for (const auto &x : counts) { nPairs += x.pairs; nInliers += x.inliers; nOutliers += x.outliers; }
                cout << "Stats:" << endl << nPairs << " half pairs" << endl << nInliers << " inliers" << endl << nOutliers << " outliers" << endl;
                cout << "Outlier ratio (%): " << (float)100 * nOutliers / nPairs << endl;
            }
            g_comm.barrier(cm);
        };

        std::vector<uint32_t> sb(shardBegin);
        ck(g_comm.bind(cm, c, sb.data()), "frog_comm_bind");
        ck(frog_linear_init(c, linearInitializationAnchor), "frog_linear_init");       // :37
        transformPoints(0);                                                           // :38

        if (root) cout << endl << "Linear registration" << endl;
        g_comm.barrier(cm);
        auto t0 = clk::now();
        for (int iteration = 0; iteration < linearIterations; iteration++) {
            if (root && !quiet) cout << "Linear registration, iteration " << iteration + 1 << "/" << linearIterations << endl;
            if (!(iteration % statIntervalUpdate)) updateStats();
            ck(frog_linear_step_local(c), "frog_linear_step_local");
            double E = 0;
            if (two) {
                ck(g_comm.gather_points(cm, 0, 0, 0xBu), "frog_comm_gather_points");        // the two sums ride on the gather
                ck(frog_step_finish(c, &E), "frog_step_finish");
                gathered = true;
            } else {
                ck(g_comm.all_reduce(cm, FROG_BUF_ENERGY), "frog_comm_all_reduce");
                ck(frog_energy_read(c, &E, nullptr), "frog_energy_read");
            }
            transformPoints(0);
            if (root) computeLandmarkDistances((float)E);
        }
        g_comm.barrier(cm);
        if (root) { loopSeconds += std::chrono::duration<double>(clk::now() - t0).count(); loopIterations += linearIterations; }
        transformPoints(1);                                                           // :70
        ck(frog_synchronize(c), "frog_synchronize");
        g_comm.barrier(cm);
        if (root) saveDistanceHistograms("histograms_linear.csv");                    // :71 (reads every rank's samples)
        g_comm.barrier(cm);

        if (deformableLevels) {
            if (root) cout << endl << "Deformable registration" << endl;
            census();                                                                 // :76
            for (int level = 0; level < deformableLevels; level++) {
                if (root) cout << endl << "Level " << level + 1 << "/" << deformableLevels << endl;
                setup(level);                                                         // :81
                transformPoints(0);
                int numberOfGrids = 1;
                float alpha = deformableAlpha;
                if (root) cout << "alpha = " << alpha << endl;
                int numberOfDiffeomorphicIterations = 0;
                g_comm.barrier(cm);
                t0 = clk::now();
                for (int iteration = 0; iteration < deformableIterations; iteration++) {
                    if (root && !quiet)
                        cout << "Level " << level + 1 << "/" << deformableLevels << ", Iteration " << iteration + 1 << "/"
                             << deformableIterations << endl;
                    if (!(iteration % statIntervalUpdate)) updateStats();
                    if (!phaseAQueued) ck(frog_deformable_phase_a(c, alpha), "frog_deformable_phase_a");
                    phaseAQueued = false;
                    ck(g_comm.all_reduce(cm, FROG_BUF_GRIDSUM), "frog_comm_all_reduce");        // the shared common-space grid, :400-432
                    ck(frog_deformable_phase_b(c), "frog_deformable_phase_b");
                    double E = 0;
                    if (two) {
                        // the oversize count rides on the coordinate gather, the transform queued speculatively (frog_hip.h)
                        ck(g_comm.gather_points(cm, 0, 1, 0x4u), "frog_comm_gather_points");
                        if (speculate && iteration + 1 < deformableIterations && (iteration + 1) % statIntervalUpdate != 0) {
                            ck(frog_step_speculate(c), "frog_step_speculate");
                            ck(frog_deformable_phase_a(c, alpha), "frog_deformable_phase_a");
                            phaseAQueued = true;
                        }
                        ck(frog_step_finish(c, &E), "frog_step_finish");
                        gathered = (float)E >= 0;
                        if (!gathered) phaseAQueued = false;        // rejected: rolled back
                    } else {
                        ck(g_comm.all_reduce(cm, FROG_BUF_ENERGY), "frog_comm_all_reduce");         // energy sums + oversize count
                        ck(frog_deformable_phase_c(c, &E), "frog_deformable_phase_c");
                    }
                    const float e = (float)E;
                    if (e < 0) {                                                  // :97-115, the same on every rank
                        if (root) cout << endl << "Diffeomorphism is not guaranteed : Iteration canceled" << endl;
                        if (numberOfDiffeomorphicIterations == 0) {
                            alpha /= 2;
                            if (root) cout << "Halving alpha. New Value : " << alpha << endl;
                        }
                        if (root) cout << " creating new grid" << endl;
                        numberOfGrids++;
                        iteration--;
                        transformPoints(1);
                        setup(level);
                        transformPoints(0);
                        numberOfDiffeomorphicIterations = 0;
                        continue;
                    }
                    numberOfDiffeomorphicIterations++;
                    transformPoints(0);
                    if (root) computeLandmarkDistances(e);
                }
                g_comm.barrier(cm);
                if (root) { loopSeconds += std::chrono::duration<double>(clk::now() - t0).count(); loopIterations += deformableIterations; }
                census();                                                             // :123
                if (root) {
                    cout << "Number of grids for this level : " << numberOfGrids << endl;
                    gridsPerLevel.push_back(numberOfGrids);
                }
                transformPoints(1);
            }
        }
        ck(frog_synchronize(c), "frog_synchronize");
        g_comm.barrier(cm);
        if (std::getenv("FROG_CHECK_REPLICAS")) {
            // every rank holds a replica of all transformed coordinates (the all-gather's product) and of the mixture table:
            // they must be the same bits everywhere, whatever carried the collectives
            const uint64_t nP = frog_num_points(c);
            const uint32_t nI = frog_num_images(c);
            std::vector<float> xyz(3 * nP), xyz2(3 * nP), em(3 * (size_t)nI);
            ck(frog_get_points(c, xyz.data(), xyz2.data()), "frog_get_points");
            for (uint32_t i = 0; i < nI; i++) ck(frog_get_em(c, i, &em[3 * (size_t)i]), "frog_get_em");
            uint64_t h = 1469598103934665603ull;            // FNV-1a over the bit patterns
            auto eat = [&](const std::vector<float> &v) {
                for (float f : v) { uint32_t b; std::memcpy(&b, &f, 4); h = (h ^ b) * 1099511628211ull; }
            };
            eat(xyz2); eat(em);
            #pragma omp critical
            replicaHashes[r] = h;
            g_comm.barrier(cm);
            if (root) {
                bool same = true;
                for (int k = 1; k < N; k++) same = same && replicaHashes[k] == replicaHashes[0];
                cout << "Replicas identical : " << (same ? "yes" : "NO") << " (" << N << " contexts, xyz2 of " << nP << " points + " << nI << " mixtures)" << endl;
                if (!same) exit(1);
            }
            g_comm.barrier(cm);
        }
    }
}

void ImageGroup::check(int rc, const char *what)
{
    if (rc == FROG_OK) return;
    cout << "Error : " << what << " failed (" << rc << "): " << frog_last_error() << endl;
    exit(1);
}

// readPairs, imageGroup.cxx:1353-1417
void ImageGroup::readPairs(const char *fileName)
{
    int status = 0;
    const auto t_read = std::chrono::steady_clock::now();
    frog_pairs *p = frog_pairs_read(fileName, &status);
    if (std::getenv("FROG_TIMING"))
        cout << "[timing] readPairs : " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_read).count() << "s" << endl;
    if (!p) {
        // a block of size 0 is the reference's "Error : number of pairs is 0", exit(1)
        cout << "Error : number of pairs is 0 or unreadable file " << fileName << endl;
        exit(1);
    }
    usePairs(p);
    ownPairs = true;
    const uint64_t n = frog_pairs_num_pairs(pairs);
    cout << n << " pairs read : " << n * 2 << " half pairs " << endl;
}

void ImageGroup::usePairs(frog_pairs *p)
{
    if (pairs && ownPairs) frog_pairs_free(pairs);
    pairs = p;
    ownPairs = false;
}

// ImageGroup ctor state + setupStats (:1151): the reservoirs are sized in frog_create.
// Called from run() so that -ss/-emi/-se given after the file name take effect, as in
// the reference where setupStats runs inside run().
void ImageGroup::createContext()
{
    frog_options o;
    frog_options_default(&o);
    o.linear_alpha = linearAlpha;
    o.use_scale = useScale;
    o.initial_grid_size = initialGridSize;
    o.bounding_box_margin = boundingBoxMargin;
    o.inlier_threshold = inlierThreshold;
    o.guarantee_diffeomorphism = guaranteeDiffeomorphism;
    o.max_displacement_ratio = maxDisplacementRatio;
    o.max_levels_hint = deformableLevels; o.stats_max_size = statsMaxSize;
    o.reference_order = exact;
    o.selections_in_background = 1;             // a whole run: the ahead-of-time draws beside the first iterations
    o.stats_max_iterations = statsMaxIterations;
    o.stats_epsilon = statsEpsilon;
    o.n_fixed_images = numberOfFixedImages;
    frog_model m;
    frog_pairs_model(pairs, &m);
    check(frog_create(&m, &o, device, 0, m.n_images, &ctx), "frog_create");
    if (!hardLinks.empty()) {                                           // Point::hardLinks, weight :237
        std::vector<uint64_t> a, b;
        for (const auto &hl : hardLinks) {
            a.push_back((uint64_t)m.point_offset[hl.first.image] + hl.first.point);
            b.push_back((uint64_t)m.point_offset[hl.second.image] + hl.second.point);
        }
        const float constraintWeight = m.n_images * landmarksConstraintsWeight;
        check(frog_set_hard_links(ctx, a.data(), b.data(), a.size(), constraintWeight * constraintWeight), "frog_set_hard_links");
    }
    counts.assign(m.n_images, frog_counts{});
}

// readAndApplyFixedImagesTransforms, imageGroup.cxx:1419-1456: the keypoints of the fixed images move to
// their registered position (`xyz := T(xyz)`) before anything else; without -fd the transform is the identity.
// T is evaluated on the device (frog_chain.h), one link at a time with the point rounded to float in
// between, as vtkGeneralTransform's float TransformPoint does.
void ImageGroup::readAndApplyFixedImagesTransforms()
{
    if (!fixedTransformsDirectory) return;
    cout << "Reading transforms in directory " << fixedTransformsDirectory << endl;
    frog_model m;
    frog_pairs_model(pairs, &m);
    for (int i = 0; i < numberOfFixedImages; i++) {
        std::ostringstream file;
        file << fixedTransformsDirectory << "/" << i << ".json";
        int status = 0;
        frog_transform_file *tf = frog_transform_read(file.str().c_str(), &status);
        if (!tf) { cout << "Error : could not read " << file.str() << endl; exit(1); }
        const uint32_t b = m.point_offset[i], n = m.point_offset[i + 1] - b;
        std::vector<float> xyz(m.xyz + 3 * (size_t)b, m.xyz + 3 * (size_t)(b + n));
        std::vector<double> in(3 * (size_t)n), out(3 * (size_t)n);
        const frog_chain_link *links = frog_transform_links(tf);
        for (uint32_t l = 0; l < frog_transform_num_links(tf) && n; l++) {
            frog_chain *chain = nullptr;
            check(frog_chain_create(links + l, 1, device, &chain), "frog_chain_create");
            for (size_t k = 0; k < in.size(); k++) in[k] = xyz[k];
            check(frog_chain_apply(chain, in.data(), out.data(), n), "frog_chain_apply");
            for (size_t k = 0; k < in.size(); k++) xyz[k] = (float)out[k];
            frog_chain_destroy(chain);
        }
        frog_transform_free(tf);
        check(frog_pairs_set_points(pairs, i, xyz.data()), "frog_pairs_set_points");
    }
}

// run, imageGroup.cxx:31-157
void ImageGroup::run()
{
    using clk = std::chrono::steady_clock;
    if (!pairs) { cout << "Error : no pairs" << endl; exit(1); }
    if (numberOfFixedImages < 0 || numberOfFixedImages >= (int)frog_pairs_num_images(pairs)) {
        cout << "Error : -fi must leave at least one image to register" << endl;
        exit(1);
    }
    // FROG_SHARDED_ALWAYS: `-ng 1` also takes the sharded host (one rank, a real RCCL communicator of one device): the only
    // way to execute that code path -- ncclCommInitAll, grouped broadcasts, all-reduces on the context's stream -- on a
    // machine with a single GPU (tests)
    if (nGpus > 1 || (nGpus == 1 && std::getenv("FROG_SHARDED_ALWAYS"))) {
        // images sharded over several GPUs: the loops run in runSharded(), everything after them (error maps, files)
        // below is shared with the single-GPU path and asks the context that owns each image
        { const auto t_ctx = clk::now(); createShardedContexts();
          if (std::getenv("FROG_TIMING")) cout << "[timing] frog_create x " << nGpus << " : " << std::chrono::duration<double>(clk::now() - t_ctx).count() << "s" << endl; }
        runSharded();
        finishRun();
        return;
    }
    if (numberOfFixedImages) readAndApplyFixedImagesTransforms();       // :34
    { const auto t_ctx = clk::now(); createContext();                   // :36 setupStats
      if (std::getenv("FROG_TIMING")) {
          cout << "[timing] frog_create : " << std::chrono::duration<double>(clk::now() - t_ctx).count() << "s" << endl;
          double part[3] = { 0, 0, 0 };
          if (frog_create_seconds(ctx, part, nullptr) == FROG_OK)
              cout << "[timing] frog_create, layout build : " << part[0] << "s" << endl << "[timing] frog_create, allocations + uploads : " << part[1] << "s" << endl
                   << "[timing] frog_create, selections queued : " << part[2] << "s" << endl;
      } }
    check(frog_linear_init(ctx, linearInitializationAnchor), "frog_linear_init");   // :37
    check(frog_transform_points(ctx, 0), "frog_transform_points");      // :38

    auto t0 = clk::now();
    if (useRANSAC && numberOfFixedImages) {                             // :40-49
        frog_model m;
        frog_pairs_model(pairs, &m);
        frog_ransac_options ro;
        ro.iterations = numberOfRANSACIterations;
        ro.batches = RANSACBatches > 0 ? RANSACBatches : omp_get_num_procs();   // :635
        ro.inlier_distance = RANSACInlierDistance;
        ro.max_scale = RANSACMaxScale;
        for (uint32_t i = numberOfFixedImages; i < m.n_images; i++) {
            auto start = std::chrono::system_clock::now();
            cout << "RANSAC registration for image " << i << ": " << std::flush;
            int64_t inliers = 0;
            check(frog_ransac(ctx, &m, i, &ro, &inliers), "frog_ransac");
            auto end = std::chrono::system_clock::now();
            cout << inliers << " inliers, computed in " << std::chrono::duration<float>(end - start).count() << "s" << endl;
            ransacInliers.push_back({ (int)i, (long long)inliers });
        }
        check(frog_transform_points(ctx, 0), "frog_transform_points");
        check(frog_update_stats(ctx), "frog_update_stats");
        if (printStats) displayStats();
        if (printLinear) displayLinearTransforms();
    } else {
        cout << endl << "Linear registration" << endl;
        t0 = clk::now();
        for (int iteration = 0; iteration < linearIterations; iteration++) {
            if (!quiet) cout << "Linear registration, iteration " << iteration + 1 << "/" << linearIterations << endl;
            if (!(iteration % statIntervalUpdate)) check(frog_update_stats(ctx), "frog_update_stats");
            if (printStats) displayStats();
            double E = 0;
            check(frog_linear_step(ctx, &E), "frog_linear_step");
            float e = (float)E;
            if (printLinear) displayLinearTransforms();
            check(frog_transform_points(ctx, 0), "frog_transform_points");
            computeLandmarkDistances(e);
        }
        loopSeconds += std::chrono::duration<double>(clk::now() - t0).count();
        loopIterations += linearIterations;

    }

    check(frog_transform_points(ctx, 1), "frog_transform_points");      // :70
    saveDistanceHistograms("histograms_linear.csv");                    // :71

    if (deformableLevels) {
        cout << endl << "Deformable registration" << endl;
        countInliers();                                                 // :76
        for (int level = 0; level < deformableLevels; level++) {
            cout << endl << "Level " << level + 1 << "/" << deformableLevels << endl;
            auto setup = [&]() {
                frog_grid_info info;
                check(frog_deformable_setup(ctx, level, &info), "frog_deformable_setup");
                double length[3];
                for (int k = 0; k < 3; k++) length[k] = info.bbox[2 * k + 1] - info.bbox[2 * k];
                cout << "Bounding box : "; print(info.bbox, 6);
                cout << "Box length : "; print(length, 3);
                cout << "Grid origin : "; print(info.origin, 3);
                cout << "Grid spacing : "; print(info.spacing, 3);
                cout << "Grid dimensions (control points): "; print(info.dims, 3);
            };
            setup();                                                    // :81
            check(frog_transform_points(ctx, 0), "frog_transform_points");
            int numberOfGrids = 1;
            float alpha = deformableAlpha;
            cout << "alpha = " << alpha << endl;
            int numberOfDiffeomorphicIterations = 0;
            t0 = clk::now();
            for (int iteration = 0; iteration < deformableIterations; iteration++) {
                if (!quiet)
                    cout << "Level " << level + 1 << "/" << deformableLevels << ", Iteration " << iteration + 1 << "/"
                         << deformableIterations << endl;
                if (!(iteration % statIntervalUpdate)) check(frog_update_stats(ctx), "frog_update_stats");
                if (printStats) displayStats();
                double E = 0;
                check(frog_deformable_step(ctx, alpha, &E), "frog_deformable_step");
                float e = (float)E;
                if (e < 0) {                                            // :97-115
                    cout << endl << "Diffeomorphism is not guaranteed : Iteration canceled" << endl;
                    if (numberOfDiffeomorphicIterations == 0) {
                        alpha /= 2;
                        cout << "Halving alpha. New Value : " << alpha << endl;
                    }
                    cout << " creating new grid" << endl;
                    numberOfGrids++;
                    iteration--;
                    check(frog_transform_points(ctx, 1), "frog_transform_points");
                    setup();
                    check(frog_transform_points(ctx, 0), "frog_transform_points");
                    numberOfDiffeomorphicIterations = 0;
                    continue;
                }
                numberOfDiffeomorphicIterations++;
                check(frog_transform_points(ctx, 0), "frog_transform_points");
                computeLandmarkDistances(e);
            }
            loopSeconds += std::chrono::duration<double>(clk::now() - t0).count();
            loopIterations += deformableIterations;
            countInliers();                                             // :123
            cout << "Number of grids for this level : " << numberOfGrids << endl;
            gridsPerLevel.push_back(numberOfGrids);
            check(frog_transform_points(ctx, 1), "frog_transform_points");
        }
    }
    finishRun();
}

// run(), imageGroup.cxx:130-155: everything after the iteration loops
void ImageGroup::finishRun()
{
    using clk = std::chrono::steady_clock;
    if (deformableLevels) {
        int total = 0;
        cout << "Grids per level : ";
        for (int n : gridsPerLevel) { total += n; cout << n << " "; }
        cout << endl << "Total number of grids : " << total << endl;
        { auto t = clk::now(); saveErrorMaps();                         // :141
          if (std::getenv("FROG_TIMING")) cout << "[timing] error maps : " << std::chrono::duration<double>(clk::now() - t).count() << "s" << endl; }
    }

    displayStats();                                                     // :144
    const bool timing = std::getenv("FROG_TIMING") != nullptr;         // new: where the time after the solve goes
    auto lap = [&, last = clk::now()](const char *what) mutable {
        if (timing) cout << "[timing] " << what << " : " << std::chrono::duration<double>(clk::now() - last).count() << "s" << endl;
        last = clk::now();
    };
    saveDistanceHistograms("histograms.csv");
    saveMeasures(outputFileName);
    lap("histograms + measures");
    saveTransforms();
    lap("transforms");
    saveLandmarkDistances();                                            // :149
    saveTransformedLandmarks();                                         // :150
    if (writePairs) writeLinksDistances();                              // :151
    saveStatsJSON();
    lap("landmarks + pairs + bbox.json");
}

// saveErrorMaps, imageGroup.cxx:475-567: the residual sums come from the device (one sweep),
// the nearest-node binning keeps the reference's point order (frog_get_error_map).
void ImageGroup::saveErrorMaps()
{
    std::filesystem::create_directory(errorMapsSubdirectory.c_str());
    if (ctxs.empty()) check(frog_residual_sums(ctx), "frog_residual_sums");
    else for (frog_ctx *c : ctxs) check(frog_residual_sums(c), "frog_residual_sums");      // every rank: its own images
    const uint32_t n = frog_num_images(ctx);
    std::vector<frog_grid_info> infos(n);
    std::vector<std::vector<float>> maps(n);
    for (uint32_t image = numberOfFixedImages; image < n; image++)                 // :481 (the lattice's geometry; serial: the first call
        check(frog_get_error_map(ctxOf(image), image, &infos[image], nullptr, 0), "frog_get_error_map");     // joins the context's streams)
    int failed = 0;
    // binning (frog_get_error_map works on the host copies frog_residual_sums left: read-only from here), compression and file
    // output on all host threads -- the binning of 100 images one after the other was half of this function's 0.17 s
    #pragma omp parallel for schedule(dynamic, 1) num_threads(frog::host_threads())
    for (int image = numberOfFixedImages; image < (int)n; image++) {
        const frog_grid_info &info = infos[image];
        maps[image].resize((size_t)4 * info.dims[0] * info.dims[1] * info.dims[2]);
        frog_grid_info again;
        if (frog_get_error_map(ctxOf(image), (uint32_t)image, &again, maps[image].data(), maps[image].size())) {
            #pragma omp atomic
            failed++;
            continue;
        }
        std::ostringstream file;
        file << errorMapsSubdirectory << "/" << image << ".nii.gz";
        const uint32_t dims[3] = { (uint32_t)info.dims[0], (uint32_t)info.dims[1], (uint32_t)info.dims[2] };
        if (frog_nifti_write(file.str().c_str(), dims, info.spacing, info.origin, 4, maps[image].data())) {
            #pragma omp atomic
            failed++;
        }
    }
    if (failed) check(FROG_E_IO, "frog_nifti_write");
}

// Stats::getInlierProbability + chipdf, stats.h:10-16,84-92 (promotions as upstream)
static float inlierProbability(float d, float c1, float c2, float ratio)
{
    const float eps = 1e-10;
    if (d < 0.1) return 1;
    auto chipdf = [](float x) -> float {
        float c = 0.797884560802865;
        float x2 = x * x;
        return c * x2 * exp(-0.5 * x2);
    };
    float x1 = ratio * chipdf(d / (c1 + eps)) / (c1 + eps);
    float x2 = (1.0 - ratio) * chipdf(d / (c2 + eps)) / (c2 + eps);
    return x1 / (x1 + x2 + eps);
}

// writeLinksDistances, imageGroup.cxx:924-986 ("-wp 1"): every half-link as
// image1,point1,image2,point2,distance,probability(image1's stats only), all six stored as
// f32 like upstream, sorted by distance, gzip'd CSV without a trailing newline.
void ImageGroup::writeLinksDistances()
{
    frog_model m;
    frog_pairs_model(pairs, &m);
    const uint64_t P = frog_num_points(ctx);
    std::vector<float> xyz2(3 * P);
    check(frog_get_points(ctx, nullptr, xyz2.data()), "frog_get_points");
    struct Row { float v[6]; };
    std::vector<Row> rows;
    rows.reserve(m.row_ptr[P]);
    for (uint32_t i1 = 0; i1 < m.n_images; i1++) {
        float em[3];
        check(frog_get_em(ctx, i1, em), "frog_get_em");
        for (uint32_t p = m.point_offset[i1]; p < m.point_offset[i1 + 1]; p++) {
            const float *pA = &xyz2[3 * (size_t)p];
            for (uint64_t l = m.row_ptr[p]; l < m.row_ptr[p + 1]; l++) {
                const uint32_t i2 = m.link_image[l], p2 = m.link_point[l];
                const float *pB = &xyz2[3 * ((size_t)m.point_offset[i2] + p2)];
                float d2 = 0;
                for (int k = 0; k < 3; k++) { const float t = pA[k] - pB[k]; d2 += t * t; }     // vtkMath::Distance2BetweenPoints
                const float dist = std::sqrt(d2);
                rows.push_back(Row{ { (float)i1, (float)(p - m.point_offset[i1]), (float)i2, (float)p2, dist,
                                      inlierProbability(dist, em[0], em[1], em[2]) } });
            }
        }
    }
    std::sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) { return a.v[4] < b.v[4]; });
    gzFile f = gzopen("pairs.csv.gz", "wb");
    if (!f) { cout << "Error : cannot write pairs.csv.gz" << endl; exit(1); }
    std::ostringstream out;
    for (size_t i = 0; i < rows.size(); i++) {
        for (int j = 0; j < 6; j++) { out << rows[i].v[j]; if (j < 5) out << ","; }
        if (i + 1 < rows.size()) out << "\n";
        if (out.tellp() > (1 << 20) || i + 1 == rows.size()) {
            const std::string s = out.str();
            if (!s.empty() && gzwrite(f, s.data(), (unsigned)s.size()) != (int)s.size()) { cout << "Error : cannot write pairs.csv.gz" << endl; exit(1); }
            out.str(std::string());
        }
    }
    gzclose(f);
}

// addLandmarks, imageGroup.cxx:1161-1227, validation landmarks (-l): one file per image in the
// directory (sorted names), lines "name,x,y,z" ('#' comments); every landmark becomes an extra,
// link-less point of its image -- so it is moved by transformPoints and enters the bounding boxes,
// exactly as upstream.  With -lc the landmarks of one name are also hard-linked to each other.
void ImageGroup::addLandmarks(const char *path, bool asConstraints)
{
    if (!pairs) { cout << "Error : no pairs" << endl; exit(1); }
    std::map<std::string, std::vector<Landmark>> constraints;
    std::vector<std::string> files;
    for (const auto &p : std::filesystem::directory_iterator(path)) files.push_back(p.path().string());
    std::sort(files.begin(), files.end());
    const uint32_t nImages = frog_pairs_num_images(pairs);
    for (size_t i = 0; i < files.size(); i++) {
        std::ifstream infile(files[i]);
        std::string line;
        if (i + 1 > nImages) continue;                                  // `i > images.size() - 1`
        frog_model m;
        frog_pairs_model(pairs, &m);
        uint32_t count = m.point_offset[i + 1] - m.point_offset[i];     // landmark.point = points.size()
        std::vector<float> xyz;
        while (std::getline(infile, line)) {
            if (line.empty() || line[0] == '#') continue;
            size_t pos = line.find(',');
            const std::string name = line.substr(0, pos);
            line.erase(0, pos + 1);
            float pt[3];
            for (int j = 0; j < 3; j++) {
                pos = line.find(',');
                const std::string coord = line.substr(0, pos);
                line.erase(0, pos == std::string::npos ? line.size() : pos + 1);
                pt[j] = std::stof(coord);
                if (j < 2 && invertLandmarksCoordinates) pt[j] *= -1;   // get opposite x and y coordinates!
            }
            xyz.insert(xyz.end(), pt, pt + 3);
            landmarks[name].push_back(Landmark{ (uint32_t)i, count });
            constraints[name].push_back(Landmark{ (uint32_t)i, count });
            count++;
        }
        if (!xyz.empty()) check(frog_pairs_append_points(pairs, (uint32_t)i, xyz.data(), (uint32_t)(xyz.size() / 3)), "frog_pairs_append_points");
    }
    if (!asConstraints) return;
    // :1210-1225: every landmark is hard-linked to every other landmark of the same name
    for (const auto &kv : constraints)
        for (const Landmark &landmark : kv.second)
            for (const Landmark &landmark2 : kv.second) {
                if (landmark.image == landmark2.image && landmark.point == landmark2.point) continue;
                hardLinks.push_back({ landmark, landmark2 });
            }
}

// xyz2 of every landmark, in the order of the map (name, then entry)
void ImageGroup::fetchLandmarks()
{
    frog_model m;
    frog_pairs_model(pairs, &m);
    std::vector<uint64_t> idx;
    for (const auto &kv : landmarks)
        for (const Landmark &l : kv.second) idx.push_back((uint64_t)m.point_offset[l.image] + l.point);
    landmarkXyz2.resize(3 * idx.size());
    check(frog_get_points2_subset(ctx, idx.data(), idx.size(), landmarkXyz2.data()), "frog_get_points2_subset");
}

// computeLandmarkDistances, imageGroup.cxx:1229-1282
void ImageGroup::computeLandmarkDistances(float e)
{
    if (!quiet || !landmarks.empty()) cout << "E = " << e;
    if (std::isnan(e)) {
        cout << endl << "Error : NaN" << endl;
        exit(1);
    }
    if (landmarks.empty()) {
        if (!quiet) cout << endl;
        measures.push_back(Measure{ e, 0, 0, 0 });
        return;
    }
    fetchLandmarks();
    std::vector<float> distances;
    size_t at = 0;
    for (const auto &kv : landmarks) {
        const size_t n = kv.second.size();
        float center[3] = { 0, 0, 0 };
        for (size_t l = 0; l < n; l++)
            for (int k = 0; k < 3; k++) center[k] += landmarkXyz2[3 * (at + l) + k] / n;
        for (size_t l = 0; l < n; l++) {
            float d2 = 0;
            for (int k = 0; k < 3; k++) { const float t = landmarkXyz2[3 * (at + l) + k] - center[k]; d2 += t * t; }
            distances.push_back(std::sqrt(d2));
        }
        at += n;
    }
    double sum = std::accumulate(distances.begin(), distances.end(), 0.0);
    double mean = sum / distances.size();
    auto max = std::max_element(distances.begin(), distances.end());
    double sq_sum = std::inner_product(distances.begin(), distances.end(), distances.begin(), 0.0);
    double stdev = std::sqrt(sq_sum / distances.size() - mean * mean);
    cout << ", " << distances.size() << " landmarks:max=" << *max << ",average=" << mean << ",stdev=" << stdev << endl;
    measures.push_back(Measure{ e, (float)mean, *max, (float)stdev });
}

// saveTransformedLandmarks, imageGroup.cxx:1284-1316
bool ImageGroup::saveTransformedLandmarks()
{
    if (landmarks.empty()) return false;
    fetchLandmarks();
    frogjson::Value root = frogjson::Value::object();
    size_t at = 0;
    for (const auto &kv : landmarks) {
        frogjson::Value arr = frogjson::Value::array();
        for (const Landmark &l : kv.second) {
            frogjson::Value land = frogjson::Value::object();
            land["image"] = frogjson::Value((double)l.image);
            frogjson::Value coords = frogjson::Value::array();
            for (int i = 0; i < 3; i++) coords.push(frogjson::Value((double)landmarkXyz2[3 * at + i]));
            land["xyz"] = coords;
            arr.push(land);
            at++;
        }
        root[kv.first] = arr;
    }
    std::fstream fs;
    fs.open("transformedLandmarks.json", std::fstream::out | std::fstream::trunc);
    fs << root.serialize();
    fs.close();
    return true;
}

// saveLandmarkDistances, imageGroup.cxx:1318-1351
void ImageGroup::saveLandmarkDistances()
{
    if (landmarks.empty()) return;
    fetchLandmarks();
    std::fstream fs;
    fs.open("distances.txt", std::fstream::out | std::fstream::trunc);
    size_t at = 0;
    for (const auto &kv : landmarks) {
        const size_t n = kv.second.size();
        float center[3] = { 0, 0, 0 };
        for (size_t l = 0; l < n; l++)
            for (int k = 0; k < 3; k++) center[k] += landmarkXyz2[3 * (at + l) + k] / n;
        for (size_t l = 0; l < n; l++) {
            float d2 = 0;
            for (int k = 0; k < 3; k++) { const float t = landmarkXyz2[3 * (at + l) + k] - center[k]; d2 += t * t; }
            fs << std::sqrt(d2) << "," << kv.first << "," << kv.second[l].image << endl;
        }
        at += n;
    }
    fs.close();
}

// displayStats, imageGroup.cxx:899-908 + Stats::displayParameters, stats.cxx:72-93
void ImageGroup::displayStats()
{
    const uint32_t n = frog_num_images(ctx);
    std::vector<float> smp((size_t)std::max(1, statsMaxSize));
    for (uint32_t i = 0; i < n; i++) {
        float em[3];
        check(frog_get_em(ctx, i, em), "frog_get_em");
        int s = 0;
        check(frog_get_samples(ctxOf(i), i, smp.data(), nullptr, (int)smp.size(), &s), "frog_get_samples");
        s = std::min<int>(s, (int)smp.size());
        cout << "Stats " << i << ":";
        cout << "c1=" << em[0] << ",c2=" << em[1] << ",r=" << em[2] << ",nSamples=" << s;
        if (s > 0) {
            double sum = std::accumulate(smp.begin(), smp.begin() + s, 0.0);
            double mean = sum / s;
            double sq = std::inner_product(smp.begin(), smp.begin() + s, smp.begin(), 0.0);
            float mx = *std::max_element(smp.begin(), smp.begin() + s);
            cout << ",max=" << mx << ",mean=" << mean << ",stdev=" << std::sqrt(sq / s - mean * mean);
        }
        cout << endl;
    }
}

// displayLinearTransforms, imageGroup.cxx:600-627
void ImageGroup::displayLinearTransforms()
{
    const uint32_t n = frog_num_images(ctx);
    for (uint32_t i = numberOfFixedImages; i < n; i++) {                           // :602
        double m[16];
        check(frog_get_linear(ctxOf(i), i, m), "frog_get_linear");
        cout << "Image " << i << ", translation=" << m[3] << " " << m[7] << " " << m[11] << endl;
        cout << "scale=" << m[0] << " " << m[5] << " " << m[10] << endl;
    }
}

// countInliers, imageGroup.cxx:988-1060
void ImageGroup::countInliers()
{
    check(frog_count_inliers(ctx, counts.data()), "frog_count_inliers");
    long long nPairs = 0, nInliers = 0, nOutliers = 0;
    for (const auto &c : counts) { nPairs += c.pairs; nInliers += c.inliers; nOutliers += c.outliers; }
    cout << "Stats:" << endl;
    cout << nPairs << " half pairs" << endl;
    cout << nInliers << " inliers" << endl;
    cout << nOutliers << " outliers" << endl;
    cout << "Outlier ratio (%): " << (float)100 * nOutliers / nPairs << endl;
}

// saveDistanceHistograms, imageGroup.cxx:850-885
void ImageGroup::saveDistanceHistograms(const char *file)
{
    const uint32_t n = frog_num_images(ctx);
    std::vector<std::vector<float>> hist(n);
    size_t maxSize = 0;
    std::fstream fs;
    fs.open(file, std::fstream::out | std::fstream::trunc);
    for (uint32_t i = 0; i < n; i++) {
        int sz = 0;
        check(frog_get_histogram(ctxOf(i), i, nullptr, 0, &sz), "frog_get_histogram");
        hist[i].assign((size_t)sz, 0.f);
        if (sz) check(frog_get_histogram(ctxOf(i), i, hist[i].data(), sz, &sz), "frog_get_histogram");
        maxSize = std::max(maxSize, hist[i].size());
        fs << "image " << i;
        if (i < n - 1) fs << ","; else fs << endl;
    }
    for (size_t d = 0; d < maxSize; d++)
        for (uint32_t i = 0; i < n; i++) {
            if (hist[i].size() <= d) fs << 0; else fs << hist[i][d];
            if (i < n - 1) fs << ","; else fs << endl;
        }
    fs.close();
}

// saveMeasures, imageGroup.cxx:1475-1491
void ImageGroup::saveMeasures(const char *file)
{
    std::fstream fs;
    fs.open(file, std::fstream::out | std::fstream::trunc);
    fs << "Iteration, E, landmarkAv, landmarkMax, landmarkSTD" << endl;
    for (size_t i = 0; i < measures.size(); i++)
        fs << i << "," << measures[i].E << "," << measures[i].landmarkAv << "," << measures[i].landmarkMax << ','
           << measures[i].landmarkSTD << endl;
    fs.close();
}

// saveTransforms, imageGroup.cxx:1458-1473 -> writeFrogJSON, tools/transformIO.h:163-258.
// One entry per transform in creation order (PostMultiply chain): the matrix, then every
// lattice.  Default ("compact", transformIO.h:196-208): the coefficients go to a sidecar
// <image>.json.<n>.nii.gz (3-component f32 NIfTI, origin in the qform) and the JSON entry only
// names it ("file"); with "-j": dimensions/origin/spacing/coeffs inside the JSON.
void ImageGroup::saveTransforms()
{
    std::filesystem::create_directory(transformSubdirectory.c_str());
    const uint32_t n = frog_num_images(ctx);
    const int nGrids = frog_num_grids(ctx);
    // One thread brings an image's matrix and lattices back from the device (the context's stream) and hands them over as a task;
    // the other host threads format, compress and write while it fetches the next image: gzip of 700 sidecars is what this
    // function costs (1.76 s on one thread for 100 images x 7 lattices; fetching everything first, then writing: 0.056 + 0.11 s)
    struct Fetched { double m[16]; std::vector<frog_grid_info> info; std::vector<std::vector<float>> coeffs; };
    std::vector<Fetched> all(n);
    int failed = 0, fetch_rc = FROG_OK;
    const char *fetch_what = "";
    auto write_image = [&](int image) {
        const Fetched &f = all[image];
        frogjson::Value transforms = frogjson::Value::array();
        {
            frogjson::Value t = frogjson::Value::object();
            t["type"] = frogjson::Value("vtkMatrixToLinearTransform");
            frogjson::Value mat = frogjson::Value::array();
            for (int k = 0; k < 16; k++) mat.push(frogjson::Value(f.m[k]));
            t["matrix"] = mat;
            transforms.push(t);
        }
        for (int k = 0; k < nGrids; k++) {
            const frog_grid_info &info = f.info[k];
            const std::vector<float> &c = f.coeffs[k];
            frogjson::Value t = frogjson::Value::object();
            t["type"] = frogjson::Value("vtkBSplineTransform");
            if (!writeSingleFileTransforms) {
                std::ostringstream base;
                base << image << ".json." << k << ".nii.gz";
                t["file"] = frogjson::Value(base.str());
                const std::string path = transformSubdirectory + "/" + base.str();
                const uint32_t d3[3] = { (uint32_t)info.dims[0], (uint32_t)info.dims[1], (uint32_t)info.dims[2] };
                if (frog_nifti_write(path.c_str(), d3, info.spacing, info.origin, 3, c.data())) {
                    #pragma omp atomic
                    failed++;
                }
                transforms.push(t);
                continue;
            }
            frogjson::Value dims = frogjson::Value::array(), ori = frogjson::Value::array(), sp = frogjson::Value::array();
            for (int a = 0; a < 3; a++) {
                dims.push(frogjson::Value((double)info.dims[a]));
                ori.push(frogjson::Value(info.origin[a]));
                sp.push(frogjson::Value(info.spacing[a]));
            }
            t["dimensions"] = dims; t["origin"] = ori; t["spacing"] = sp;
            frogjson::Value coeffs = frogjson::Value::array();
            coeffs.arr.reserve(c.size());
            for (size_t j = 0; j < c.size(); j++) coeffs.arr.push_back(frogjson::Value((double)c[j]));
            t["coeffs"] = coeffs;
            transforms.push(t);
        }
        frogjson::Value root = frogjson::Value::object();
        root["transforms"] = transforms;
        std::ostringstream file;
        file << transformSubdirectory << "/" << image << ".json";
        std::fstream fs;
        fs.open(file.str(), std::fstream::out | std::fstream::trunc);
        fs << root.serialize();
        fs.close();
    };
    #pragma omp parallel num_threads(frog::host_threads())
    #pragma omp single
    for (uint32_t image = numberOfFixedImages; image < n; image++) {               // :1464
        Fetched &f = all[image];
        frog_ctx *owner = ctxOf(image);
        if ((fetch_rc = frog_get_linear(owner, image, f.m)) != FROG_OK) { fetch_what = "frog_get_linear"; break; }
        f.info.resize(nGrids); f.coeffs.resize(nGrids);
        for (int k = 0; k < nGrids && fetch_rc == FROG_OK; k++) {
            fetch_rc = frog_get_grid(owner, image, k, &f.info[k], nullptr, 0);
            if (fetch_rc != FROG_OK) break;
            f.coeffs[k].resize((size_t)3 * f.info[k].dims[0] * f.info[k].dims[1] * f.info[k].dims[2]);
            fetch_rc = frog_get_grid(owner, image, k, &f.info[k], f.coeffs[k].data(), f.coeffs[k].size());
        }
        if (fetch_rc != FROG_OK) { fetch_what = "frog_get_grid"; break; }
        #pragma omp task firstprivate(image)
        {
            write_image((int)image);
            std::vector<std::vector<float>>().swap(all[image].coeffs);              // the lattices of 500 images need not stay
        }
    }
    check(fetch_rc, fetch_what);
    if (failed) check(FROG_E_IO, "frog_nifti_write");
}

// bbox.json: the `stats` object of the reference (imageGroup.cxx:152-155) =
// countInliers' per-image records (:1036-1058) + saveBoundingBox (:1493-1511).
void ImageGroup::saveStatsJSON()
{
    frogjson::Value stats = frogjson::Value::object();
    const uint32_t n = frog_num_images(ctx);
    if (deformableLevels) {
        long long nPairs = 0, nInliers = 0, nOutliers = 0;
        frogjson::Value images = frogjson::Value::array();
        for (uint32_t i = 0; i < n; i++) {
            if ((int)i < numberOfFixedImages) { images.push(frogjson::Value::object()); continue; }   // :995-1000
            const frog_counts &c = counts[i];
            nPairs += c.pairs; nInliers += c.inliers; nOutliers += c.outliers;
            frogjson::Value s = frogjson::Value::object();
            s["points"] = frogjson::Value((double)c.points);
            s["pairs"] = frogjson::Value((double)c.pairs);
            s["inliers"] = frogjson::Value((double)c.inliers);
            s["outliers"] = frogjson::Value((double)c.outliers);
            frogjson::Value em = frogjson::Value::object();
            em["c1"] = frogjson::Value((double)c.c1);
            em["c2"] = frogjson::Value((double)c.c2);
            em["ratio"] = frogjson::Value((double)c.ratio);
            s["EMStats"] = em;
            images.push(s);
        }
        stats["images"] = images;
        stats["halfPairs"] = frogjson::Value((double)nPairs);
        stats["inliers"] = frogjson::Value((double)nInliers);
        stats["outliers"] = frogjson::Value((double)nOutliers);
        stats["outlierRatio"] = frogjson::Value((double)nOutliers / (double)nPairs);
    }
    if (!ransacInliers.empty()) {                                       // :707-714
        frogjson::Value arr = frogjson::Value::array();
        for (const auto &r : ransacInliers) {
            frogjson::Value e = frogjson::Value::object();
            e["image"] = frogjson::Value((double)r.first);
            e["threshold"] = frogjson::Value((double)RANSACInlierDistance);
            e["inliers"] = frogjson::Value((double)r.second);
            arr.push(e);
        }
        stats["RANSAC"] = arr;
    }
    // bounding box of every image's xyz (getBoundingBox(box, true))
    const uint64_t P = frog_num_points(ctx);
    std::vector<float> xyz(3 * P);
    check(frog_get_points(ctx, xyz.data(), nullptr), "frog_get_points");
    if (!ctxs.empty()) {
        // a context re-bases only its own images' xyz: take every image's rows from its owner
        frog_model m;
        frog_pairs_model(pairs, &m);
        std::vector<float> part(3 * P);
        for (size_t r = 1; r < ctxs.size(); r++) {
            check(frog_get_points(ctxs[r], part.data(), nullptr), "frog_get_points");
            const size_t b = 3 * (size_t)m.point_offset[shardBegin[r]], e = 3 * (size_t)m.point_offset[shardBegin[r + 1]];
            std::copy(part.begin() + b, part.begin() + e, xyz.begin() + b);
        }
    }
    double mn[3] = { 1e300, 1e300, 1e300 }, mx[3] = { -1e300, -1e300, -1e300 };
    for (uint64_t p = 0; p < P; p++)
        for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], (double)xyz[3 * p + k]); mx[k] = std::max(mx[k], (double)xyz[3 * p + k]); }
    frogjson::Value bmin = frogjson::Value::array(), bmax = frogjson::Value::array(), bbox = frogjson::Value::array();
    for (int k = 0; k < 3; k++) { bmin.push(frogjson::Value(mn[k])); bmax.push(frogjson::Value(mx[k])); }
    bbox.push(bmin); bbox.push(bmax);
    stats["bbox"] = bbox;
    std::fstream fs;
    fs.open("bbox.json", std::fstream::out | std::fstream::trunc);
    fs << stats.serialize();
    fs.close();
}

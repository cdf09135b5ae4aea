// frog_main.cpp -- the `frog` command line of the MI355X build.
//
// Same surface as the reference CLI (registration/frog.cxx:8-221):
//     frog inputPairs.bin [options]
// readPairs(argv[1]) happens BEFORE the flags are parsed (frog.cxx:69 vs :74),
// flags are key/value pairs compared with strcmp, unknown keys are skipped two at
// a time, -j is the only valueless flag and -lanchor takes three values.
// Outputs go to the current directory.
//
// Additions of this build:  -dev <n> HIP device, -q 0/1 quiet iterations, -ng <n> images sharded over n GPUs
// (one host thread per GPU, RCCL collectives; -ngl <n>: n contexts on one device, host-staged collectives),
// and a generator mode   frog --synth out.bin nImages pointsPerImage pairsPerBlock [seed]
// that writes a synthetic pairs.bin (the reference ships no data).

#include "image_group.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <thread>

#include <omp.h>

using std::cout;
using std::endl;

int main(int argc, char *argv[])
{
    auto start = std::chrono::system_clock::now();
    if (std::getenv("FROG_TIMING"))              // with the caller's own clock around the process: what loading and leaving cost
        cout << "[timing] main entered at " << std::fixed << std::chrono::duration<double>(start.time_since_epoch()).count() << std::defaultfloat << endl;
    auto groupOwner = std::make_unique<ImageGroup>();      // (released by hand at the end: FROG_TIMING times it)
    ImageGroup &group = *groupOwner;

    if (argc >= 2 && strcmp(argv[1], "--synth") == 0) {
        if (argc < 6) { cout << "Usage : frog --synth out.bin nImages pointsPerImage pairsPerBlock [seed]" << endl; return 1; }
        frog_synth_params sp;
        frog_synth_defaults(&sp);
        sp.n_images = (uint32_t)atoi(argv[3]);
        sp.points_per_image = (uint32_t)atoi(argv[4]);
        sp.pairs_per_block = atof(argv[5]);
        if (argc > 6) sp.seed = (uint64_t)atoll(argv[6]);
        frog_pairs *p = frog_synth_generate(&sp);
        if (!p) { cout << "Error : bad generator parameters" << endl; return 1; }
        int rc = frog_pairs_write(p, argv[2]);
        cout << frog_pairs_num_pairs(p) << " pairs written to " << argv[2] << endl;
        frog_pairs_free(p);
        return rc ? 1 : 0;
    }

    if (argc < 2) {
        cout << "Usage : frog inputPairs.bin [options]" << endl;
        cout << "Options : " << endl;
        cout << endl << "*Linear registration:" << endl;
        cout << "-dlinear 0/1  : display linear parameters during registration. default : " << group.printLinear << endl;
        cout << "-lanchor x y z: set initialization anchor relative position. Default : " << group.linearInitializationAnchor[0]
             << " " << group.linearInitializationAnchor[1] << " " << group.linearInitializationAnchor[2] << endl;
        cout << "-la <value>   : set alpha. Default : " << group.linearAlpha << endl;
        cout << "-li number    : number of iterations. Default : " << group.linearIterations << endl;
        cout << "-s 0/1        : use scale. Default : " << group.useScale << endl;
        cout << endl << "*Deformable registration:" << endl;
        cout << "-da value     : set alpha. Default : " << group.deformableAlpha << endl;
        cout << "-di number    : number of iterations for each level. Default : " << group.deformableIterations << endl;
        cout << "-dl number    : number of levels. Default : " << group.deformableLevels << endl;
        cout << "-g spacing    : initial grid spacing. Default : " << group.initialGridSize << endl;
        cout << "-gd 0/1       : guaranteed diffeomorphism. Default : " << group.guaranteeDiffeomorphism << endl;
        cout << "-gm ratio     : maximal displacement ratio to guarantee diffeomorphism. Default : " << group.maxDisplacementRatio << endl;
        cout << endl << "*EM Weighting:" << endl;
        cout << "-dstats 0/1   : display stats during registration. Default : " << group.printStats << endl;
        cout << "-emi number   : max number of iterations for EM weighting. Default : " << group.statsMaxIterations << endl;
        cout << "-si number    : interval update for statistics. Default : " << group.statIntervalUpdate << endl;
        cout << "-se number    : stats epsilon. Default : " << group.statsEpsilon << endl;
        cout << "-ss number    : stats maximal sample size. Default : " << group.statsMaxSize << endl;
        cout << "-t threshold  : inlier probability threshold. Default : " << group.inlierThreshold << endl;
        cout << endl << "*Registration with fixed images:" << endl;
        cout << "-fd path      : fixed images transforms directory." << endl;
        cout << "-fi number    : number of fixed images. Default : " << group.numberOfFixedImages << endl;
        cout << "-r 0/1        : use RANSAC instead of linear registration. Default : " << group.useRANSAC << endl;
        cout << "-ri number    : number of RANSAC iterations. Default : " << group.numberOfRANSACIterations << endl;
        cout << "-rs maxScale  : maximum allowed scale for RANSAC iterations. Default : " << group.RANSACMaxScale << endl;
        cout << "-rid value    : RANSAC inlier distance. Default : " << group.RANSACInlierDistance << endl;
        cout << "-rb number    : RANSAC candidate batches (seeds 0, 1000, ...). Default : number of cores, as upstream" << endl;
        cout << endl << "*Reference landmarks:" << endl;
        cout << "-l path       : path containing reference landmarks." << endl;
        cout << "-lc path      : path containing constraint landmarks." << endl;
        cout << "-il 0/1       : invert landmarks x and y coordinates. Default : " << group.invertLandmarksCoordinates << endl;
        cout << "-lcw path     : landmarks constraints weight. Default: " << group.landmarksConstraintsWeight << endl;
        cout << endl << "*Other parameters:" << endl;
        cout << "-nt number    : set number of host threads. Default : number of cores" << endl;
        cout << "-mf file      : path+name of measure.csv file." << endl;
        cout << "-wp 0/1       : write pairs, distances and probabilities. Default : " << group.writePairs << endl;
        cout << "-j            : outputs a single big JSON file for each transform. Default : " << group.writeSingleFileTransforms << endl;
        cout << "-ts subdir    : subdirectory where transforms will be written. Default : " << group.transformSubdirectory << endl;
        cout << "-dev number   : HIP device. Default : " << group.device << endl;
        cout << "-ng number    : shard the images over this many GPUs (devices dev .. dev+n-1), RCCL collectives. Default : " << group.nGpus << endl;
        cout << "-ngl number   : the same with n contexts on ONE device and host-staged collectives (rehearsal)" << endl;
        cout << "-q 0/1        : do not print one line per iteration. Default : " << group.quiet << endl;
        cout << "-exact 0/1    : every solver loop in the reference's own order and arithmetic (bit-equal to the CPU path's" << endl
             << "                coefficients; about 10 times slower: 285 instead of 2 500 iterations/s on the 100-image" << endl
             << "                benchmark group). Default : " << group.exact << endl;
        return 1;
    }

    // The HIP runtime's first call (device discovery, the code objects of libfrog_hip.so) costs 0.15-0.25 s: started here, beside
    // readPairs, instead of inside frog_create behind it.  -dev is looked up ahead of the flag loop for this alone.
    int warmDevice = 0;
    for (int i = 2; i + 1 < argc; i++) if (strcmp(argv[i], "-dev") == 0) warmDevice = atoi(argv[i + 1]);
    // (readPairs and the flag loop leave through exit(1) on bad input: the thread is joined by an exit handler then, before the
    // runtime's own static objects go -- leaving it in the middle of the runtime's start-up ended in malloc's abort now and then)
    static std::thread warm;
    warm = std::thread([warmDevice] { (void)frog_device_warm(warmDevice); });
    std::atexit([] { if (warm.joinable()) warm.join(); });

    cout << "Reading : " << argv[1] << endl;
    group.readPairs(argv[1]);

    int argumentsIndex = 2;
    while (argumentsIndex < argc) {
        char *key = argv[argumentsIndex];
        char *value = argumentsIndex + 1 < argc ? argv[argumentsIndex + 1] : (char *)"0";
        int increment = 2;

        if (strcmp(key, "-da") == 0) group.deformableAlpha = atof(value);
        if (strcmp(key, "-dlinear") == 0) group.printLinear = atoi(value);
        if (strcmp(key, "-dstats") == 0) group.printStats = atoi(value);
        if (strcmp(key, "-di") == 0) group.deformableIterations = atoi(value);
        if (strcmp(key, "-dl") == 0) group.deformableLevels = atoi(value);
        if (strcmp(key, "-emi") == 0) group.statsMaxIterations = atoi(value);
        if (strcmp(key, "-fi") == 0) group.numberOfFixedImages = atoi(value);
        if (strcmp(key, "-fd") == 0) group.fixedTransformsDirectory = value;
        if (strcmp(key, "-g") == 0) group.initialGridSize = atof(value);
        if (strcmp(key, "-gd") == 0) group.guaranteeDiffeomorphism = atoi(value);
        if (strcmp(key, "-gm") == 0) group.maxDisplacementRatio = atof(value);
        if (strcmp(key, "-il") == 0) group.invertLandmarksCoordinates = atoi(value);
        if (strcmp(key, "-lanchor") == 0 && argumentsIndex + 3 < argc) {
            increment = 4;
            for (int i = 0; i < 3; i++) group.linearInitializationAnchor[i] = atof(argv[argumentsIndex + 1 + i]);
        }
        if (strcmp(key, "-la") == 0) group.linearAlpha = atof(value);
        if (strcmp(key, "-li") == 0) group.linearIterations = atoi(value);
        if (strcmp(key, "-nt") == 0) omp_set_num_threads(atoi(value));
        if (strcmp(key, "-r") == 0) group.useRANSAC = atoi(value);
        if (strcmp(key, "-ri") == 0) group.numberOfRANSACIterations = atoi(value);
        if (strcmp(key, "-rs") == 0) group.RANSACMaxScale = atof(value);
        if (strcmp(key, "-rid") == 0) group.RANSACInlierDistance = atof(value);
        if (strcmp(key, "-rb") == 0) group.RANSACBatches = atoi(value);
        if (strcmp(key, "-s") == 0) group.useScale = atoi(value);
        if (strcmp(key, "-se") == 0) group.statsEpsilon = atof(value);
        if (strcmp(key, "-si") == 0) group.statIntervalUpdate = atoi(value);
        if (strcmp(key, "-ss") == 0) group.statsMaxSize = atoi(value);
        if (strcmp(key, "-t") == 0) group.inlierThreshold = atof(value);
        if (strcmp(key, "-ts") == 0) group.transformSubdirectory = std::string(value);
        if (strcmp(key, "-l") == 0) group.addLandmarks(value);          // at parse time, as upstream (frog.cxx:187-190)
        if (strcmp(key, "-lc") == 0) group.addLandmarks(value, true);   // frog.cxx:191-193
        if (strcmp(key, "-lcw") == 0) group.landmarksConstraintsWeight = atof(value);
        if (strcmp(key, "-mf") == 0) group.outputFileName = value;
        if (strcmp(key, "-wp") == 0) group.writePairs = atoi(value);
        if (strcmp(key, "-dev") == 0) group.device = atoi(value);
        if (strcmp(key, "-ng") == 0) { group.nGpus = atoi(value); group.loopback = false; }
        if (strcmp(key, "-ngl") == 0) { group.nGpus = atoi(value); group.loopback = true; }
        if (strcmp(key, "-q") == 0) group.quiet = atoi(value);
        if (strcmp(key, "-exact") == 0) group.exact = atoi(value);
        if (strcmp(key, "-j") == 0) {
            group.writeSingleFileTransforms = true;
            increment = 1;
        }
        argumentsIndex += increment;
    }

    group.run();                             // (its first HIP call waits for the runtime the thread above is bringing up)
    if (warm.joinable()) warm.join();
    auto end = std::chrono::system_clock::now();
    cout << "Iteration loops : " << group.loopIterations << " iterations in " << group.loopSeconds << "s" << endl;
    cout << "Total time : " << std::chrono::duration<float>(end - start).count() << "s" << endl;
    if (std::getenv("FROG_TIMING"))
        cout << "[timing] main returns at " << std::fixed << std::chrono::duration<double>(end.time_since_epoch()).count() << std::defaultfloat << endl;
    groupOwner.reset();                      // frog_destroy (device buffers, streams) and the host copy of pairs.bin
    if (std::getenv("FROG_TIMING"))
        cout << "[timing] group released : " << std::chrono::duration<double>(std::chrono::system_clock::now() - end).count() << "s" << endl;
    return 0;
}

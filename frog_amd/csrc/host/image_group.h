// image_group.h -- host-side ImageGroup of the MI355X build.
//
// Same public surface as the reference's class (registration/imageGroup.h:10-82):
// option fields with the reference's names and defaults, readPairs(), run().
// run() keeps the control flow of imageGroup.cxx:31-157 and delegates every
// numeric step to libfrog_hip.so (include/frog_hip.h).  No solver arithmetic and
// no CPU fallback live here.
#pragma once

#include <map>
#include <string>
#include <vector>

#include "../../../include/frog_hip.h"
#include "../../../include/frog_host.h"

class ImageGroup {
public:
    ImageGroup();
    ~ImageGroup();

    void run();
    void addLandmarks(const char *path, bool asConstraints = false);   // :1161 (validation landmarks only)
    void readPairs(const char *fileName);        // imageGroup.cxx:1353
    void usePairs(frog_pairs *p);                // adopt an already parsed / synthetic group

    // imageGroup.h:17-50 (same names, same defaults)
    bool printStats, printLinear;
    int linearIterations;
    bool useScale;
    int deformableLevels, deformableIterations;
    float linearInitializationAnchor[3];
    float linearAlpha, deformableAlpha;
    int statIntervalUpdate;
    float initialGridSize, boundingBoxMargin, inlierThreshold;
    bool guaranteeDiffeomorphism;
    float maxDisplacementRatio;
    bool invertLandmarksCoordinates;
    float landmarksConstraintsWeight;
    const char *outputFileName = "measures.csv";
    bool writePairs;
    int numberOfFixedImages;
    bool useRANSAC;
    int numberOfRANSACIterations;
    float RANSACInlierDistance, RANSACMaxScale;
    char *fixedTransformsDirectory;
    bool writeSingleFileTransforms;
    std::string transformSubdirectory, errorMapsSubdirectory;
    // Stats statics (stats.cxx:10-12), set by -ss / -emi / -se
    int statsMaxSize, statsMaxIterations;
    float statsEpsilon;

    int RANSACBatches = 0;       // candidate batches (new: -rb; 0 = omp_get_num_procs() as upstream, imageGroup.cxx:635)
    std::vector<std::pair<int, long long>> ransacInliers;   // (image, best census) for bbox.json
    int device = 0;              // HIP device ordinal (new: -dev); with -ng N the devices are device .. device + N - 1
    int nGpus = 1;               // new: -ng N, images sharded over N GPUs, one host thread per GPU, collectives over RCCL
    bool loopback = false;       // new: -ngl N, the same control flow with N contexts on ONE device and host-staged
                                 // collectives (rehearsal / tests on a single-GPU box)
    bool quiet = false;          // suppress per-iteration lines (new: -q)
    bool exact = false;          // new: -exact 1 = frog_options::reference_order: every solver loop in the reference's own order
                                 // and arithmetic (the CPU restatement's bits, raw coefficients included), ~100 x slower

    // results of run()
    struct Measure { float E, landmarkAv, landmarkMax, landmarkSTD; };
    struct Landmark { uint32_t image, point; };      // point = index inside the image (an appended, link-less point)
    std::map<std::string, std::vector<Landmark>> landmarks;
    std::vector<std::pair<Landmark, Landmark>> hardLinks;   // (point, partner) in upstream's push order (-lc)
    std::vector<float> landmarkXyz2;                 // xyz2 of all landmarks, in map order (filled per iteration)
    void fetchLandmarks();
    std::vector<Measure> measures;
    std::vector<int> gridsPerLevel;
    double loopSeconds = 0;      // time spent inside the two iteration loops
    int loopIterations = 0;

protected:
    frog_pairs *pairs = nullptr;
    bool ownPairs = false;
    frog_ctx *ctx = nullptr;     // rank 0's context (the only one with one GPU)
    std::vector<frog_counts> counts;
    // -ng N: one context and one communicator per rank, rank r owns images [shardBegin[r], shardBegin[r + 1])
    std::vector<frog_ctx *> ctxs;
    std::vector<struct frog_comm *> comms;
    std::vector<uint32_t> shardBegin;
    frog_ctx *ctxOf(uint32_t image) const;       // the context that owns (updates) `image`
    void planShards();
    void createShardedContexts();
    void runSharded();                           // run()'s loops, executed by one thread per rank in lockstep
    void finishRun();                            // :130-155, everything after the loops

    void createContext();
    void readAndApplyFixedImagesTransforms();    // :1419
    void check(int rc, const char *what);
    void computeLandmarkDistances(float e);      // :1229
    bool saveTransformedLandmarks();             // :1284
    void saveLandmarkDistances();                // :1318
    void displayStats();                         // :899
    void displayLinearTransforms();              // :600
    void countInliers();                         // :988
    void saveDistanceHistograms(const char *file);   // :850
    void saveMeasures(const char *file);         // :1475
    void saveTransforms();                       // :1458
    void saveErrorMaps();                        // :475
    void writeLinksDistances();                  // :924
    void saveStatsJSON();                        // :152-155, :1493-1511
};

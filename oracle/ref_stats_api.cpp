// ref_stats_api.cpp -- C ABI around the REFERENCE's own Stats class.
//
// TEST INFRASTRUCTURE.  This file contains no reference code: it #includes
// registration/stats.h from where it lies under /root/reference and is linked
// against registration/stats.cxx compiled unmodified from there (oracle/Makefile,
// target _ref/libfrog_refstats.so).  The output lives only in oracle/_ref/
// (git-ignored; it travels to the GPU box like any other built .so).
//
// Purpose: pin the oracle's restatement of Stats (reservoir sampling, EM fit,
// inlier probability, histogram) against the real thing, and generate
// tests/golden/stats_*.json (tests/golden/make_stats_golden.py).

#include "stats.h"      // -I/root/reference/registration

#include <algorithm>
#include <cstring>

extern "C" {

void *refstats_new(int max_size, int max_iterations, float epsilon)
{
    // statics shared by all instances, as in the reference (stats.cxx:10-12)
    Stats::maxSize = max_size;
    Stats::maxIterations = max_iterations;
    Stats::epsilon = epsilon;
    return new Stats();
}
void refstats_free(void *p) { delete (Stats *)p; }
void refstats_add_slots(void *p, int n) { Stats *s = (Stats *)p; for (int i = 0; i < n; i++) s->addSlot(); }
void refstats_reset(void *p) { ((Stats *)p)->reset(); }
void refstats_add_samples(void *p, const float *v, int n)
{
    Stats *s = (Stats *)p;
    for (int i = 0; i < n; i++) s->addSample(v[i]);
}
void refstats_estimate(void *p) { ((Stats *)p)->estimateDistribution(); }
float refstats_inlier_probability(void *p, float d) { return ((Stats *)p)->getInlierProbability(d); }
void refstats_inlier_probability_n(void *p, const float *d, int n, float *out)
{
    Stats *s = (Stats *)p;
    for (int i = 0; i < n; i++) out[i] = s->getInlierProbability(d[i]);
}
void refstats_get_params(void *p, float o[3]) { Stats *s = (Stats *)p; o[0] = s->c1; o[1] = s->c2; o[2] = s->ratio; }
void refstats_set_params(void *p, const float i[3]) { Stats *s = (Stats *)p; s->c1 = i[0]; s->c2 = i[1]; s->ratio = i[2]; }
int refstats_size(void *p) { return ((Stats *)p)->size; }
int refstats_get_samples(void *p, float *out, int cap)
{
    Stats *s = (Stats *)p;
    int n = std::min(cap, s->size);
    std::memcpy(out, s->samples.data(), (size_t)n * sizeof(float));
    return s->size;
}
int refstats_histogram(void *p, float bin, float *out, int cap)
{
    Stats *s = (Stats *)p;
    s->getHistogram(bin);
    int n = std::min<int>(cap, (int)s->histogram.size());
    std::memcpy(out, s->histogram.data(), (size_t)n * sizeof(float));
    return (int)s->histogram.size();
}
float refstats_chipdf(float x) { return chipdf(x); }

}

// rank_schedule.cpp -- frog_run_schedule (include/frog_host.h): one rank's share of a timed registration schedule, the
// loop bodies of ImageGroup::run (registration/imageGroup.cxx:54-66, :78-128) over the C ABI of libfrog_hip.so and, when
// the images are sharded, the collectives of libfrog_comm.so.  No solver arithmetic here.

#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/frog_host.h"
#include "comm_api.h"

namespace frog { void set_last_error(const std::string &s); }      // libfrog_hip.so

namespace {
using clk = std::chrono::steady_clock;
double since(clk::time_point t) { return std::chrono::duration<double>(clk::now() - t).count(); }

struct Runner {
    frog_ctx *c;
    frog_comm *cm;
    CommApi *api;
    const frog_schedule_plan *plan;
    frog_schedule_result *out;
    bool whole = false;             // the context owns every image and there is no communicator: the plain entry points
    // Two collectives per deformable iteration, one per linear iteration (include/frog_hip.h frog_comm_mode): the energy sums ride
    // on the all-reduce of the proposal sums, the oversize count on the coordinate gather, whose transform is queued speculatively.
    // FROG_THREE_COLLECTIVES=1 keeps the flow of rounds 2-4 (three / two), for comparison.
    bool two = false;
    bool gathered = false;          // the step just finished has already transformed and gathered: the transformPoints() that follows it is done
    // ... and the NEXT step's phase A is queued before this step's decision has reached the host (frog_step_speculate): the GPU
    // does not wait for the host's read + launch latency once per iteration.  FROG_NO_SPECULATION=1 switches it off.
    bool speculate = false;
    bool phaseAQueued = false;      // the coming deformableStep finds its phase A already in the queue
    uint32_t ib = 0, ie = 0, nI = 0;
    int rc = 0;

    bool ok(int r) { if (r && !rc) rc = r; return rc == 0; }

    void transformPoints(int apply)
    {
        if (rc) return;
        if (whole) { ok(frog_transform_points(c, apply)); return; }
        if (two) {
            const bool done = gathered && !apply;
            gathered = false;
            if (!done) ok(api->gather_points(cm, apply, 0, 0u));
            return;
        }
        if (!ok(frog_transform_points_local(c, apply))) return;
        if (cm) ok(api->all_gather_xyz2(cm));
    }
    void updateStats()
    {
        if (rc) return;
        if (whole) { ok(frog_update_stats(c)); return; }
        if (!ok(frog_update_stats_local(c))) return;
        if (cm) { if (!ok(api->all_reduce(cm, FROG_BUF_EM))) return; }
        else if (plan->proxy_em) {
            if (!ok(frog_set_em_rows(c, plan->proxy_em, 0, ib)) || !ok(frog_set_em_rows(c, plan->proxy_em, ie, nI))) return;
        }
        ok(frog_stats_publish(c));
    }
    double linearStep()
    {
        double E = 0;
        if (rc) return E;
        if (whole) { ok(frog_linear_step(c, &E)); return E; }
        if (!ok(frog_linear_step_local(c))) return E;
        if (two) {
            // the step's two sums and its list flag travel in the trailers of the coordinate gather (imageGroup.cxx:1147 needs
            // them for the printed E only; the matrices are image-local): ONE collective per linear iteration
            if (!ok(api->gather_points(cm, 0, 0, 0xBu))) return E;
            ok(frog_step_finish(c, &E));
            gathered = true;
            return E;
        }
        if (cm && !ok(api->all_reduce(cm, FROG_BUF_ENERGY))) return E;
        ok(frog_energy_read(c, &E, nullptr));
        return E;
    }
    void setup(int level)
    {
        if (rc) return;
        const auto t0 = clk::now();
        frog_grid_info info{};
        if (whole) ok(frog_deformable_setup(c, level, &info));
        else {
            double mn[3], mx[3];
            if (!ok(frog_bounds_local(c, mn, mx))) return;
            if (cm && !ok(api->all_reduce_bounds(cm, mn, mx))) return;
            ok(frog_deformable_setup_bounds(c, level, mn, mx, &info));
        }
        if (rc) return;
        if (out->n_lattices < FROG_SCHEDULE_MAX_LATTICES) {
            frog_schedule_lattice &la = out->lattices[out->n_lattices++];
            la.level = level; la.iterations = 0; la.setup_host_s = since(t0);
            for (int k = 0; k < 3; k++) la.dims[k] = info.dims[k];
        }
    }
    // next_plain: the iteration after this one, if this one is accepted, is an ordinary one (same level, no statistics refresh first)
    double deformableStep(float alpha, bool next_plain)
    {
        double E = 0;
        if (rc) return E;
        if (whole) { ok(frog_deformable_step(c, alpha, &E)); return E; }
        const bool queued = phaseAQueued;
        phaseAQueued = false;
        if (!queued && !ok(frog_deformable_phase_a(c, alpha))) return E;
        if (cm && !ok(api->all_reduce(cm, FROG_BUF_GRIDSUM))) return E;        // the shared common-space grid, :400-432 (+ the energy sums when `two`)
        if (!ok(frog_deformable_phase_b(c))) return E;
        if (two) {
            // the transform that follows the step, queued before the group's oversize count exists (each rank goes by its own),
            // the counts in the gather's trailers; decision and commit once they are added up.  A rejected step has left
            // speculative coordinates in the replicas: run()'s reject path re-bases and gathers before anything reads them.
            if (!ok(api->gather_points(cm, 0, 1, 0x4u))) return E;
            if (speculate && next_plain) {
                if (!ok(frog_step_speculate(c)) || !ok(frog_deformable_phase_a(c, alpha))) return E;
                phaseAQueued = true;
            }
            ok(frog_step_finish(c, &E));
            gathered = (float)E >= 0;
            if (!gathered) phaseAQueued = false;        // rejected: frog_step_finish has rolled the speculation back
            return E;
        }
        if (cm && !ok(api->all_reduce(cm, FROG_BUF_ENERGY))) return E;         // energy sums + oversize count
        ok(frog_deformable_phase_c(c, &E));
        return E;
    }
    void barrier()
    {
        if (rc) return;
        if (!ok(frog_synchronize(c))) return;
        if (cm) ok(api->barrier(cm));
    }
    void phaseEnd(int phase, clk::time_point t0)
    {
        if (rc) return;
        if (plan->profile == 1) {
            if (!ok(frog_synchronize(c))) return;
            out->phase_s[phase] = since(t0);
            frog_kernel_time kt[FROG_K_COUNT_];
            if (!ok(frog_profile_read(c, kt, 1))) return;
            for (int k = 0; k < FROG_K_COUNT_; k++) {
                out->kernels_by_phase[phase][k] = kt[k];
                out->kernels[k].ms_total += kt[k].ms_total; out->kernels[k].launches += kt[k].launches;
            }
        } else {
            out->phase_s[phase] = since(t0);
        }
    }
};
} // namespace

extern "C" int frog_run_schedule(frog_ctx *ctx, frog_comm *comm, const frog_schedule_plan *plan, frog_schedule_result *out)
{
    if (!ctx || !plan || !out) { frog::set_last_error("null argument"); return FROG_E_INVALID; }
    if (plan->plan_bytes != sizeof(frog_schedule_plan) || plan->result_bytes != sizeof(frog_schedule_result)) {
        frog::set_last_error("frog_schedule_plan / frog_schedule_result layout differs from the caller's");
        return FROG_E_INVALID;
    }
    if (plan->n_levels < 0 || plan->n_levels > FROG_SCHEDULE_MAX_LEVELS || plan->stat_interval < 1 || plan->warmup_linear < 0 || plan->linear < 0) {
        frog::set_last_error("bad schedule");
        return FROG_E_INVALID;
    }
    std::memset(out, 0, sizeof *out);
    Runner r{ ctx, comm, nullptr, plan, out };
    if (comm) {
        std::string err;
        if (!host_comm_api().load(err)) { frog::set_last_error("cannot load libfrog_comm.so: " + err); return FROG_E_INVALID; }
        r.api = &host_comm_api();
    }
    r.nI = frog_num_images(ctx);
    {
        size_t b = 0, e = 0;
        int rc = frog_comm_buffer(ctx, FROG_BUF_EM, nullptr, nullptr, &b, &e);
        if (rc) return rc;
        r.ib = (uint32_t)b; r.ie = (uint32_t)e;
    }
    r.whole = !comm && r.ib == 0 && r.ie == r.nI;
    r.two = comm && !getenv("FROG_THREE_COLLECTIVES");
    r.speculate = r.two && !getenv("FROG_NO_SPECULATION");
    if (comm) { const int rc = frog_comm_mode(ctx, r.two ? 1 : 0); if (rc) return rc; }

    // ---- untimed: linear set-up, first transform, (proxy: the other ranks' coordinates), warm-up iterations
    const bool trace = getenv("FROG_SCHEDULE_TRACE") != nullptr;
    const auto t_in = clk::now();
    r.ok(frog_linear_init(ctx, plan->anchor));
    if (trace) std::fprintf(stderr, "[schedule] linear_init returned at %.4f s\n", since(t_in));
    r.transformPoints(0);
    if (trace) { frog_synchronize(ctx); std::fprintf(stderr, "[schedule] first transform done at %.4f s\n", since(t_in)); }
    if (!r.rc && plan->proxy_xyz2 && !comm && !r.whole) {
        size_t pb = 0, pe = 0;
        const uint64_t P = frog_num_points(ctx);
        std::vector<float> cur(3 * P);
        r.ok(frog_comm_buffer(ctx, FROG_BUF_XYZ2, nullptr, nullptr, &pb, &pe));
        r.ok(frog_get_points(ctx, nullptr, cur.data()));
        if (!r.rc) {
            std::memcpy(cur.data(), plan->proxy_xyz2, 3 * pb * sizeof(float));
            std::memcpy(cur.data() + 3 * pe, plan->proxy_xyz2 + 3 * pe, 3 * (P - pe) * sizeof(float));
            r.ok(frog_set_points2(ctx, cur.data()));
        }
    }
    int it = 0;
    double E = 0;
    for (int w = 0; w < plan->warmup_linear && !r.rc; w++, it++) {
        if (it % plan->stat_interval == 0) r.updateStats();
        E = r.linearStep();
        r.transformPoints(0);
    }

    if (trace) { frog_synchronize(ctx); std::fprintf(stderr, "[schedule] warm-up done at %.4f s\n", since(t_in)); }
    // ---- timed region
    r.ok(frog_profile_enable(ctx, plan->profile));
    if (comm && plan->time_comm) r.ok(r.api->timing(comm, 1));
    r.barrier();
    const auto t_start = clk::now();
    auto tp = t_start;
    for (int k = 0; k < plan->linear && !r.rc; k++, it++) {
        if (it % plan->stat_interval == 0) r.updateStats();
        E = r.linearStep();
        r.transformPoints(0);
        out->iterations++;
    }
    r.transformPoints(1);                                       // :70
    r.phaseEnd(0, tp);
    for (int level = 0; level < plan->n_levels && !r.rc; level++) {
        const int n = plan->per_level[level];
        if (n <= 0) continue;
        tp = clk::now();
        r.setup(level);                                         // :81
        r.transformPoints(0);
        int grids = 1, diffeo = 0;
        float alpha = plan->deformable_alpha;
        for (int iteration = 0; iteration < n && !r.rc; iteration++) {
            if (iteration % plan->stat_interval == 0) r.updateStats();
            const double e = r.deformableStep(alpha, iteration + 1 < n && (iteration + 1) % plan->stat_interval != 0);
            if (r.rc) break;
            if ((float)e < 0) {                                 // :97-115
                if (diffeo == 0) alpha /= 2;
                grids++;
                iteration--;
                r.transformPoints(1);
                r.setup(level);
                r.transformPoints(0);
                diffeo = 0;
                continue;
            }
            diffeo++;
            r.transformPoints(0);
            E = e;
            out->iterations++;
            if (out->n_lattices > 0) out->lattices[out->n_lattices - 1].iterations++;
        }
        out->grids_per_level[level] = grids;
        r.transformPoints(1);                                   // :126
        r.phaseEnd(1 + level, tp);
    }
    r.barrier();
    out->elapsed_s = since(t_start);
    if (trace) std::fprintf(stderr, "[schedule] timed region %.4f s, ends at %.4f s\n", out->elapsed_s, since(t_in));
    out->final_E = (double)(float)E;
    if (r.rc) return r.rc;

    // ---- after the timed region: kernel and collective times, the replica's hash
    if (plan->profile != 1) r.ok(frog_profile_read(ctx, out->kernels, 1));
    r.ok(frog_profile_enable(ctx, 0));
    if (comm && plan->time_comm) {
        r.ok(r.api->timing_read(comm, out->comm_ms, out->comm_calls, out->comm_sampled));
        r.ok(r.api->timing(comm, 0));
    }
    if (plan->n_levels == 0) r.transformPoints(1);      // linear only: leave re-based coordinates, as the levels do
    if (r.rc) return r.rc;
    {
        const uint64_t P = frog_num_points(ctx);
        std::vector<float> xyz2(3 * P), em(3 * (size_t)r.nI);
        r.ok(frog_get_points(ctx, nullptr, xyz2.data()));
        for (uint32_t i = 0; i < r.nI && !r.rc; i++) r.ok(frog_get_em(ctx, i, &em[3 * (size_t)i]));
        uint64_t h = 1469598103934665603ull;                    // FNV-1a over the bit patterns
        auto eat = [&](const std::vector<float> &v) { for (float f : v) { uint32_t b; std::memcpy(&b, &f, 4); h = (h ^ b) * 1099511628211ull; } };
        eat(xyz2); eat(em);
        out->replica_hash = h;
    }
    return r.rc;
}

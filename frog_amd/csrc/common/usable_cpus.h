// usable_cpus.h -- how many host threads a parallel loop should start: the CPUs this process may actually run on.
//
// omp_get_num_procs() answers with the machine's (or the affinity mask's) CPUs.  In a container with a CPU quota -- the GPU
// boxes this is measured on: 256 hardware threads visible, cgroup cpu.max = 16 CPUs -- that starts 256 threads which the
// scheduler then throttles together for most of every 100 ms period: readPairs' counting pass took 0.097 s there against
// 0.016 s on an 8-CPU container without a quota.  So: min(affinity mask, cgroup quota rounded up), and OMP_NUM_THREADS /
// omp_set_num_threads (bin/frog -nt) still win when they ask for fewer.
#pragma once

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include <omp.h>
#include <sched.h>

namespace frog {

inline int cgroup_cpu_quota()                   // CPUs the cgroup allows (rounded up), 0 = no quota found
{
    auto quota_of = [](const std::string &dir) -> double {
        double q = 0, p = 0;
        char word[64] = { 0 };
        if (FILE *f = std::fopen((dir + "/cpu.max").c_str(), "r")) {                      // cgroup v2: "<quota|max> <period>"
            const int n = std::fscanf(f, "%63s %lf", word, &p);
            std::fclose(f);
            if (n == 2 && std::strcmp(word, "max") != 0 && p > 0) { q = std::atof(word); return q > 0 ? q / p : 0; }
            return 0;
        }
        if (FILE *f = std::fopen((dir + "/cpu.cfs_quota_us").c_str(), "r")) {              // cgroup v1
            const int n = std::fscanf(f, "%lf", &q);
            std::fclose(f);
            if (n != 1 || q <= 0) return 0;
            if (FILE *g = std::fopen((dir + "/cpu.cfs_period_us").c_str(), "r")) {
                const int m = std::fscanf(g, "%lf", &p);
                std::fclose(g);
                if (m == 1 && p > 0) return q / p;
            }
        }
        return 0;
    };
    double best = 0;
    auto take = [&](double v) { if (v > 0 && (best == 0 || v < best)) best = v; };
    take(quota_of("/sys/fs/cgroup"));
    take(quota_of("/sys/fs/cgroup/cpu"));
    // the process's own group and its parents, where the hierarchy is visible (a quota anywhere on the path binds)
    if (FILE *f = std::fopen("/proc/self/cgroup", "r")) {
        char line[4096];
        while (std::fgets(line, sizeof line, f)) {
            std::string s(line);
            while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
            const size_t c1 = s.find(':'), c2 = c1 == std::string::npos ? c1 : s.find(':', c1 + 1);
            if (c2 == std::string::npos) continue;
            const std::string ctrl = s.substr(c1 + 1, c2 - c1 - 1);
            std::string path = s.substr(c2 + 1);
            std::string root;
            if (ctrl.empty()) root = "/sys/fs/cgroup";                                      // v2
            else if (ctrl.find("cpu") != std::string::npos && ctrl.find("cpuset") == std::string::npos) root = "/sys/fs/cgroup/cpu";
            else continue;
            while (path.size() > 1) {
                take(quota_of(root + path));
                const size_t slash = path.find_last_of('/');
                path = slash == std::string::npos || slash == 0 ? std::string("/") : path.substr(0, slash);
            }
        }
        std::fclose(f);
    }
    return best > 0 ? std::max(1, (int)(best + 0.999)) : 0;
}

inline int usable_cpus()
{
    static const int n = [] {
        int n = (int)std::thread::hardware_concurrency();
        if (n < 1) n = 1;
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min(n, c); }
        const int q = cgroup_cpu_quota();
        if (q > 0) n = std::min(n, q);
        return std::max(1, n);
    }();
    return n;
}

// threads for one parallel region: what OpenMP would start (OMP_NUM_THREADS, omp_set_num_threads), capped by the usable CPUs
inline int host_threads()
{
    const int n = std::min(std::min(omp_get_max_threads(), usable_cpus()), 64);
    return n < 1 ? 1 : n;
}

} // namespace frog

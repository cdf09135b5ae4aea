// comm_api.h -- libfrog_comm.so (include/frog_comm.h) as this library sees it: loaded on demand with dlopen, so that a
// process that never shards over GPUs never maps RCCL.
#pragma once

#include <string>

#include "../../../include/frog_comm.h"

struct CommApi {
    decltype(&frog_comm_create_rccl) create_rccl = nullptr;
    decltype(&frog_comm_create_loopback) create_loopback = nullptr;
    decltype(&frog_comm_destroy_all) destroy_all = nullptr;
    decltype(&frog_comm_bind) bind = nullptr;
    decltype(&frog_comm_all_gather_xyz2) all_gather_xyz2 = nullptr;
    decltype(&frog_comm_gather_points) gather_points = nullptr;
    decltype(&frog_comm_all_reduce) all_reduce = nullptr;
    decltype(&frog_comm_all_reduce_bounds) all_reduce_bounds = nullptr;
    decltype(&frog_comm_barrier) barrier = nullptr;
    decltype(&frog_comm_timing) timing = nullptr;
    decltype(&frog_comm_timing_read) timing_read = nullptr;
    bool load(std::string &err);
};
CommApi &host_comm_api();

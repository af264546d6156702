// Multi-GPU part of the C-ABI: one process per GPU, characters sharded over the ranks (pastml/acr.py:226-231 runs the
// characters independently), and ONE collective on the path: the sum of the per-rank log-likelihoods over RCCL / xGMI
// (SURVEY.md 8b item 9, 8e).  librccl is opened at run time (dlopen) the first time a communicator is asked for, so a
// single-GPU user never loads it; the communicator works on the ctx's own stream: the all-reduce is ordered after the
// sweep that produced the values without any host synchronisation in between.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only; the entry points are resolved with dlsym

struct PmlRccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;   // optional (reports only)
    std::string error;
};

static PmlRccl* pml_rccl() {
    static PmlRccl r;
    static bool tried = false;
    if (tried) return &r;
    tried = true;
    const char* names[] = {getenv("PASTML_HIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* name : names) {
        if (!name || !*name) continue;
        r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
        r.error = dlerror();
    }
    if (!r.handle) return &r;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
    r.CommCount = (decltype(r.CommCount))dlsym(r.handle, "ncclCommCount");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) {
        r.error = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
        dlclose(r.handle);
        r.handle = nullptr;
    }
    return &r;
}

struct PmlComm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    double* d_buf = nullptr;  // device staging of the reduced values
    double* h_buf = nullptr;  // pinned
    size_t cap = 0;
    // pml_marginal_pass with a communicator: sum of the rank's log-likelihoods formed on the device and all-reduced on
    // the sweep's stream; the result waits here for pml_loglik_total
    double* d_total = nullptr;
    double* h_total = nullptr;  // pinned
    bool total_fresh = false;
};

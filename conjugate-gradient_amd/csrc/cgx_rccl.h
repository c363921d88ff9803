// cgx_rccl.h -- run-time binding to RCCL (librccl.so.1), the MI355X replacement for the reference's
// MPI collectives (code/MPI/cg.cc:87-88,92,106,117,135-136,140-142).
//
// RCCL is bound with dlopen instead of a link-time dependency so that (a) the single-GPU path and the
// CPU-side symbol checks work on machines where RCCL cannot initialise, and (b) inside a Python process
// that already imported torch, the SAME librccl/libamdhip64 instance torch loaded is reused (both have
// SONAME librccl.so.1 / libamdhip64.so.7), never a second HIP runtime.
#pragma once

#include <rccl/rccl.h>

#include <string>

namespace cgx {

struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    void *handle = nullptr;
};

// Returns the process-wide API table, loading librccl on first use; nullptr + err on failure.
const RcclApi *rccl_api(std::string *err);

}  // namespace cgx

// cgx_rccl.cpp -- dlopen binding of the RCCL entry points the CG path uses.
#include "cgx_rccl.h"

#include <dlfcn.h>

#include <mutex>

namespace cgx {

static RcclApi g_api;
static bool g_tried = false;
static std::string g_err;
static std::mutex g_mu;

template <typename F>
static bool bind(void *h, const char *name, F *out, std::string *err)
{
    void *sym = dlsym(h, name);
    if (!sym) {
        *err = std::string("librccl: missing symbol ") + name;
        return false;
    }
    *out = reinterpret_cast<F>(sym);
    return true;
}

const RcclApi *rccl_api(std::string *err)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_tried) {
        g_tried = true;
        // SONAME first: resolves to the instance already mapped into this process (torch's, if any).
        const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        void *h = nullptr;
        for (const char *nm : names) {
            h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) {
            const char *e = dlerror();
            g_err = std::string("cannot load librccl.so.1: ") + (e ? e : "unknown");
        } else {
            RcclApi a;
            a.handle = h;
            bool ok = bind(h, "ncclGetUniqueId", &a.GetUniqueId, &g_err) &&
                      bind(h, "ncclCommInitRank", &a.CommInitRank, &g_err) &&
                      bind(h, "ncclCommDestroy", &a.CommDestroy, &g_err) &&
                      bind(h, "ncclAllGather", &a.AllGather, &g_err) &&
                      bind(h, "ncclBroadcast", &a.Broadcast, &g_err) &&
                      bind(h, "ncclGroupStart", &a.GroupStart, &g_err) &&
                      bind(h, "ncclGroupEnd", &a.GroupEnd, &g_err) &&
                      bind(h, "ncclGetErrorString", &a.GetErrorString, &g_err) &&
                      bind(h, "ncclGetVersion", &a.GetVersion, &g_err) &&
                      bind(h, "ncclCommCount", &a.CommCount, &g_err) &&
                      bind(h, "ncclCommUserRank", &a.CommUserRank, &g_err);
            if (ok) g_api = a;
        }
    }
    if (!g_api.handle) {
        if (err) *err = g_err;
        return nullptr;
    }
    return &g_api;
}

}  // namespace cgx

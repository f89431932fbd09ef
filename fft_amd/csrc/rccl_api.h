// rccl_api.h — librccl resolved with dlopen / dlsym when an RCCL path is first asked for: libspecengine.so carries no
// link-time dependency on it (a one-GPU GUI host needs no collective library), and callers get SA_EUNSUP if it is absent.
#pragma once
#include <dlfcn.h>

#include <rccl/rccl.h>       // types and prototypes only

namespace sa {

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllToAll) AllToAll = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
inline const RcclApi* rccl_api() {
    static const RcclApi api = [] {
        RcclApi a;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (a.lib) break;
        }
        if (!a.lib) return a;
        a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(a.lib, "ncclCommInitAll"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(a.lib, "ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(a.lib, "ncclGroupEnd"));
        a.Send = reinterpret_cast<decltype(a.Send)>(dlsym(a.lib, "ncclSend"));
        a.Recv = reinterpret_cast<decltype(a.Recv)>(dlsym(a.lib, "ncclRecv"));
        a.AllToAll = reinterpret_cast<decltype(a.AllToAll)>(dlsym(a.lib, "ncclAllToAll"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
        a.ok = a.CommInitAll && a.CommInitRank && a.GetUniqueId && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Send && a.Recv &&
               a.AllToAll && a.GetErrorString;
        return a;
    }();
    return api.ok ? &api : nullptr;
}

}  // namespace sa

// rt_rccl.hpp -- RCCL (rccl.h) bound at first use.
//
// The tile gather of a frame sharded over several GPUs is the only collective of the path; a
// single-GPU tracer never needs it.  librccl.so is 570 MB of code objects, so the library does not carry it
// as a load-time dependency: the entry points below are resolved with dlopen/dlsym when the first
// multi-device tracer (rt_tracer_create_multi) or group member (rt_tracer_join_group) is created.  A copy
// that is already mapped into the process (e.g. by a PyTorch wheel) is reused, so that one process never
// holds two RCCLs.  Types and signatures come from rccl.h itself.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

namespace rtc {

struct Rccl {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;          // optional (reporting only)
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  void* handle = nullptr;
  std::string why;        // why loading failed
  std::string path;       // what was loaded

  bool ok() const { return handle != nullptr; }

  // process-wide instance; the first call loads the library
  static Rccl& get() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { r.load(); });
    return r;
  }

 private:
  template <class F>
  bool sym(F& f, const char* name) {
    f = reinterpret_cast<F>(dlsym(handle, name));
    if (f == nullptr) { why = std::string("librccl: missing symbol ") + name; return false; }
    return true;
  }

  void load() {
    const char* env = getenv("RT_MI355X_RCCL");          // explicit path (tests, unusual installs)
    const char* names[] = {env, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);   // a copy this process already uses
    if (h != nullptr) path = "librccl.so.1 (already mapped)";
    for (size_t i = 0; h == nullptr && i < sizeof names / sizeof names[0]; ++i) {
      if (names[i] == nullptr || names[i][0] == '\0') continue;
      h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
      if (h != nullptr) path = names[i];
      else if (const char* e = dlerror()) why = e;
    }
    if (h == nullptr) { if (why.empty()) why = "librccl.so.1 not found"; return; }
    handle = h;
    const bool all = sym(GetUniqueId, "ncclGetUniqueId") && sym(CommInitRank, "ncclCommInitRank") &&
                     sym(CommInitAll, "ncclCommInitAll") && sym(CommDestroy, "ncclCommDestroy") &&
                     sym(GroupStart, "ncclGroupStart") && sym(GroupEnd, "ncclGroupEnd") && sym(Send, "ncclSend") &&
                     sym(Recv, "ncclRecv") && sym(GetErrorString, "ncclGetErrorString");
    if (!all) { handle = nullptr; return; }
    GetVersion = reinterpret_cast<decltype(GetVersion)>(dlsym(handle, "ncclGetVersion"));
    CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(handle, "ncclCommCount"));
    CommCuDevice = reinterpret_cast<decltype(CommCuDevice)>(dlsym(handle, "ncclCommCuDevice"));
    CommUserRank = reinterpret_cast<decltype(CommUserRank)>(dlsym(handle, "ncclCommUserRank"));
  }
};

}  // namespace rtc

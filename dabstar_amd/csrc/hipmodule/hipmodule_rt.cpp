// hipmodule_rt.cpp -- the "hipModule" form of libdabx (north_star: "a thin C-ABI hipModule shim").
//
// Build (dabstar_amd/build.py, build_hipmodule): every csrc/*.hip is compiled TWICE -- `--cuda-device-only` into a code object
// dabx_gfx950_<name>.hsaco, and `--cuda-host-only` into a host object that carries no device code at all -- and the host
// objects, the .cpp files and this file are linked into hipmodule/libdabx.so.  Same sources, same C ABI (include/dabx.h); the
// kernels reach the GPU through hipModuleLoad + hipModuleLaunchKernel instead of the fat binary hipcc embeds by default.
//
// How, without touching a launch site: for `kernel<<<grid, block, shm, stream>>>(args...)` clang's host side calls
// __hipPushCallConfiguration, then the kernel's host stub, which pops the configuration and calls
// hipLaunchKernel(handle, grid, block, void *args[], shm, stream); a module constructor announces every kernel with
// __hipRegisterFunction(fatbin handle, handle, ..., device name, ...).  Those six entry points (and hipMemcpyToSymbol /
// __hipRegisterVar for the three device variables of the timeline tool) are DEFINED HERE and bound inside the library
// (-Bsymbolic, not exported): registration only records handle -> device name, and a launch looks the name up in the code
// objects loaded from the library's own directory and calls hipModuleLaunchKernel with the very args[] array the stub built.
// The HIP runtime proper (libamdhip64) is used for everything else, unchanged.
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <dirent.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct Registry {
  std::mutex mu;
  std::unordered_map<const void *, std::string> kernels, vars;     // host handle -> device symbol
  std::unordered_map<const void *, hipFunction_t> fn;              // resolved per handle, all of them when the code objects are loaded (self_check)
  std::vector<hipModule_t> modules;
  bool loaded = false;
  int device = -1;                                                 // the device the code objects were loaded for: ONE device per process in this form
  std::string error;
};
Registry &reg() { static Registry *r = new Registry(); return *r; }   // never destroyed: kernels are launched from static destructors' siblings

struct Cfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local std::vector<Cfg> t_stack;

std::string own_dir()
{
  Dl_info info;
  if (!dladdr((const void *)&own_dir, &info) || !info.dli_fname) return ".";
  std::string p(info.dli_fname);
  const size_t k = p.rfind('/');
  return k == std::string::npos ? "." : p.substr(0, k);
}

// loads every dabx_gfx950_*.hsaco that lies next to the library (called with the registry locked)
hipError_t load_modules(Registry &r)
{
  if (r.loaded) return r.modules.empty() ? hipErrorFileNotFound : hipSuccess;
  r.loaded = true;
  const std::string dir = own_dir();
  if (DIR *d = opendir(dir.c_str())) {
    std::vector<std::string> names;
    while (dirent *e = readdir(d)) {
      const std::string n(e->d_name);
      if (n.rfind("dabx_gfx950_", 0) == 0 && n.size() > 6 && n.compare(n.size() - 6, 6, ".hsaco") == 0) names.push_back(n);
    }
    closedir(d);
    for (const std::string &n : names) {
      hipModule_t m;
      const hipError_t err = hipModuleLoad(&m, (dir + "/" + n).c_str());
      if (err != hipSuccess) { r.error = "hipModuleLoad(" + n + ") failed: " + hipGetErrorString(err); std::fprintf(stderr, "libdabx (hipModule): %s\n", r.error.c_str()); return err; }
      r.modules.push_back(m);
    }
  }
  if (r.modules.empty()) {
    r.error = "no dabx_gfx950_*.hsaco next to the library in " + dir;
    std::fprintf(stderr, "libdabx (hipModule): %s\n", r.error.c_str());
    return hipErrorFileNotFound;
  }
  (void)hipGetDevice(&r.device);
  // Self-check, once, at load: EVERY kernel the host stubs have registered (clang's module constructors ran before main) must be in one
  // of the code objects.  This form leans on clang's private host-stub ABI (__hipRegisterFunction, __hipPushCallConfiguration, the
  // stub's call of hipLaunchKernel): a toolchain that changes it -- stubs that register under other names, or not at all -- must stop the
  // process here, with the versions, not surface later as one launch that fails.
  std::string missing;
  for (const auto &kv : r.kernels) {
    hipFunction_t f = nullptr;
    for (hipModule_t m : r.modules)
      if (hipModuleGetFunction(&f, m, kv.second.c_str()) == hipSuccess && f) break;
    if (f) r.fn[kv.first] = f;
    else missing += (missing.empty() ? "" : ", ") + kv.second;
  }
  (void)hipGetLastError();                                 // the look-ups in the other code objects failed by design
  if (r.kernels.empty() || !missing.empty()) {
    int rt = 0, drv = 0;
    (void)hipRuntimeGetVersion(&rt);
    (void)hipDriverGetVersion(&drv);
    std::fprintf(stderr, "libdabx (hipModule): FATAL: %s (HIP runtime %d, driver %d, built with clang %s; %zu kernels registered, %zu code objects in %s).\n"
                         "The hipModule form depends on clang's host-stub ABI; use the default build (dabstar_amd/libdabx.so) with this toolchain.\n",
                 r.kernels.empty() ? "the host stubs registered no kernel at all" : ("kernels registered by the host stubs are in none of the code objects: " + missing).c_str(),
                 rt, drv, __clang_version__, r.kernels.size(), r.modules.size(), dir.c_str());
    std::abort();
  }
  return hipSuccess;
}

}  // namespace

extern "C" {

void **__hipRegisterFatBinary(const void *) { static void *handle = nullptr; return &handle; }     // (the host objects' fat binary is a stand-in symbol)
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *host_fun, char *, const char *device_name, unsigned, void *, void *, void *, void *, int *)
{
  Registry &r = reg();
  std::lock_guard<std::mutex> lk(r.mu);
  r.kernels[host_fun] = device_name;
}
void __hipRegisterVar(void **, void *var, char *, char *device_name, int, size_t, int, int)
{
  Registry &r = reg();
  std::lock_guard<std::mutex> lk(r.mu);
  r.vars[var] = device_name;
}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream)
{
  t_stack.push_back(Cfg{grid, block, shmem, stream});
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream)
{
  if (t_stack.empty()) return hipErrorInvalidValue;
  const Cfg c = t_stack.back();
  t_stack.pop_back();
  *grid = c.grid; *block = c.block; *shmem = c.shmem; *stream = c.stream;
  return hipSuccess;
}

hipError_t hipLaunchKernel(const void *handle, dim3 grid, dim3 block, void **args, size_t shmem, hipStream_t stream)
{
  Registry &r = reg();
  hipFunction_t f = nullptr;
  {
    std::lock_guard<std::mutex> lk(r.mu);
    auto hit = r.fn.find(handle);
    if (hit != r.fn.end()) f = hit->second;
    else {
      auto it = r.kernels.find(handle);
      if (it == r.kernels.end()) { std::fprintf(stderr, "libdabx (hipModule): launch of a kernel no host stub has registered\n"); return hipErrorInvalidDeviceFunction; }
      if (hipError_t err = load_modules(r)) return err;     // resolves every registered kernel (or aborts)
      hit = r.fn.find(handle);
      if (hit == r.fn.end()) return hipErrorInvalidDeviceFunction;
      f = hit->second;
    }
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != r.device) {
      std::fprintf(stderr, "libdabx (hipModule): the code objects were loaded for device %d, this launch is for device %d: one device per process in this form "
                           "(run one process per GPU, as bench.py does, or use the default build)\n", r.device, cur);
      return hipErrorInvalidDevice;
    }
  }
  return hipModuleLaunchKernel(f, grid.x, grid.y, grid.z, block.x, block.y, block.z, (unsigned)shmem, stream, args, nullptr);
}

hipError_t hipMemcpyToSymbol(const void *symbol, const void *src, size_t bytes, size_t offset, hipMemcpyKind kind)
{
  Registry &r = reg();
  hipDeviceptr_t dptr = nullptr;
  size_t size = 0;
  {
    std::lock_guard<std::mutex> lk(r.mu);
    auto it = r.vars.find(symbol);
    if (it == r.vars.end()) return hipErrorInvalidSymbol;
    if (hipError_t err = load_modules(r)) return err;
    bool found = false;
    for (hipModule_t m : r.modules)
      if (hipModuleGetGlobal(&dptr, &size, m, it->second.c_str()) == hipSuccess) { found = true; break; }
    (void)hipGetLastError();
    if (!found) return hipErrorInvalidSymbol;
  }
  if (offset + bytes > size) return hipErrorInvalidValue;
  return hipMemcpy((char *)dptr + offset, src, bytes, kind);
}

// 1: this library launches its kernels from code objects through hipModuleLaunchKernel (0 in the default fat-binary build, capi_stage.cpp)
int dabx_internal_hipmodule(void) { return 1; }

}  // extern "C"

// lazy_lib.h (host mirror) -- the libraries that bring a large ROCm dependency into the process (libflex_vendor.so: hipSPARSE +
// rocSPARSE, 490 MB; libflex_mg.so: RCCL, 570 MB; libflex_axw.so: rocBLAS) are loaded at their first use, not with the binary.
// A plain run never needed them resident; under `--counters` it matters: with a profiler attached the runtime loads the code
// objects of every library in the process eagerly, and on a freshly provisioned box that page-in took minutes.
#pragma once
#include <dlfcn.h>

#include <stdexcept>
#include <string>

inline void *lazy_lib(const char *file) {
    void *h = dlopen(file, RTLD_NOW | RTLD_LOCAL);  // next to the binary (RUNPATH $ORIGIN); dlopen counts references itself
    if (!h) throw std::runtime_error(std::string(file) + ": " + dlerror());
    return h;
}

template <class F>
F *lazy_fn(const char *file, const char *symbol) {
    void *p = dlsym(lazy_lib(file), symbol);
    if (!p) throw std::runtime_error(std::string(file) + " lacks " + symbol);
    return reinterpret_cast<F *>(p);
}

// FLEX_VENDOR(flex_vendor_spmm_run)(h, stream): the declaration in include/*.h gives the type, the library gives the address
#define FLEX_VENDOR(fn) lazy_fn<decltype(fn)>("libflex_vendor.so", #fn)
#define FLEX_MG(fn) lazy_fn<decltype(fn)>("libflex_mg.so", #fn)
#define FLEX_AXW(fn) lazy_fn<decltype(fn)>("libflex_axw.so", #fn)

// Host cost of DeviceAllocate / DeviceFree pairs served from libmem's block cache, from C (no ctypes in the way):
// with and without libalgorithm.so's deferral hooks registered, with 0 / 1 / 2 query streams registered.
// build: g++ -O2 -std=c++17 -Iinclude -o /tmp/ubench_libmem tools/ubench_libmem.cpp -ldl
// usage: ubench_libmem <dir with libmem.so, libalgorithm.so> [reps]
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "ares_memory.h"

template <typename F>
static F sym(void *h, const char *name) {
  void *p = dlsym(h, name);
  if (!p) {
    fprintf(stderr, "missing %s\n", name);
    exit(2);
  }
  return reinterpret_cast<F>(p);
}

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : "aresdb_amd/lib";
  const int reps = argc > 2 ? atoi(argv[2]) : 3000;
  void *mem = dlopen((dir + "/libmem.so").c_str(), RTLD_NOW | RTLD_GLOBAL);
  if (!mem) {
    fprintf(stderr, "%s\n", dlerror());
    return 2;
  }
  auto allocate = sym<CGoCallResHandle (*)(size_t, int)>(mem, "DeviceAllocate");
  auto release = sym<CGoCallResHandle (*)(void *, int)>(mem, "DeviceFree");
  auto create = sym<CGoCallResHandle (*)(int)>(mem, "CreateCudaStream");
  using clk = std::chrono::steady_clock;
  auto round = [&](const char *label) {
    const size_t bytes = 8u << 20;
    std::vector<void *> warm;
    for (int i = 0; i < 4; i++) warm.push_back(allocate(bytes, 0).res);
    for (void *p : warm) release(p, 0);
    double ta = 0, tf = 0;
    for (int i = 0; i < reps; i++) {
      auto t0 = clk::now();
      void *p = allocate(bytes, 0).res;
      auto t1 = clk::now();
      release(p, 0);
      auto t2 = clk::now();
      ta += std::chrono::duration<double, std::micro>(t1 - t0).count();
      tf += std::chrono::duration<double, std::micro>(t2 - t1).count();
    }
    printf("%-58s DeviceAllocate %6.2f us  DeviceFree %6.2f us\n", label, ta / reps, tf / reps);
  };
  round("libmem alone, no stream registered");
  create(0);
  round("libmem alone, one stream");
  create(0);
  round("libmem alone, two streams");
  void *algo = dlopen((dir + "/libalgorithm.so").c_str(), RTLD_NOW | RTLD_GLOBAL);
  if (!algo) {
    fprintf(stderr, "%s\n", dlerror());
    return 2;
  }
  sym<CGoCallResHandle (*)()>(algo, "BootstrapDevice")();
  round("+ libalgorithm.so's hooks registered, two streams");
  return 0;
}

// Issue cost of the integer instructions murmur3 is made of, on gfx950: 16 wavefronts per CU (4 per SIMD, the
// scan kernels' occupancy), each running a chain of N dependent instructions of one kind on 8 independent
// registers.  Prints cycles per wave-instruction per SIMD (4 waves share a SIMD: busy cycles / (4 * N * 8)).
//   hipcc --offload-arch=gfx950 -O2 -o bin/ubench_valu ubench_valu.hip && bin/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 16384

template <int KIND>
__global__ __launch_bounds__(1024) void chain(unsigned *out, unsigned long long *cycles, unsigned seed) {
  unsigned r[8];
  for (int i = 0; i < 8; i++) r[i] = seed * (threadIdx.x + 1) + i;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int k = 0; k < REP; k++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (KIND == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
      if (KIND == 1) { unsigned long long t; asm volatile("v_mad_u64_u32 %0, vcc, %1, 5, %2" : "=v"(t) : "v"(r[i]), "v"((unsigned long long)seed) : "vcc"); r[i] = (unsigned)t; }
      if (KIND == 2) asm volatile("v_lshl_add_u32 %0, %0, 2, %0" : "+v"(r[i]));
      if (KIND == 3) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
      if (KIND == 4) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
      if (KIND == 5) asm volatile("v_alignbit_b32 %0, %0, %0, 17" : "+v"(r[i]));
      if (KIND == 6) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
      if (KIND == 7) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(r[i]) : "v"(seed));
      if (KIND == 8) asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(seed));
      if (KIND == 9) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "s"(seed));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  unsigned s = 0;
  for (int i = 0; i < 8; i++) s ^= r[i];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char *name, unsigned *out, unsigned long long *cyc) {
  const int grid = 256;
  chain<KIND><<<grid, 1024>>>(out, cyc, 12345u);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  chain<KIND><<<grid, 1024>>>(out, cyc, 12345u);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto c : h) avg += c;
  avg /= grid;
  // readcyclecounter ticks at a fixed 100 MHz on this family; derive shader cycles from the event time at 2.4 GHz too
  printf("%-28s %8.3f ms   counter ticks %10.0f   per wave-instruction per SIMD: %.2f cycles @2.4 GHz\n", name, ms, avg,
         ms * 1e-3 * 2.4e9 / (4.0 * REP * 8));
}

int main() {
  unsigned *out; unsigned long long *cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
  run<0>("v_mul_lo_u32 (vgpr)", out, cyc);
  run<9>("v_mul_lo_u32 (sgpr const)", out, cyc);
  run<1>("v_mad_u64_u32", out, cyc);
  run<4>("v_mul_hi_u32", out, cyc);
  run<3>("v_mul_u32_u24", out, cyc);
  run<7>("v_mad_u32_u24", out, cyc);
  run<2>("v_lshl_add_u32", out, cyc);
  run<5>("v_alignbit_b32", out, cyc);
  run<6>("v_xor_b32", out, cyc);
  run<8>("v_add3_u32", out, cyc);
  return 0;
}

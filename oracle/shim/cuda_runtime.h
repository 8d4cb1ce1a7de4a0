/* Build shim (test infrastructure only): lets the UNMODIFIED reference sources
 * under /root/reference compile with hipcc in QUERY_MODE=HOST (no RUN_ON_DEVICE).
 * The reference only uses the handful of CUDA runtime names below in HOST mode
 * (query/utils.cu:44-59 CheckCUDAError, stream typedefs in signatures). */
#ifndef ORACLE_SHIM_CUDA_RUNTIME_H_
#define ORACLE_SHIM_CUDA_RUNTIME_H_
#include <hip/hip_runtime.h>
typedef hipStream_t cudaStream_t;
typedef hipError_t cudaError_t;
#define cudaSuccess hipSuccess
/* HOST mode never touches a device; report success so the harmless
 * "no ROCm-capable device" print of utils.cu:55-57 stays quiet on CPU boxes. */
#define cudaGetLastError() (hipSuccess)
#define cudaGetErrorString hipGetErrorString
#endif

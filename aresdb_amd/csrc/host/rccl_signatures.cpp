// Build-time check of the RCCL entry points ares_driver.cpp binds with dlsym under hand-declared signatures
// (AresCommCreateRccl, rccl_all_gather, rccl_all_to_all): where <rccl/rccl.h> is installed, every declared signature is
// compared with the header's — argument by argument, with the substitutions the binding makes (int for the 4-byte
// result / datatype enums, void * for the opaque communicator and stream handles, a 128-byte struct for ncclUniqueId).
// No code is generated; without the header the file is empty.
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>) && __has_include(<hip/hip_runtime_api.h>)
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <rccl/rccl.h>

#include <cstddef>
#include <type_traits>

namespace {
template <typename A, typename B>
constexpr bool abi_same() {  // same size and same register class: how the dlsym'd pointer is called
  return sizeof(A) == sizeof(B) && std::is_pointer<A>::value == std::is_pointer<B>::value &&
         (std::is_integral<A>::value || std::is_enum<A>::value) == (std::is_integral<B>::value || std::is_enum<B>::value);
}
template <typename F>
struct Sig;
template <typename R, typename... A>
struct Sig<R (*)(A...)> {
  template <typename R2, typename... B>
  static constexpr bool matches(R2 (*)(B...)) {
    if constexpr (sizeof...(A) != sizeof...(B)) return false;
    else return abi_same<R, R2>() && (abi_same<A, B>() && ...);
  }
};
template <typename Declared, typename Real>
constexpr bool same_abi(Real real) { return Sig<Declared>::matches(real); }

struct NcclUniqueId { char internal[128]; };
static_assert(sizeof(NcclUniqueId) == sizeof(ncclUniqueId), "ncclUniqueId is passed by value");
static_assert(static_cast<int>(ncclUint8) == 1, "rccl_all_gather passes 1 for ncclUint8");
static_assert(static_cast<int>(ncclSuccess) == 0, "a result of 0 is success");

static_assert(same_abi<int (*)(const void *, void *, size_t, int, void *, void *)>(&ncclAllGather), "ncclAllGather");
static_assert(same_abi<int (*)(const void *, size_t, int, int, void *, void *)>(&ncclSend), "ncclSend");
static_assert(same_abi<int (*)(void *, size_t, int, int, void *, void *)>(&ncclRecv), "ncclRecv");
static_assert(same_abi<int (*)()>(&ncclGroupStart), "ncclGroupStart");
static_assert(same_abi<int (*)()>(&ncclGroupEnd), "ncclGroupEnd");
static_assert(same_abi<int (*)(void *)>(&ncclCommDestroy), "ncclCommDestroy");
static_assert(same_abi<const char *(*)(int)>(&ncclGetErrorString), "ncclGetErrorString");
static_assert(same_abi<int (*)(void *)>(&ncclGetUniqueId), "ncclGetUniqueId");  // (pointer to the 128-byte id)
// ncclCommInitRank(ncclComm_t *, int, ncclUniqueId by value, int)
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t *, int, ncclUniqueId, int)>::value, "ncclCommInitRank");
}  // namespace
#endif
#endif

// CPU test TU (tests/test_abi.py compiles it with -fsyntax-only): the hand-declared RCCL ABI slice of
// mfrec_amd/csrc/mfx_rccl_abi.h against the real <rccl/rccl.h> of this ROCm.  Nothing here is linked or run.
#include <rccl/rccl.h>
#include <type_traits>
#include "../mfrec_amd/csrc/mfx_rccl_abi.h"

namespace {

// two parameter (or return) types are call-compatible on x86-64 / SysV when both are pointers, or both integral / enum of one
// size, or both trivially copyable aggregates of one size and alignment (the by-value 128-byte id travels in memory)
template <class A, class B>
constexpr bool abi_same()
{
    if constexpr (std::is_pointer<A>::value || std::is_pointer<B>::value)
        return std::is_pointer<A>::value && std::is_pointer<B>::value;
    else if constexpr (std::is_class<A>::value || std::is_class<B>::value)
        return std::is_class<A>::value && std::is_class<B>::value && sizeof(A) == sizeof(B) && alignof(A) == alignof(B) &&
               std::is_trivially_copyable<A>::value && std::is_trivially_copyable<B>::value;
    else
        return (std::is_integral<A>::value || std::is_enum<A>::value) && (std::is_integral<B>::value || std::is_enum<B>::value) &&
               sizeof(A) == sizeof(B);
}

template <class F, class G> struct Compat : std::false_type {};
template <class R1, class... A, class R2, class... B>
struct Compat<R1 (*)(A...), R2 (*)(B...)> {
    static constexpr bool args()
    {
        if constexpr (sizeof...(A) != sizeof...(B)) return false;
        else return (abi_same<A, B>() && ... && true);
    }
    static constexpr bool value = abi_same<R1, R2>() && args();
};

#define MFX_CHECK(name, type) static_assert(Compat<decltype(&name), mfx_rccl::type>::value, #name " no longer matches mfx_rccl::" #type)
MFX_CHECK(ncclGetUniqueId, GetUniqueId_t);
MFX_CHECK(ncclCommInitRank, CommInitRank_t);
MFX_CHECK(ncclCommDestroy, CommDestroy_t);
MFX_CHECK(ncclAllReduce, AllReduce_t);
MFX_CHECK(ncclAllGather, AllGather_t);
MFX_CHECK(ncclBroadcast, Broadcast_t);
MFX_CHECK(ncclSend, Send_t);
MFX_CHECK(ncclRecv, Recv_t);
MFX_CHECK(ncclGroupStart, GroupStart_t);
MFX_CHECK(ncclGroupEnd, GroupEnd_t);
MFX_CHECK(ncclGetErrorString, GetErrorString_t);
MFX_CHECK(ncclGetLastError, GetLastError_t);

static_assert((int)::ncclSuccess == (int)mfx_rccl::ncclSuccess, "ncclSuccess");
static_assert((int)::ncclFloat64 == (int)mfx_rccl::ncclFloat64, "ncclFloat64");
static_assert((int)::ncclSum == (int)mfx_rccl::ncclSum, "ncclSum");
static_assert((int)::ncclMax == (int)mfx_rccl::ncclMax, "ncclMax");
static_assert(sizeof(::ncclUniqueId) == sizeof(mfx_rccl::ncclUniqueId) && sizeof(::ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
static_assert(sizeof(::ncclComm_t) == sizeof(mfx_rccl::ncclComm_t), "ncclComm_t is a pointer");

}  // namespace

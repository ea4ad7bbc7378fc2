// The slice of the RCCL (NCCL-compatible) C ABI that mfx_comm.hip binds with dlsym: twelve entry points, one data type,
// two reduction operators, the 128-byte unique id.  Declared by hand so that libmfx.so neither links librccl nor needs its
// headers at build time; tests/rccl_abi_check.cpp includes <rccl/rccl.h> beside this file and static_asserts that every
// pointer type below is call-compatible with the real declaration and that the enum values agree (tests/test_abi.py).
#pragma once
#include <stddef.h>
#include <hip/hip_runtime_api.h>

namespace mfx_rccl {

typedef struct { char internal[128]; } ncclUniqueId;
typedef void *ncclComm_t;
enum { ncclSuccess = 0 };
enum { ncclFloat64 = 8 };
enum { ncclSum = 0, ncclMax = 2 };

typedef int (*GetUniqueId_t)(ncclUniqueId *);
typedef int (*CommInitRank_t)(ncclComm_t *, int, ncclUniqueId, int);
typedef int (*CommDestroy_t)(ncclComm_t);
typedef int (*AllReduce_t)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*AllGather_t)(const void *, void *, size_t, int, ncclComm_t, hipStream_t);
typedef int (*Broadcast_t)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*Send_t)(const void *, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*Recv_t)(void *, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*GroupStart_t)();
typedef int (*GroupEnd_t)();
typedef const char *(*GetErrorString_t)(int);
typedef const char *(*GetLastError_t)(ncclComm_t);

}  // namespace mfx_rccl

// tests/host/emu/rccl/rccl.h -- TEST INFRASTRUCTURE: the handful of RCCL declarations simdjson_amd/csrc/sjgpu_comm.hip uses, for
// the CPU tier's build of that file against tests/host/emu (no ROCm headers on that include path: <hip/hip_runtime.h> is the
// emulator's).  Names, argument order and enumerator values are the published NCCL API's; the library behind them in that build is
// tests/stubs/rccl_loopback.cpp (ranks = threads of one process), opened through SJGPU_RCCL_LIB like the real one.
#ifndef SJ_EMU_RCCL_H
#define SJ_EMU_RCCL_H

#include <hip/hip_runtime.h>

#include <cstddef>

extern "C" {

#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct ncclComm *ncclComm_t;

typedef enum {
  ncclSuccess = 0,
  ncclUnhandledCudaError = 1,
  ncclSystemError = 2,
  ncclInternalError = 3,
  ncclInvalidArgument = 4,
  ncclInvalidUsage = 5,
  ncclRemoteError = 6,
  ncclInProgress = 7,
  ncclNumResults = 8
} ncclResult_t;

typedef enum {
  ncclInt8 = 0, ncclChar = 0,
  ncclUint8 = 1,
  ncclInt32 = 2, ncclInt = 2,
  ncclUint32 = 3,
  ncclInt64 = 4,
  ncclUint64 = 5
} ncclDataType_t;

ncclResult_t ncclGetUniqueId(ncclUniqueId *uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count);
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
const char *ncclGetErrorString(ncclResult_t result);

} // extern "C"
#endif

// The host-supplied all-reduce of INTEGRATION.md section 3 as a file: an MPI build of PIPS-IPM++ that prefers its own
// (GPU-aware) MPI over RCCL hands this callback to pips_hip_comm_create_external; the library synchronises its stream, calls
// back with the device buffer and continues.  Syntax-checked by tests/test_adapter_compiles.py where an mpi.h exists.
#ifndef PIPS_MPI_ALLREDUCE_CALLBACK_H
#define PIPS_MPI_ALLREDUCE_CALLBACK_H
#include <mpi.h>
#include <cstddef>
#include "pips_hip.h"

// = PIPS_MPIsumArrayInPlace (Utilities/pipsdef.h) on a device buffer
static int pips_mpi_sum_in_place(void* user, double* buf_dev, size_t n) {
   return MPI_Allreduce(MPI_IN_PLACE, buf_dev, static_cast<int>(n), MPI_DOUBLE, MPI_SUM, *static_cast<MPI_Comm*>(user)) != MPI_SUCCESS;
}

// the communicator handle the fused entry points take (pips_hip_kkt_create / pips_ipm_create_rank); mpi_comm must outlive it
inline void* pips_make_mpi_comm(MPI_Comm* mpi_comm) {
   void* comm = nullptr;
   return pips_hip_comm_create_external(&comm, pips_mpi_sum_in_place, mpi_comm) == 0 ? comm : nullptr;
}
#endif

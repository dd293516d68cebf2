// The exchange step of a multi-device matcher (include/ndt2d_hip.h,
// ndt2d_matcher_create_multi): the one collective of the sharded hot path
//  -- an all-reduce(sum) of a [n_dev, k] table of doubles in which
// every device fills its own row -- over RCCL's single-process communicators
// (ncclCommInitAll: one host thread, n devices, xGMI between them).
// This interface is HIP-free: ndt2d_host.cpp is compiled by the host compiler.
#ifndef NDT2D_EXCHANGE_H_
#define NDT2D_EXCHANGE_H_

#include <cstddef>
#include <string>

namespace ndt2d
{

struct Exchange;

// One communicator per device of device_ids[n] (all different: RCCL refuses a device
// twice).  librccl is loaded on the first call (dlopen: a single-device host never
// loads it).  Returns 0, or a NDT2D_ERR_* code with *err filled in.
int exchange_create(Exchange ** out, const int * device_ids, int n, std::string * err);
void exchange_destroy(Exchange * ex);

// In-place all-reduce(sum) of `count` doubles at d_buf[r] (memory of device r) on
// hip_streams[r], all n ranks fused in one ncclGroupStart / ncclGroupEnd.  Asynchronous.
int exchange_all_reduce(Exchange * ex, double * const * d_buf, size_t count,
                        void * const * hip_streams, std::string * err);

// d_out[c] = d_table[0][c] + d_table[1][c] + ... (row order: the fixed rank order the
// host-combine path uses, so both exchanges give the same bits).  Asynchronous.
int sum_rows_launch(int device, const double * d_table, int rows, int cols, double * d_out,
                    void * hip_stream, std::string * err);

}  // namespace ndt2d

#endif  // NDT2D_EXCHANGE_H_

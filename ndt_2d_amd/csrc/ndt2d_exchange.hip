// RCCL exchange of a multi-device matcher (ndt2d_exchange.h): single-process
// communicators (ncclCommInitAll), one in-place all-reduce(sum) of the record
// table per search / particle batch, fused across the devices in one group.
// librccl.so is loaded with dlopen when the first multi-device matcher with
// distinct devices is created -- libndt2d_hip.so itself does not link it, so a
// single-GPU host never pays for loading it (573 MB on ROCm 7.2).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "ndt2d_exchange.h"
#include "ndt2d_hip.h"

namespace ndt2d
{

namespace
{

struct RcclApi
{
  void * lib = nullptr;
  decltype(&ncclCommInitAll) comm_init_all = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  std::string load_error;
};

// Loaded once per process; never unloaded (communicators may outlive any one matcher).
RcclApi * rccl_api()
{
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, []() {
    const char * names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char * name : names)
    {
      api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (api.lib != nullptr) break;
    }
    if (api.lib == nullptr)
    {
      const char * e = dlerror();
      api.load_error = std::string("librccl.so.1 could not be loaded: ") + (e != nullptr ? e : "?");
      return;
    }
    api.comm_init_all = reinterpret_cast<decltype(api.comm_init_all)>(dlsym(api.lib, "ncclCommInitAll"));
    api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(dlsym(api.lib, "ncclCommDestroy"));
    api.all_reduce = reinterpret_cast<decltype(api.all_reduce)>(dlsym(api.lib, "ncclAllReduce"));
    api.group_start = reinterpret_cast<decltype(api.group_start)>(dlsym(api.lib, "ncclGroupStart"));
    api.group_end = reinterpret_cast<decltype(api.group_end)>(dlsym(api.lib, "ncclGroupEnd"));
    api.error_string = reinterpret_cast<decltype(api.error_string)>(dlsym(api.lib, "ncclGetErrorString"));
    if (api.comm_init_all == nullptr || api.comm_destroy == nullptr || api.all_reduce == nullptr ||
        api.group_start == nullptr || api.group_end == nullptr || api.error_string == nullptr)
    {
      api.load_error = "librccl.so.1 lacks a symbol of the collective API";
    }
  });
  return &api;
}

int fail(std::string * err, int code, const std::string & msg)
{
  if (err != nullptr) *err = msg;
  return code;
}

// rows summed in row order by one thread per column: a handful of additions
__global__ void sum_rows_kernel(const double * table, int rows, int cols, double * out)
{
  const int c = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
  if (c >= cols) return;
  double acc = table[c];
  for (int r = 1; r < rows; ++r) acc += table[static_cast<size_t>(r) * cols + c];
  out[c] = acc;
}

}  // namespace

struct Exchange
{
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;
};

int exchange_create(Exchange ** out, const int * device_ids, int n, std::string * err)
{
  if (out == nullptr || device_ids == nullptr || n <= 0) return fail(err, NDT2D_ERR_INVALID, "exchange_create: bad argument");
  *out = nullptr;
  for (int i = 0; i < n; ++i)
  {
    for (int j = 0; j < i; ++j)
    {
      if (device_ids[i] == device_ids[j])
      {
        return fail(err, NDT2D_ERR_INVALID, "exchange_create: RCCL takes every device once (use the host exchange)");
      }
    }
  }
  RcclApi * api = rccl_api();
  if (!api->load_error.empty()) return fail(err, NDT2D_ERR_HIP, api->load_error);
  Exchange * ex = new (std::nothrow) Exchange();
  if (ex == nullptr) return fail(err, NDT2D_ERR_INVALID, "exchange_create: out of memory");
  ex->devices.assign(device_ids, device_ids + n);
  ex->comms.assign(static_cast<size_t>(n), nullptr);
  // (the node's two matcher instances live on two threads: one communicator set is made at a time)
  static std::mutex init_mutex;
  std::lock_guard<std::mutex> lock(init_mutex);
  const ncclResult_t r = api->comm_init_all(ex->comms.data(), n, ex->devices.data());
  if (r != ncclSuccess)
  {
    delete ex;
    (void)hipGetLastError();
    return fail(err, NDT2D_ERR_HIP, std::string("ncclCommInitAll: ") + api->error_string(r));
  }
  *out = ex;
  return NDT2D_OK;
}

void exchange_destroy(Exchange * ex)
{
  if (ex == nullptr) return;
  RcclApi * api = rccl_api();
  for (size_t r = 0; r < ex->comms.size(); ++r)
  {
    if (ex->comms[r] != nullptr)
    {
      (void)hipSetDevice(ex->devices[r]);
      (void)api->comm_destroy(ex->comms[r]);
    }
  }
  delete ex;
}

int exchange_all_reduce(Exchange * ex, double * const * d_buf, size_t count, void * const * hip_streams,
                        std::string * err)
{
  if (ex == nullptr || d_buf == nullptr || hip_streams == nullptr || count == 0)
  {
    return fail(err, NDT2D_ERR_INVALID, "exchange_all_reduce: bad argument");
  }
  RcclApi * api = rccl_api();
  // One communicator set is enqueued on all its devices before the next one starts: the node
  // runs its local and its global matcher on two threads (src/ndt_mapper.cpp:141-142), and
  // two sets over the same GPUs whose collectives reach the devices in different orders can
  // deadlock (NCCL's rule for concurrent communicators).  Process-wide, held for the group only.
  static std::mutex group_mutex;
  std::lock_guard<std::mutex> group_lock(group_mutex);
  ncclResult_t r = api->group_start();
  if (r != ncclSuccess) return fail(err, NDT2D_ERR_HIP, std::string("ncclGroupStart: ") + api->error_string(r));
  ncclResult_t first_bad = ncclSuccess;
  for (size_t k = 0; k < ex->comms.size(); ++k)
  {
    (void)hipSetDevice(ex->devices[k]);
    r = api->all_reduce(d_buf[k], d_buf[k], count, ncclDouble, ncclSum, ex->comms[k],
                        static_cast<hipStream_t>(hip_streams[k]));
    if (r != ncclSuccess && first_bad == ncclSuccess) first_bad = r;
  }
  r = api->group_end();   // (always closed, whatever a rank said)
  if (first_bad != ncclSuccess) r = first_bad;
  if (r != ncclSuccess) return fail(err, NDT2D_ERR_HIP, std::string("ncclAllReduce: ") + api->error_string(r));
  return NDT2D_OK;
}

int sum_rows_launch(int device, const double * d_table, int rows, int cols, double * d_out, void * hip_stream,
                    std::string * err)
{
  if (d_table == nullptr || d_out == nullptr || rows <= 0 || cols <= 0)
  {
    return fail(err, NDT2D_ERR_INVALID, "sum_rows_launch: bad argument");
  }
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess)
  {
    sum_rows_kernel<<<dim3((cols + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(hip_stream)>>>(d_table, rows, cols,
                                                                                                 d_out);
    e = hipGetLastError();
  }
  if (e != hipSuccess) return fail(err, NDT2D_ERR_HIP, std::string("sum_rows_kernel: ") + hipGetErrorString(e));
  return NDT2D_OK;
}

}  // namespace ndt2d

// The C boundary lets no C++ exception through (include/ndt2d_hip.h: "int status, no exceptions";
// the plugin's caller is rclcpp's executor, reference src/ndt_mapper.cpp).  Every multi-line
// extern "C" entry point of ndt2d_host.cpp / ndt2d_device.hip has its body between these two
// macros: std::bad_alloc / std::length_error -- a std::vector sized by a pose of 1e15 or an
// infinite range_max that slipped past the argument checks -- become NDT2D_ERR_ALLOC, anything else
// NDT2D_ERR_INTERNAL, the message goes where ndt2d_last_error / ndt2d_matcher_last_error find it,
// and the handle stays usable.  guard_note(handle, text) is defined by the including file for its
// handle type and for nullptr; it must not throw.
#ifndef NDT2D_GUARD_H_
#define NDT2D_GUARD_H_

#include <exception>
#include <new>
#include <stdexcept>

#define NDT2D_C_TRY try {
#define NDT2D_C_CATCH(handle)                                              \
  }                                                                        \
  catch (const std::bad_alloc &)                                           \
  {                                                                        \
    guard_note(handle, "out of memory (std::bad_alloc)");                  \
    return NDT2D_ERR_ALLOC;                                                \
  }                                                                        \
  catch (const std::length_error &)                                        \
  {                                                                        \
    guard_note(handle, "a size passed the container limits (std::length_error)"); \
    return NDT2D_ERR_ALLOC;                                                \
  }                                                                        \
  catch (const std::exception & guarded_exception_)                        \
  {                                                                        \
    guard_note(handle, guarded_exception_.what());                         \
    return NDT2D_ERR_INTERNAL;                                             \
  }                                                                        \
  catch (...)                                                              \
  {                                                                        \
    guard_note(handle, "unknown exception");                               \
    return NDT2D_ERR_INTERNAL;                                             \
  }

#endif  // NDT2D_GUARD_H_

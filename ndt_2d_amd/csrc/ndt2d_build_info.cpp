// ndt2d_build_info(): which sources this library was compiled from.  build.py passes the
// sha256 of csrc/*.hip, *.cpp, the headers and the compiler flags (build.source_sha256()) as
// NDT2D_SOURCE_SHA; tests/conftest.py rebuilds a library whose hash is not that of the tree,
// tests/test_capi_symbols.py asserts the equality and bench.py prints it
// (`lib_matches_source`): the .so is git-ignored but travels to the GPU box, and nothing
// else ties it to the sources it is measured and tested as.
#include "ndt2d_hip.h"

#ifndef NDT2D_SOURCE_SHA
#error "compile through ndt_2d_amd/build.py (it passes -DNDT2D_SOURCE_SHA=\"<sha256 of the sources>\")"
#endif
#ifndef NDT2D_BUILD_ARCH
#define NDT2D_BUILD_ARCH "gfx950"
#endif

extern "C" const char * ndt2d_build_info(void)
{
  // (the marker is what build.embedded_sha256() looks for in the file)
  // (hooks=1: libndt2d_hip_hooks.so, the test build with -DNDT2D_TEST_HOOKS -- never the product)
#ifdef NDT2D_BUILD_HOOKS
  return "NDT2D_SOURCE_SHA256=" NDT2D_SOURCE_SHA " arch=" NDT2D_BUILD_ARCH " compiler=" __VERSION__ " hooks=1";
#else
  return "NDT2D_SOURCE_SHA256=" NDT2D_SOURCE_SHA " arch=" NDT2D_BUILD_ARCH " compiler=" __VERSION__ " hooks=0";
#endif
}

// clw_context.hpp -- drop-in for opencl_wrapper/include/clw_context.hpp over the clwh C ABI.
//
// Same surface as the reference class (default ctor, ctor taking a GL context pointer, non-copyable,
// non-movable); instead of an OpenCL context + profiling queue on platform[0]/device[0]
// (opencl_wrapper/src/clw_context.cpp:38-50) it owns a HIP device ordinal and one in-order stream.
// The device is taken from the environment variable CLW_DEVICE (default 0).
#pragma once

#include <cstdlib>

#include "clw_helper.hpp"
#include "clwh.h"

class clw_context {
 public:
  clw_context() { open(); }
  // the reference shares a GL context for CL-GL interop (never used by the app); accepted and ignored
  explicit clw_context(void * /*gl_context*/) { open(); }
  ~clw_context() {
    if (m_ctx) clw_fail_hard_on_error(clwh_ctx_destroy(m_ctx));
  }
  clw_context(const clw_context &) = delete;
  clw_context(clw_context &&) = delete;
  clw_context &operator=(const clw_context &) = delete;
  clw_context &operator=(clw_context &&) = delete;

  clwh_ctx *get_handle() const { return m_ctx; }
  void finish() const { clw_fail_hard_on_error(clwh_ctx_finish(m_ctx)); }

 private:
  void open() {
    const char *dev = std::getenv("CLW_DEVICE");
    clw_fail_hard_on_error(clwh_ctx_create(dev ? std::atoi(dev) : 0, &m_ctx));
  }
  clwh_ctx *m_ctx = nullptr;
};

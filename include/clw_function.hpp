// clw_function.hpp -- drop-in for opencl_wrapper/include/clw_function.hpp over the clwh C ABI.
//
// The reference constructor flattens `#clw_include_once`, prepends `prepend` and JIT-compiles the
// OpenCL source (clw_function.hpp:74-111).  Here (file, entry) selects a precompiled gfx950 kernel
// and `prepend` -- the generated `is_event_gen` source -- is parsed into a launch-time table, so
// constructing a clw_function on every transfer-function flush costs microseconds.
// execute() marshals arguments like the reference (clw_function.hpp:152-167): arithmetic arguments
// by value, everything else through get_device_reference().
#pragma once

#include <array>
#include <cassert>
#include <cstdint>
#include <iostream>
#include <string>
#include <type_traits>
#include <utility>

#include "clw_context.hpp"
#include "clw_helper.hpp"
#include "clw_vector.hpp"

class clw_function {
 public:
  clw_function(const clw_context &context, const std::string &path, const std::string &function_name,
               const std::string prepend = "")
      : m_function_name(function_name), m_path(path), m_prepend(prepend), m_context(&context) {
    clw_fail_hard_on_error(clwh_kernel_get(context.get_handle(), path.c_str(), function_name.c_str(), prepend.c_str(), &m_kernel));
  }
  ~clw_function() { release(); }
  clw_function(const clw_function &) = delete;
  clw_function(clw_function &&) = delete;
  clw_function &operator=(const clw_function &) = delete;
  clw_function &operator=(clw_function &&other) {
    assert(this != &other);
    release();
    m_context = other.m_context;
    m_kernel = other.m_kernel;
    other.m_context = nullptr;
    other.m_kernel = nullptr;
    m_path = std::move(other.m_path);
    m_function_name = std::move(other.m_function_name);
    m_prepend = std::move(other.m_prepend);
    return *this;
  }

  template <typename... Args>
  void execute(std::array<size_t, 3> global_size, std::array<size_t, 3> local_size, const Args &...arg) const {
    for (auto &v : global_size)
      if (v == 0) v = 1;
    for (auto &v : local_size)
      if (v == 0) v = 1;
    if (local_size[0] * local_size[1] * local_size[2] > 256) {
      std::cerr << "Warning, creating local size incompatible with AMD GPUs.\n"
                << "  used local size: " << local_size[0] * local_size[1] * local_size[2] << " > 256\n";
    }
    for (int k = 0; k < 3; ++k) {
      assert(global_size[k] >= local_size[k]);
      assert((global_size[k] % local_size[k]) == 0);
    }
    clwh_arg args[sizeof...(Args) > 0 ? sizeof...(Args) : 1];
    int n = 0;
    (void)std::initializer_list<int>{(marshal(args[n++], arg), 0)...};
    clw_fail_hard_on_error(clwh_launch(m_kernel, global_size.data(), local_size.data(), args, (int)sizeof...(Args)));
  }

  // not in the reference: the kernel handle, for the C ABI's entry points beyond the generic launch (clwh_render)
  clwh_kernel *get_kernel() const { return m_kernel; }

 private:
  template <typename Arg>
  static void marshal(clwh_arg &out, const Arg &a) {
    out.reserved = 0;
    if constexpr (std::is_floating_point<Arg>::value) {
      if (sizeof(Arg) == 4) { out.kind = CLWH_ARG_F32; out.v.f32 = (float)a; }
      else { out.kind = CLWH_ARG_F64; out.v.f64 = (double)a; }
    } else if constexpr (std::is_integral<Arg>::value) {
      if (sizeof(Arg) <= 4) {
        if (std::is_signed<Arg>::value) { out.kind = CLWH_ARG_I32; out.v.i32 = (int32_t)a; }
        else { out.kind = CLWH_ARG_U32; out.v.u32 = (uint32_t)a; }
      } else {
        if (std::is_signed<Arg>::value) { out.kind = CLWH_ARG_I64; out.v.i64 = (int64_t)a; }
        else { out.kind = CLWH_ARG_U64; out.v.u64 = (uint64_t)a; }
      }
    } else {
      out.kind = CLWH_ARG_MEM;
      out.v.mem = a.get_device_reference();
    }
  }
  void release() {
    if (m_kernel) clw_fail_hard_on_error(clwh_kernel_release(m_kernel));
    m_kernel = nullptr;
  }
  clwh_kernel *m_kernel = nullptr;
  std::string m_function_name, m_path, m_prepend;
  const clw_context *m_context;
};

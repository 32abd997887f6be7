// clw_foreign_memory.hpp -- drop-in for opencl_wrapper/include/clw_foreign_memory.hpp over the clwh C ABI:
// device memory that belongs to somebody else (the display), used as the frame `render` writes, so that the
// frame never takes the blocking readback of renderer::render_frame (app/renderer.cpp:150) and the re-upload of
// ui::run (app/ui.cpp:298-308).
//
// The reference makes a cl_mem from a GL texture (clCreateFromGLTexture, :13-19) and brackets its use with
// clEnqueueAcquireGLObjects / clEnqueueReleaseGLObjects (:40-46).  HIP's graphics interop maps a registered GL
// buffer to a plain device pointer (hipGraphicsGLRegisterBuffer, hipGraphicsMapResources,
// hipGraphicsResourceGetMappedPointer); that mapping needs a GL context and is the caller's three lines -- this
// class starts from the mapped pointer, which keeps it usable for any other owner of device memory as well
// (another framework's allocator, a compositor's shared buffer).  Same surface as the reference class:
// non-copyable, get_device_reference(), acquire(), release(); passed to clw_function::execute like a clw_image.
#pragma once

#include <array>
#include <cstddef>

#include "clw_context.hpp"
#include "clw_helper.hpp"

class clw_foreign_memory {
 public:
  // `device_ptr`: width x height RGBA8 pixels, row-major, owned by the caller and alive as long as this object;
  // `foreign_stream`: the hipStream_t the owner reads it on (nullptr = the default stream)
  clw_foreign_memory(const clw_context &context, void *device_ptr, size_t width, size_t height, void *foreign_stream = nullptr)
      : m_context(context), m_stream(foreign_stream) {
    const std::array<size_t, 3> dims{width, height, 1};
    clw_fail_hard_on_error(clwh_image_wrap(context.get_handle(), device_ptr, dims.data(), 4, CLWH_ELEM_U8, &m_foreign_array));
  }
  ~clw_foreign_memory() {
    if (m_foreign_array) clw_fail_hard_on_error(clwh_mem_release(m_foreign_array));  // the wrap only; the memory stays the owner's
  }
  clw_foreign_memory(const clw_foreign_memory &) = delete;
  clw_foreign_memory(clw_foreign_memory &&) = delete;
  clw_foreign_memory &operator=(const clw_foreign_memory &) = delete;
  clw_foreign_memory &operator=(clw_foreign_memory &&) = delete;

  clwh_mem *const &get_device_reference() const { return m_foreign_array; }

  // the owner has finished reading the previous frame: kernels launched after this may overwrite it
  void acquire() const { clw_fail_hard_on_error(clwh_ctx_acquire_from(m_context.get_handle(), m_stream)); }
  // the frame is complete for whatever the owner queues on its stream after this; no host synchronisation
  void release() const { clw_fail_hard_on_error(clwh_ctx_release_to(m_context.get_handle(), m_stream)); }

 private:
  clwh_mem *m_foreign_array = nullptr;
  const clw_context &m_context;
  void *m_stream;
};

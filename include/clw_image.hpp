// clw_image.hpp -- drop-in for opencl_wrapper/include/clw_image.hpp over the clwh C ABI:
// a host std::vector mirrored by a 1-D / 2-D / 3-D device image, blocking push()/pull().
//
// Element kind follows (signedness, sizeof) of TDevice and the channel count must be 1, 2 or 4, as in
// the reference (clw_image.hpp:68-110); dims of 0 count as 1; a host vector of the wrong size is a
// fatal error (clw_image.hpp:37-42).  The device layout is the shim's business: images are stored
// linearly (x fastest) and the render kernels keep their own bricked copy.
#pragma once

#include <array>
#include <cassert>
#include <cstddef>
#include <cstdlib>
#include <iostream>
#include <type_traits>
#include <utility>
#include <vector>

#include "clw_context.hpp"
#include "clw_helper.hpp"

template <typename TDevice, size_t ChannelSize = 1>
class clw_image {
  using TInternal = typename std::remove_const<TDevice>::type;
  static_assert(ChannelSize == 1 || ChannelSize == 2 || ChannelSize == 4, "Invalid channel size.");
  static_assert(sizeof(TInternal) <= 4, "Error, only 32 bit or smaller values are supported.");

 public:
  clw_image(const clw_context &context, std::vector<TInternal> &&data, std::array<size_t, 3> dimensions,
            const bool push_on_construction = false)
      : m_host(std::move(data)), m_context(&context) {
    for (auto &d : dimensions)
      if (d == 0) d = 1;
    m_dimensions = dimensions;
    const size_t expected = dimensions[0] * dimensions[1] * dimensions[2] * ChannelSize;
    if (expected != m_host.size()) {
      std::cerr << "Error, moved array is not of the correct size\n"
                << " Expected: " << expected << '\n'
                << " Received: " << m_host.size() << '\n';
      std::exit(1);
    }
    if (!(dimensions[0] > 1)) {
      std::cerr << "Error, image size is not valid \n";
      std::exit(1);
    }
    const int flags = std::is_const<TDevice>::value ? CLWH_MEM_READ_ONLY : CLWH_MEM_READ_WRITE;
    clw_fail_hard_on_error(clwh_image_create(m_context->get_handle(), m_dimensions.data(), (int)ChannelSize,
                                             element_kind(), flags, &m_mem));
    pin_host();
    if (push_on_construction) push();
  }
  // copying re-allocates a device image with the same host contents, like the reference
  clw_image(const clw_image &other, const bool push_on_construction = false)
      : clw_image(*other.m_context, std::vector<TInternal>(other.m_host), other.m_dimensions, push_on_construction) {}
  clw_image(clw_image &&) = delete;
  clw_image &operator=(clw_image &) = delete;
  clw_image &operator=(clw_image &&other) {
    assert(this != &other);
    release();
    m_context = other.m_context;
    m_dimensions = other.m_dimensions;
    m_mem = other.m_mem;
    m_host = std::move(other.m_host);  // the buffer (and its page lock) changes owner, not address
    m_pinned = other.m_pinned;
    other.m_pinned = false;
    other.m_mem = nullptr;
    other.m_context = nullptr;
    return *this;
  }
  ~clw_image() { release(); }

  TInternal &operator[](std::size_t index) { return m_host[index]; }
  const TInternal &operator[](std::size_t index) const { return m_host[index]; }

  void push() const {
    clw_fail_hard_on_error(clwh_mem_push(m_context->get_handle(), m_mem, m_host.data(), m_host.size() * sizeof(TInternal)));
  }
  void pull() {
    clw_fail_hard_on_error(clwh_mem_pull(m_context->get_handle(), m_mem, m_host.data(), m_host.size() * sizeof(TInternal)));
  }
  clwh_mem *const &get_device_reference() const { return m_mem; }
  size_t size() const { return m_host.size(); }
  size_t pixel_count() const { return m_host.size() / ChannelSize; }
  size_t channels() const { return ChannelSize; }
  const std::array<size_t, 3> &get_dimensions() const { return m_dimensions; }

 private:
  static constexpr int element_kind() {
    if (!std::is_integral<TInternal>::value) return CLWH_ELEM_F32;
    if (std::is_signed<TInternal>::value)
      return sizeof(TInternal) == 1 ? CLWH_ELEM_S8 : (sizeof(TInternal) == 2 ? CLWH_ELEM_S16 : CLWH_ELEM_S32);
    return sizeof(TInternal) == 1 ? CLWH_ELEM_U8 : (sizeof(TInternal) == 2 ? CLWH_ELEM_U16 : CLWH_ELEM_U32);
  }
  // Not in the reference: the host mirror of a frame-sized image (what every render_frame pulls; environment maps) is
  // page-locked so that push() / pull() are single DMAs.  Volumes (pushed once, up to 16 GiB) are not worth locking.
  // A failure to lock is not an error: transfers just stay staged.
  void pin_host() {
    const size_t bytes = m_host.size() * sizeof(TInternal);
    if (bytes >= ((size_t)1 << 20) && bytes <= ((size_t)64 << 20))
      m_pinned = clwh_host_register(m_host.data(), m_host.size() * sizeof(TInternal)) == CLWH_OK;
  }
  void release() {
    if (m_pinned && !m_host.empty()) (void)clwh_host_unregister(m_host.data());
    m_pinned = false;
    if (m_mem) clw_fail_hard_on_error(clwh_mem_release(m_mem));
    m_mem = nullptr;
  }
  bool m_pinned = false;
  clwh_mem *m_mem = nullptr;
  std::vector<TInternal> m_host;
  const clw_context *m_context;  // not owned
  std::array<size_t, 3> m_dimensions{1, 1, 1};
};

// clw_vector.hpp -- drop-in for opencl_wrapper/include/clw_vector.hpp over the clwh C ABI:
// a host std::vector mirrored by a device buffer, blocking push()/pull().
#pragma once

#include <cassert>
#include <cstddef>
#include <type_traits>
#include <utility>
#include <vector>

#include "clw_context.hpp"
#include "clw_helper.hpp"

template <typename TDevice>
class clw_vector {
  using TInternal = typename std::remove_const<TDevice>::type;

 public:
  clw_vector(const clw_context &context, std::vector<TInternal> &&data, const bool push_on_construction = false)
      : m_host(std::move(data)), m_context(&context) {
    allocate();
    if (push_on_construction) push();
  }
  // copying re-allocates a device buffer of the same size (contents are NOT pushed), like the reference
  clw_vector(const clw_vector &other, const bool push_on_construction = false)
      : clw_vector(*other.m_context, std::vector<TInternal>(other.m_host), push_on_construction) {}
  clw_vector(clw_vector &&) = delete;
  clw_vector &operator=(const clw_vector &) = delete;
  clw_vector &operator=(clw_vector &&other) {
    assert(this != &other);
    release();
    m_context = other.m_context;
    m_mem = other.m_mem;
    m_host = std::move(other.m_host);
    other.m_mem = nullptr;
    other.m_context = nullptr;
    return *this;
  }
  ~clw_vector() { release(); }

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

 private:
  void allocate() {
    const int flags = std::is_const<TDevice>::value ? CLWH_MEM_READ_ONLY : CLWH_MEM_READ_WRITE;
    clw_fail_hard_on_error(clwh_mem_create(m_context->get_handle(), m_host.size() * sizeof(TInternal), flags, &m_mem));
  }
  void release() {
    if (m_mem) clw_fail_hard_on_error(clwh_mem_release(m_mem));
    m_mem = nullptr;
  }
  clwh_mem *m_mem = nullptr;
  std::vector<TInternal> m_host;
  const clw_context *m_context;  // not owned
};

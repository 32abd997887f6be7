// clw_helper.hpp -- fail-hard error convention of the clw_* wrappers, over the clwh C ABI.
//
// The reference wraps every OpenCL call in clw_fail_hard_on_error(status): on failure it prints the
// call site and the error name to stderr and exits with status 1
// (opencl_wrapper/include/clw_helper.hpp:293-309).  The HIP shim's C ABI only returns status codes
// (include/clwh.h); this header restores the reference's convention for C++ callers.
#pragma once

#include <cstdlib>
#include <iostream>

#include "clwh.h"

inline void clw_report_failure_and_exit(int status, const char *file, const char *function, int line) {
  if (status == CLWH_OK) return;
  std::cerr << "WARNING HIP shim call failed. See error message:\n"
            << "Failed Call Info: \n---------------------------------------\n\033[1;31m"
            << "File      : " << file << '\n'
            << "Function  : " << function << '\n'
            << "Line      : " << line << '\n'
            << "Error Msg : " << clwh_strerror(status) << " (hip error " << clwh_last_hip_error() << ")\n"
            << "\033[0m---------------------------------------\nExiting application... \n";
  std::exit(1);
}

#define clw_fail_hard_on_error(status) clw_report_failure_and_exit((status), __FILE__, __func__, __LINE__)

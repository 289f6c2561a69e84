// vdb_probe_env.hpp — the ONE place an environment variable may be read from.
//
// Environment switches (A / B probes of kernels and schedules, the loop-back collective transport of the tests) exist only in the
// PROBE build of the library: libvelesdb_hip_probe.so, the same sources compiled with -DVDB_PROBE_SWITCHES by velesdb_amd/build.py,
// loaded by tests/test_gpu_switches.py, the stub-transport tests and tools/probes — never by the package itself.  In the shipped
// libvelesdb_hip.so probe_env() is the constant nullptr: every switch folds to its default at compile time, the library imports no
// getenv and holds no switch name (tests/test_abi_exports.py::test_shipped_library_reads_no_environment_variable).
#pragma once
#include <cstdlib>

namespace vdb {
static inline const char* probe_env(const char* name) {
#ifdef VDB_PROBE_SWITCHES
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
}  // namespace vdb

// tests/plugin/activate_mi355x.cpp -- linked into the REFERENCE's own test programs (compiled from the sources
// where they lie under /root/reference/tests, never copied) so that they run with the MI355X backend as the
// active implementation.  A static initialiser performs the documented manual selection
// (doc/implementation-selection.md) before main(); the test sources themselves are untouched.
#include "mi355x_implementation.h"

#include <cstdio>
#include <cstdlib>

namespace {
struct activator {
  activator() {
    if (simdjson::mi355x::activate() != simdjson::SUCCESS) {
      std::fprintf(stderr, "mi355x backend unavailable (no HIP device): refusing to run on a CPU kernel\n");
      std::exit(3);
    }
    std::fprintf(stderr, "[active implementation: %s]\n", std::string(simdjson::get_active_implementation()->name()).c_str());
  }
} activate_before_main;
} // namespace

// The host side of the C++ mirror (field arithmetic incl. the inline-assembly products, evaluating and stored linear
// combinations, ProvingAssignment, the structure capture) compiled with AddressSanitizer + UndefinedBehaviorSanitizer and
// run over the three fixture circuits (chain, every form, seeded random structure); built and run by
// tests/test_round3_cpu.py::test_mirror_under_sanitizers.  No device involved: the rest of the C ABI is taken from
// libbellman_hip.so and never called.
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include "../../include/bellman_hip_test.h"
int main() {
  uint64_t wit[4] = {0x1234567890ABCDEFULL, 0, 0, 0};
  for (int kind = 1; kind <= 3; kind++) for (size_t rounds : {1, 7, 300, 2000}) {
    size_t c3[3];
    if (bh_test_demo_assignment(kind, rounds, 5 + rounds, wit, nullptr, c3, 0, 0, 0, 0, 0, 0, 0, 0)) { printf("rc\n"); return 1; }
    std::vector<uint64_t> a(c3[0] * 4), b(c3[0] * 4), c(c3[0] * 4), in(c3[1] * 4), aux(c3[2] * 4), d1(c3[2] / 64 + 2), d2(c3[1] / 64 + 2), d3(c3[2] / 64 + 2);
    if (bh_test_demo_assignment(kind, rounds, 5 + rounds, wit, nullptr, c3, a.data(), b.data(), c.data(), in.data(), aux.data(), d1.data(), d2.data(), d3.data())) return 1;
    size_t o4[4];
    double ms = bh_test_capture_check(kind, rounds, 5 + rounds, o4);
    printf("kind %d rounds %zu: %zu constraints, capture bad rows %zu (%.2f ms)\n", kind, rounds, c3[0], o4[3], ms);
    if (o4[3]) return 1;
  }
  printf("synthesis ms %.1f\n", bh_test_synthesis_ms(1, 1 << 16, 9, 2));
  return 0;
}

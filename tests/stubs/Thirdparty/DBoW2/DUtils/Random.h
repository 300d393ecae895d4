// test stub (tests/stubs/README.md): DUtils::Random::RandomInt as Thirdparty/DBoW2/DUtils/Random.cpp:40-43 defines it (libc rand()).
// With -DDVM_REF_DUTILS the function is only DECLARED and the driver links oracle/_ref/libdutils_ref.so -- the reference's own
// Random.cpp compiled where it lies (oracle/Makefile `_ref`); tests/test_ref_dutils.py checks the inline form below against it.
#pragma once
#include <cstdlib>
namespace DUtils {
class Random {
 public:
#ifdef DVM_REF_DUTILS
  static int RandomInt(int min, int max);
#else
  static int RandomInt(int min, int max) {
    int d = max - min + 1;
    return int(((double)rand() / ((double)RAND_MAX + 1.0)) * d) + min;
  }
#endif
};
}  // namespace DUtils

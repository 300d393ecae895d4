// test stub (tests/stubs/README.md): DUtils::Random::RandomInt as Thirdparty/DBoW2/DUtils/Random.cpp:40-43 defines it (libc rand())
#pragma once
#include <cstdlib>
namespace DUtils {
class Random {
 public:
  static int RandomInt(int min, int max) {
    int d = max - min + 1;
    return int(((double)rand() / ((double)RAND_MAX + 1.0)) * d) + min;
  }
};
}  // namespace DUtils

// test stub (tests/stubs/README.md): boost::uuids::uuid as a 16-byte value type with the comparisons the shims use
#pragma once
#include <cstdint>
#include <cstring>
namespace boost { namespace uuids {
struct uuid {
  uint8_t data[16];
  bool operator==(const uuid& o) const { return std::memcmp(data, o.data, 16) == 0; }
  bool operator!=(const uuid& o) const { return !(*this == o); }
  bool operator<(const uuid& o) const { return std::memcmp(data, o.data, 16) < 0; }
};
inline uuid nil_uuid() { uuid u; std::memset(u.data, 0, 16); return u; }
}}  // namespace boost::uuids

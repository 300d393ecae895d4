// test stub: DBoW2::FeatureVector / BowVector (see tests/stubs/README.md)
#pragma once
#include <map>
#include <vector>
namespace DBoW2 {
typedef unsigned int NodeId;
typedef unsigned int WordId;
typedef double WordValue;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};
class BowVector : public std::map<WordId, WordValue> {};
}  // namespace DBoW2

// Stand-in for DBoW2::FeatureVector (node id -> ascending feature indices), TEST INFRASTRUCTURE.
#pragma once
#include <map>
#include <vector>
namespace DBoW2 {
typedef unsigned int NodeId;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > {};
}  // namespace DBoW2

// test stub: DBoW2::BowVector lives with FeatureVector in one mock header (see tests/stubs/README.md)
#pragma once
#include "FeatureVector.h"

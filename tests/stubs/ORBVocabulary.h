// test stub (tests/stubs/README.md): inside the reference tree include/ORBVocabulary.h is replaced by the shim class
#pragma once
#include "ORBVocabulary_shim.h"

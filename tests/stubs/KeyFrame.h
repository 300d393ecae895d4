// test stub (tests/stubs/README.md)
#pragma once
#include "orbslam3_stub.h"

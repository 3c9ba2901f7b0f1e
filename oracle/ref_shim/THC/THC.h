// Stand-in for the torch header the reference's kernel headers include (TEST INFRASTRUCTURE ONLY):
// only the `at::Tensor` name is needed to parse the host-wrapper declarations, which are not compiled.
#pragma once
#include "cuda_host_shim.h"
namespace at { class Tensor; }

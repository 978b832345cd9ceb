// Name-compatibility forwarder: user code written against the reference's
// `#include "cuembed/include/embedding_lookup.cuh"` builds unchanged with
// `hipcc -I <repo>/cuembed_amd/csrc` (pass a hipStream_t where it passed a cudaStream_t).
#pragma once
#include "cuembed/include/embedding_lookup.hpp"

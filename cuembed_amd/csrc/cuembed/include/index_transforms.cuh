// Name-compatibility forwarder, see embedding_lookup.cuh.
#pragma once
#include "cuembed/include/index_transforms.hpp"

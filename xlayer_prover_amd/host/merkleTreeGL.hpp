// Forwarding header: the reference includes "merkleTreeGL.hpp"; everything lives in xlp_host.hpp.
#pragma once
#include "xlp_host.hpp"

// Forwarding header: the reference includes "poseidon_goldilocks.hpp"; everything lives in xlp_host.hpp.
#pragma once
#include "xlp_host.hpp"

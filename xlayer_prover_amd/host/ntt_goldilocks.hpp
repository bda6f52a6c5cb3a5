// Forwarding header: the reference includes "ntt_goldilocks.hpp"; everything lives in xlp_host.hpp.
#pragma once
#include "xlp_host.hpp"

// Forwarding header: the reference includes "friProve.hpp"; everything lives in xlp_host.hpp.
#pragma once
#include "xlp_host.hpp"

// Forwarding header: the reference includes "stark_info.hpp" (src/starkpil/stark_info.hpp); the mirror lives in xlp_host.hpp.
#pragma once
#include "xlp_host.hpp"

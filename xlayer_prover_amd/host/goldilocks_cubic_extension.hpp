// Forwarding header: the reference includes "goldilocks_cubic_extension.hpp"; everything lives in xlp_host.hpp.
#pragma once
#include "xlp_host.hpp"

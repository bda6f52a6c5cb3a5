// Forwarding header: the reference includes "goldilocks_base_field.hpp"; everything lives in xlp_host.hpp.
#pragma once
#include "xlp_host.hpp"

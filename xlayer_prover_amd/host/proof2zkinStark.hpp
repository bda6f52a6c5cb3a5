// Forwarding header: the reference includes "proof2zkinStark.hpp"; everything lives in xlp_host.hpp.
#pragma once
#include "xlp_host.hpp"

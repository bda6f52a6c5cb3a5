// Forwarding header: the reference includes "transcript.hpp"; everything lives in xlp_host.hpp.
#pragma once
#include "xlp_host.hpp"

// Forwarding header: the reference splits its interface over several headers;
// here every declaration lives in jpeg_amd.hpp.
#pragma once
#include "jpeg_amd.hpp"

// TEST INFRASTRUCTURE ONLY -- forwards to the stub core header.
#pragma once
#include "opencv2/core/core.hpp"

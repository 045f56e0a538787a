// TEST INFRASTRUCTURE: see ../../README.md
#pragma once
#include "spdlog/spdlog.h"

// placeholder — filled in below in this round
#pragma once
#include "pcs.hpp"

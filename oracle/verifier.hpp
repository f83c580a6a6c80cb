#pragma once
#include "pcs.hpp"

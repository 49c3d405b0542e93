#pragma once
#include "shared_ptr.hpp"

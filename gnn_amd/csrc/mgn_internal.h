// Shared by the translation units of libmgn_hip.so.
#pragma once
#include "../../include/mgn_hip.h"

void mgn_set_error(const char* fmt, ...);

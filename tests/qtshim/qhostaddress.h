// tests/qtshim/qhostaddress.h -- TEST INFRASTRUCTURE: forwards to the Qt stand-in (qtshim_core.h)
#pragma once
#include "qtshim_core.h"

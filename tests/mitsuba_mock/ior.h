// tests/mitsuba_mock: stand-in for src/bsdfs/ior.h (lookupIOR is declared in mock.h)
#include "mitsuba/mock.h"

// tests/mitsuba_mock: stand-in for src/bsdfs/microfacet.h (nothing of it is used by the shells)
#include "mitsuba/mock.h"

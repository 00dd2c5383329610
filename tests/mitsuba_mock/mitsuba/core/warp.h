// tests/mitsuba_mock: see ../mock.h (test infrastructure only)
#include "../mock.h"

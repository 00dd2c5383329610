/* dj_brdf.h -- compatibility header: lets code written against jdupuy/dj_brdf's single header
 * (`#define DJ_BRDF_IMPLEMENTATION 1` + `#include "dj_brdf.h"`) compile unchanged against the
 * MI355X engine.  It only pulls in the djb:: facade (djb_hip.hpp, link with -ldjb_hip) and the
 * standard headers the reference header itself includes (dj_brdf.h:38-39, 545-558), which client
 * code such as the reference's examples/merl_params.cpp relies on transitively (assert, printf).
 * The reference's own test and example programs build against this header without modification:
 * examples/Makefile target `reftests`, tests/test_gpu_golden.py::test_reference_programs_unchanged. */
#ifndef DJ_BRDF_COMPAT_H
#define DJ_BRDF_COMPAT_H

#include <vector>
#include <string>
#include <cmath>
#include <cstdarg>
#include <iostream>
#include <fstream>
#include <cstring>
#include <stdint.h>
#include <assert.h>
#include <stdio.h>

#include "djb_hip.hpp"

#endif

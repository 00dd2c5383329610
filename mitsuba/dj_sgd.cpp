// mitsuba/dj_sgd.cpp -- Mitsuba 0.5 BSDF plugin "dj_sgd" on top of the MI355X engine.
// Identical shell to dj_abc (the reference's two files differ only in the model class,
// jdupuy/dj_brdf mitsuba/dj_sgd.cpp:19-141 vs dj_abc.cpp:19-141); instantiated with djb::sgd.
#define DJ_MODEL sgd
#define DJ_PLUGIN dj_sgd
#define DJ_PLUGIN_STR "dj_sgd"
#include "dj_abc.cpp"

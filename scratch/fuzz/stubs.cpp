// vs_params_default for the CPU-only fuzz build (the real one lives in vs_api.cpp next to device code)
#include <cstring>
#include "vs_stab.h"
extern "C" void vs_params_default(vs_params_c* p) { memset(p, 0, sizeof *p); p->struct_size = sizeof *p; p->smoothing_radius = 30; }

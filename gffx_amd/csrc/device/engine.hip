// engine.hip -- the whole engine as ONE translation unit, for the tools that compile it into a single binary with their own
// knobs (tools/kbench.hip, tools/win_index_check.hip).  The library is built from the five files below, one object each
// (gffx_amd/csrc/Makefile); engine_private.hpp says what is where.
#define GFFX_WINDOWS_LAUNCH_ALL  // (windows_launch.hpp: all three kinds of launch in this one unit)
#include "engine_index.hip"
#include "engine_batch.hip"
#include "engine_windows.hip"
#include "engine_regions.hip"
#include "engine_depth.hip"

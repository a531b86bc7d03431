// engine_windows_plain.hip -- the window kernels' instantiations for one kind of launch (kLaunchPlain: join_pairs_kernels.hpp), a
// translation unit of its own so that the three kinds compile side by side (windows_launch.hpp).
#define GFFX_WINDOWS_LAUNCH_KIND gffx::kLaunchPlain
#include "windows_launch.hpp"

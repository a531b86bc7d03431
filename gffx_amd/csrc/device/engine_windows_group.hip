// engine_windows_group.hip -- the window kernels' instantiations for one kind of launch (kLaunchGroup: join_pairs_kernels.hpp), a
// translation unit of its own so that the three kinds compile side by side (windows_launch.hpp).
#define GFFX_WINDOWS_LAUNCH_KIND gffx::kLaunchGroup
#include "windows_launch.hpp"

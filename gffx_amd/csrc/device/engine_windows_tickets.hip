// engine_windows_tickets.hip -- the window kernels' instantiations for one kind of launch (kLaunchTickets: join_pairs_kernels.hpp), a
// translation unit of its own so that the three kinds compile side by side (windows_launch.hpp).
#define GFFX_WINDOWS_LAUNCH_KIND gffx::kLaunchTickets
#include "windows_launch.hpp"

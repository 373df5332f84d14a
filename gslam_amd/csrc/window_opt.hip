// window_opt.hip — the same optimiser state machine sized for the backend's window pose refinement
// (gslam/backend.py:447-506: torch.optim.LBFGS(history_size=10, strong_wolfe, tolerance_change=1e-7) over the poses
// of up to 8 keyframes, 9 parameters each, one `.item()` per closure at backend.py:501).  SURVEY.md 8f rank 2.
#include "gsx_common.h"
#include "pose_math.h"

#define TO_MAXN 80
#define TO_MAXH 10
#include "track_opt.h"
#include "track_tail.h"

#define TO_ENTRY(name) gsx_window_opt_##name
#define TO_TENSORS 16
#include "track_opt_impl.inc"

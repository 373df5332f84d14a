// track_opt.hip — device-resident tracking optimiser (track_opt.h) behind the C ABI: one single-wavefront launch per
// closure evaluation replaces the host-side torch.optim.Adam / torch.optim.LBFGS logic of
// gslam/frontend.py:604-662 and its per-closure `loss.item()` (frontend.py:648).
#include "gsx_common.h"
#include "pose_math.h"
#include "track_opt.h"
#include "track_tail.h"

#define TO_ENTRY(name) gsx_track_opt_##name
#define TO_TENSORS 4
#include "track_opt_impl.inc"

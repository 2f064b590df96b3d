// pdt_chain_wide_f32.hip -- the slow-wrap variants of the PLL kernels in float (k_pll_phase / _acquire_pipe / _head / _fix <float, true>):
// taken when the loop gains are so large that a single +-2 pi correction per step is not enough (run_capture: slow_wrap -- a
// caller's own loop constants, pdt_set_loop_params).  A third of the library's device code between the two precisions; units
// of their own so that they compile beside the rest.
#include "pdt_rt.h"

namespace pdtrt {
PDT_WIDE_INSTANCES(, float)
}

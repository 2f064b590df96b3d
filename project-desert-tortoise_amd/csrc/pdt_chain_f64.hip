// pdt_chain_f64.hip -- the chain in double (ARGOSdemod's build): run_capture<double>, finish_capture<double>, the stage
// entries, and every kernel they launch (but the PLL kernels' slow-wrap variants: pdt_chain_wide_f32.hip / _f64.hip).
#include "pdt_chain.inc"

namespace pdtrt {
PDT_CHAIN_INSTANCES(, double)
}

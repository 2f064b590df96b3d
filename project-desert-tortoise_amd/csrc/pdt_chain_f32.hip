// pdt_chain_f32.hip -- the chain in float (POESTIPdemod's build, both sound-card twins): run_capture<float>, finish_capture<float>,
// the stage entries, and every kernel they launch (but the PLL kernels' slow-wrap variants: pdt_chain_wide_f32.hip / _f64.hip).
#include "pdt_chain.inc"

namespace pdtrt {
PDT_CHAIN_INSTANCES(, float)
}

#!/bin/bash
# weak signals: default build vs variants (PDT_LIBPDT_PATH) and warm-up scales, on noise x4 .. x8
for v in default variants/libpdt_k1024.so; do for ws in 1.0 1.3; do
  if [ $v = default ]; then unset PDT_LIBPDT_PATH; else export PDT_LIBPDT_PATH=$PWD/$v; fi
  export PDT_PLL_WARM_SCALE=$ws
  echo "== $v warm scale $ws"; python tests/tools/noise_sweep.py 4 5 6 8 2>&1 | grep "noise x" | sed 's/identical True; //; s/agc fixes.*//'
done; done

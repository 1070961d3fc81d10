#!/bin/bash
# key generation at C3: the default (hybrid) against the dense left-looking form and the pure stream form (experiments build for the switches and the phase clock)
export PSF_LIB="$(cd "$(dirname "$0")/.." && pwd)/tools_amd/lib/libpsf_mi355x_exp.so"
for form in default gemm stream; do
  echo "== PSF_CHOL=$form"
  if [ $form = default ]; then PSF_KEYGEN_TIMING=1 python tools/keygen_time.py c3 2>&1 | tail -14; else PSF_CHOL=$form python tools/keygen_time.py c3 2>&1 | tail -2; fi
done
unset PSF_LIB
echo "== release library"; python tools/keygen_time.py c3 c2 c4 | tail -6

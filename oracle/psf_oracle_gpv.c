/* placeholder translation unit: PSFGPV / PSFGPVRing oracle (filled in below) */
#include "psf_oracle.h"

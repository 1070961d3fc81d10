"""tools_amd -- MI355X-native preimage sampling (PSF trait of qfall/tools) over libpsf_mi355x.so.

Host-side mirror of the reference's interface for this path (src/primitive/psf.rs:39-81):
GadgetParameters.init_default, PSFPerturbation / PSFGPV / PSFGPVRing with
trap_gen / samp_d / samp_p / f_a / check_domain.  Everything computes on the GPU through the C ABI.
"""
from ._ffi import PsfError, LIB_PATH  # noqa: F401
from .psf import GadgetParameters, GadgetParametersRing, PSFPerturbation, PSFGPV, PSFGPVRing  # noqa: F401
from . import gadget  # noqa: F401
from . import textio  # noqa: F401
from . import serde_json  # noqa: F401

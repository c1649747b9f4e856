"""MI355X-native OpenPose-style inference path (rtpose_vgg forward + pafprocess
post-processing) behind the reference's own Python API.  See DESIGN.md."""
from . import _capi  # noqa: F401  (fails loudly if the HIP library is not built)
from .network import get_model, RtposeVGG  # noqa: F401

__all__ = ["get_model", "RtposeVGG"]

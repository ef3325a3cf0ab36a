"""agora-blsful_amd: MI355X-native batch BLS12-381 signature verification behind the verify-path interface of
dashpay/agora-blsful.  csrc/ holds the HIP kernels and the C ABI (include/blsgpu.h); api.py is the host-side mirror
of the reference interface.  The directory name is not a Python identifier; load it with `import_pkg()` from
`__graft_entry__.py` (or importlib) under the module name `agora_blsful_amd`."""
from .api import *  # noqa: F401,F403
from . import api  # noqa: F401

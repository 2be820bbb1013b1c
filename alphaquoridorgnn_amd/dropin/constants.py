"""Drop-in shim: put alphaquoridorgnn_amd/dropin/ ahead of the reference directory on sys.path and the
reference's `import constants` / `from constants import ...` resolve to the MI355X-native implementation."""
from alphaquoridorgnn_amd.constants import *  # noqa: F401,F403
from alphaquoridorgnn_amd import constants as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

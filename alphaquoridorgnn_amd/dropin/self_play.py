"""Drop-in shim: put alphaquoridorgnn_amd/dropin/ ahead of the reference directory on sys.path and the
reference's `import self_play` / `from self_play import ...` resolve to the MI355X-native implementation."""
from alphaquoridorgnn_amd.self_play import *  # noqa: F401,F403
from alphaquoridorgnn_amd import self_play as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

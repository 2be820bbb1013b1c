"""Drop-in shim: put alphaquoridorgnn_amd/dropin/ ahead of the reference directory on sys.path and the
reference's `import train_network` / `from train_network import ...` resolve to the MI355X-native implementation."""
from alphaquoridorgnn_amd.train_network import *  # noqa: F401,F403
from alphaquoridorgnn_amd import train_network as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

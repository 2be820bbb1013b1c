"""Drop-in shim: put alphaquoridorgnn_amd/dropin/ ahead of the reference directory on sys.path and the
reference's `import agents` / `from agents import ...` resolve to this repository's baseline opponents."""
from alphaquoridorgnn_amd.agents import *  # noqa: F401,F403
from alphaquoridorgnn_amd import agents as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

"""Drop-in shim: `python -m train_cycle`-style use with alphaquoridorgnn_amd/dropin/ ahead of the reference on sys.path."""
from alphaquoridorgnn_amd.train_cycle import *  # noqa: F401,F403
from alphaquoridorgnn_amd import train_cycle as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

if __name__ == '__main__':
    _impl.train_cycle()

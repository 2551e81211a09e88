"""Drop-in module name of the reference (cama/pose_transformer.py): re-exports the MI355X implementation."""
from cama_amd.pose_transformer import *  # noqa: F401,F403
from cama_amd import pose_transformer as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

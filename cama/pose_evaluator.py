"""Drop-in module name of the reference (cama/pose_evaluator.py): re-exports the MI355X build's implementation."""
from cama_amd.pose_evaluator import *  # noqa: F401,F403
from cama_amd import pose_evaluator as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

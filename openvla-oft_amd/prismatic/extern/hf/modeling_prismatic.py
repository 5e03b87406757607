"""prismatic.extern.hf.modeling_prismatic (mirror of the reference module path)."""
from ....modeling import OpenVLAForActionPrediction, PrismaticCausalLMOutputWithPast  # noqa: F401

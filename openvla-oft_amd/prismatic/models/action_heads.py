"""prismatic.models.action_heads (mirror of the reference module path)."""
from ...diffusion import SinusoidalPositionalEncoding  # noqa: F401
from ...modeling import DiffusionActionHead, L1RegressionActionHead  # noqa: F401

"""prismatic.models.action_heads (mirror of the reference module path)."""
from ...modeling import L1RegressionActionHead  # noqa: F401

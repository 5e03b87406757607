"""prismatic.models.projectors (mirror of the reference module path)."""
from ...modeling import NoisyActionProjector, ProprioProjector  # noqa: F401

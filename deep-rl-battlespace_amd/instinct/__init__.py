"""Mirror of the reference's `instinct` package (instinct/team.py, instinct/agent.py): the scripted opponent, on device."""
from . import team  # noqa: F401
from .team import Team  # noqa: F401

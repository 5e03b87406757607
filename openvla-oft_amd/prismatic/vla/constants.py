"""Platform constants (mirror of the reference's prismatic/vla/constants.py:11-89).

The reference picks the platform by substring-matching `sys.argv` at import time (constants.py:56-70); the same rule is
kept as the default, and `set_platform()` makes the choice explicit for library use."""
import sys
from enum import Enum

IGNORE_INDEX = -100
ACTION_TOKEN_BEGIN_IDX = 31743
STOP_INDEX = 2


class NormalizationType(str, Enum):
    NORMAL = "normal"
    BOUNDS = "bounds"
    BOUNDS_Q99 = "bounds_q99"


LIBERO_CONSTANTS = {"NUM_ACTIONS_CHUNK": 8, "ACTION_DIM": 7, "PROPRIO_DIM": 8, "ACTION_PROPRIO_NORMALIZATION_TYPE": NormalizationType.BOUNDS_Q99}
UR5E_CONSTANTS = {"NUM_ACTIONS_CHUNK": 8, "ACTION_DIM": 7, "PROPRIO_DIM": 6, "ACTION_PROPRIO_NORMALIZATION_TYPE": NormalizationType.BOUNDS}
ALOHA_CONSTANTS = {"NUM_ACTIONS_CHUNK": 25, "ACTION_DIM": 14, "PROPRIO_DIM": 14, "ACTION_PROPRIO_NORMALIZATION_TYPE": NormalizationType.BOUNDS}
BRIDGE_CONSTANTS = {"NUM_ACTIONS_CHUNK": 5, "ACTION_DIM": 7, "PROPRIO_DIM": 7, "ACTION_PROPRIO_NORMALIZATION_TYPE": NormalizationType.BOUNDS_Q99}
_PLATFORMS = {"LIBERO": LIBERO_CONSTANTS, "ALOHA": ALOHA_CONSTANTS, "BRIDGE": BRIDGE_CONSTANTS, "UR5E": UR5E_CONSTANTS}


def detect_robot_platform() -> str:
    cmd_args = " ".join(sys.argv).lower()
    for key in ("libero", "aloha", "bridge", "ur5e"):
        if key in cmd_args:
            return key.upper()
    return "LIBERO"


def set_platform(name: str) -> None:
    global ROBOT_PLATFORM, NUM_ACTIONS_CHUNK, ACTION_DIM, PROPRIO_DIM, ACTION_PROPRIO_NORMALIZATION_TYPE
    c = _PLATFORMS[name.upper()]
    ROBOT_PLATFORM = name.upper()
    NUM_ACTIONS_CHUNK, ACTION_DIM, PROPRIO_DIM = c["NUM_ACTIONS_CHUNK"], c["ACTION_DIM"], c["PROPRIO_DIM"]
    ACTION_PROPRIO_NORMALIZATION_TYPE = c["ACTION_PROPRIO_NORMALIZATION_TYPE"]


set_platform(detect_robot_platform())

"""prismatic.extern.hf.configuration_prismatic: the fields of OpenVLAConfig the hot path depends on."""
from ....config import OPENVLA_7B, VitConfig, VLAConfig as OpenVLAConfig  # noqa: F401

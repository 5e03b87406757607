"""openvla-oft_amd: MI355X (gfx950) native implementation of the OpenVLA-OFT parallel-decoding action-chunk
forward/backward path.  The directory name carries a hyphen, so import it with

    import importlib; ovla = importlib.import_module("openvla-oft_amd")

(`tests/conftest.py`, `bench.py` and `__graft_entry__.py` do exactly this and alias it as `ovla_amd`).
"""
__version__ = "0.1.0"

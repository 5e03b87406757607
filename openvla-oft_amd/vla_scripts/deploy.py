"""vla-scripts/deploy.py of the reference on the HIP engine: a FastAPI server whose `/act` endpoint maps
{observation images, state, "instruction"} to an action chunk (deploy.py:47-107).

Differences, all deliberate:
  * the reference guards nothing: `get_server_action` runs in FastAPI's thread pool with no lock around the model (deploy.py:78-107).
    One GPU engine with static hipGraph input buffers is not re-entrant, so requests are serialised by a lock here;
  * `json_numpy` (the reference's wire codec, absent from this image: PARITY UNPINNED) is restated in `encode_ndarray` /
    `decode_payload`: an ndarray travels as {"__numpy__": base64(raw bytes), "dtype": numpy descr string, "shape": [...]};
    the "encoded" double-encoding of deploy.py:80-95 (whole payload as ONE json string under the key "encoded") is kept;
  * components can be handed in already built (tests; no checkpoint exists offline), otherwise they are loaded exactly like the
    reference does (get_vla / get_action_head / get_proprio_projector / get_processor);
  * merged LoRA weights + hipGraph replay are switched on (`vla.enable_graph_replay()`): the deployment configuration of DESIGN.md §6.
"""
from __future__ import annotations

import base64
import json
import logging
import threading
import traceback
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, Optional, Union

import numpy as np

from ..experiments.robot import openvla_utils as U
from ..prismatic.vla import constants as C


def encode_ndarray(a: np.ndarray) -> Dict[str, Any]:
    a = np.ascontiguousarray(a)
    return {"__numpy__": base64.b64encode(a.tobytes()).decode("ascii"), "dtype": np.lib.format.dtype_to_descr(a.dtype), "shape": list(a.shape)}


def _decode(obj):
    if isinstance(obj, dict):
        if "__numpy__" in obj:
            dt = np.lib.format.descr_to_dtype(obj["dtype"])
            return np.frombuffer(base64.b64decode(obj["__numpy__"]), dtype=dt).reshape(obj["shape"]).copy()
        return {k: _decode(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_decode(v) for v in obj]
    return obj


def _encode(obj):
    if isinstance(obj, np.ndarray):
        return encode_ndarray(obj)
    if isinstance(obj, (np.floating, np.integer)):
        return obj.item()
    if isinstance(obj, dict):
        return {k: _encode(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_encode(v) for v in obj]
    return obj


def decode_payload(payload: Dict[str, Any]):
    """-> (observation dict with ndarrays, double_encoded flag).  deploy.py:80-86."""
    double = "encoded" in payload
    if double:
        assert len(payload.keys()) == 1, "Only uses encoded payload!"
        payload = json.loads(payload["encoded"])
    return _decode(payload), double


@dataclass
class DeployConfig:
    # fmt: off
    host: str = "0.0.0.0"
    port: int = 8777
    model_family: str = "openvla"
    pretrained_checkpoint: Union[str, Path] = ""
    use_l1_regression: bool = True
    use_diffusion: bool = False
    num_diffusion_steps: int = 50
    use_film: bool = False
    num_images_in_input: int = 3
    use_proprio: bool = True
    center_crop: bool = True
    num_open_loop_steps: int = 25
    unnorm_key: Union[str, Path] = ""
    use_relative_actions: bool = False
    load_in_8bit: bool = False
    load_in_4bit: bool = False
    seed: int = 7
    graph_replay: bool = True           # hipGraph replay of predict_action (not in the reference)
    # fmt: on


class OpenVLAServer:
    def __init__(self, cfg, *, vla=None, processor=None, action_head=None, proprio_projector=None):
        self.cfg = cfg
        self.vla = vla if vla is not None else U.get_vla(cfg)
        self.proprio_projector = proprio_projector
        if self.proprio_projector is None and cfg.use_proprio:
            self.proprio_projector = U.get_proprio_projector(cfg, self.vla.llm_dim, C.PROPRIO_DIM)
        self.action_head = action_head
        if self.action_head is None and (cfg.use_l1_regression or cfg.use_diffusion):
            self.action_head = U.get_action_head(cfg, self.vla.llm_dim)
        assert cfg.unnorm_key in self.vla.norm_stats, f"Action un-norm key {cfg.unnorm_key} not found in VLA `norm_stats`!"
        self.processor = processor if processor is not None else U.get_processor(cfg)
        if getattr(cfg, "graph_replay", True) and not cfg.use_diffusion and not cfg.use_film:
            self.vla.enable_graph_replay(True)
        self._lock = threading.Lock()

    def act(self, payload: Dict[str, Any]):
        """The body of `/act` without the HTTP layer: returns a list of actions (ndarrays), or the json_numpy-encoded string
        for a double-encoded request, or "error" (deploy.py:78-107)."""
        try:
            observation, double = decode_payload(payload)
            instruction = observation["instruction"]
            with self._lock:      # one engine, static graph buffers: serialise (the reference does not lock)
                action = U.get_vla_action(self.cfg, self.vla, self.processor, observation, instruction, action_head=self.action_head,
                                          proprio_projector=self.proprio_projector, use_film=self.cfg.use_film)
            return json.dumps(_encode(action)) if double else _encode(action)
        except Exception:  # noqa: BLE001 -- the reference answers "error" to any malformed request
            logging.error(traceback.format_exc())
            logging.warning("Your request threw an error; make sure your request complies with the expected format:\\n"
                            "{'observation': dict, 'instruction': str}\\n")
            return "error"

    def build_app(self):
        from fastapi import FastAPI
        from fastapi.responses import JSONResponse

        app = FastAPI()

        @app.post("/act")
        def get_server_action(payload: Dict[str, Any]):
            return JSONResponse(self.act(payload))

        self.app = app
        return app

    def run(self, host: str = "0.0.0.0", port: int = 8777) -> None:
        import uvicorn

        uvicorn.run(self.build_app(), host=host, port=port)


def deploy(cfg: DeployConfig) -> None:
    OpenVLAServer(cfg).run(cfg.host, port=cfg.port)

"""256-bin action (de)tokenizer (mirror of prismatic/vla/action_tokenizer.py:13-72)."""
from typing import List, Union

import numpy as np


class ActionTokenizer:
    def __init__(self, tokenizer, bins: int = 256, min_action: int = -1, max_action: int = 1) -> None:
        self.tokenizer, self.n_bins, self.min_action, self.max_action = tokenizer, bins, min_action, max_action
        self.bins = np.linspace(min_action, max_action, self.n_bins)
        self.bin_centers = (self.bins[:-1] + self.bins[1:]) / 2.0
        self.action_token_begin_idx: int = int(self.tokenizer.vocab_size - (self.n_bins + 1))

    def token_ids(self, action: np.ndarray) -> np.ndarray:
        action = np.clip(action, a_min=float(self.min_action), a_max=float(self.max_action))
        return self.tokenizer.vocab_size - np.digitize(action, self.bins)

    def __call__(self, action: np.ndarray) -> Union[str, List[str]]:
        ids = self.token_ids(action)
        if len(ids.shape) == 1:
            return self.tokenizer.decode(list(ids))
        return self.tokenizer.batch_decode(ids.tolist())

    def decode_token_ids_to_actions(self, action_token_ids: np.ndarray) -> np.ndarray:
        d = np.clip(self.tokenizer.vocab_size - action_token_ids - 1, a_min=0, a_max=self.bin_centers.shape[0] - 1)
        return self.bin_centers[d]

    @property
    def vocab_size(self) -> int:
        return self.n_bins
